"""Turns the rocprofv3 PMC passes of a bench.py run into profiles/r02_spread_traffic.json (the `roofline.traffic`
record bench.py reports): FETCH_SIZE (KiB, doubled: gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md
section HBM) + WRITE_SIZE (KiB) of the dominant kernel, averaged per launch, stamped with the kernel's name, launch
shape, the workload string and a hash of the kernel's source files -- bench.py only uses the record while all of them
still match.  usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <bench line json> <out>"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("_bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def counter(d, kernel, name):
    vals, shape = [], None
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and kernel in r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("nfft::", ""):
                vals.append(float(r["Counter_Value"]))
                shape = (r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"))
    return vals, shape


fetch_dir, write_dir, line_path, out_path = sys.argv[1:5]
line = json.loads([l for l in open(line_path) if l.startswith("{")][-1])
kernel = line["roofline"]["kernel"]
short = kernel.split("<")[0] + "<" + kernel.split("<")[1].replace(" ", "")
fv, shape = counter(fetch_dir, kernel.replace(", ", ", "), "FETCH_SIZE")
wv, _ = counter(write_dir, kernel, "WRITE_SIZE")
assert fv and wv, "kernel %s not found in the PMC output" % kernel
rec = {
    "kernel": kernel,
    "workload": line["config"]["workload"],
    "source_hash": bench.kernel_source_hash(),
    "launches_sampled": [len(fv), len(wv)],
    "launch_shape": {"grid_size": shape[0], "workgroup_size": shape[1], "lds_bytes": shape[2], "vgprs": shape[3]},
    "FETCH_SIZE_KiB": sum(fv) / len(fv),
    "WRITE_SIZE_KiB": sum(wv) / len(wv),
    "fetch_bytes_corrected": 2.0 * 1024.0 * sum(fv) / len(fv),
    "write_bytes": 1024.0 * sum(wv) / len(wv),
    "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python3 bench.py "
            "--no-legs --no-cpu-baseline`; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM; WRITE_SIZE is exact "
            "for float atomics and 16-byte stores",
}
json.dump(rec, open(out_path, "w"), indent=1)
print(json.dumps(rec, indent=1))
