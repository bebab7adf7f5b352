"""Summarise rocprofv3 --pmc CSV output per kernel (developer tool)."""
import collections, csv, glob, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
        if pat and pat not in r["Kernel_Name"]:
            continue
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in agg.items():
        print(name)
        for c, v in sorted(cs.items()):
            print("   %-28s mean %.4g over %d dispatches" % (c, sum(v) / len(v), len(v)))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat and pat not in r["Kernel_Name"]:
            continue
        dur[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        print("%-60s n=%d mean %.1f us" % (k, len(v), sum(v) / len(v)))
