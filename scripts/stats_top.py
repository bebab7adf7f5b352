"""Developer tool: print the top kernels of a rocprofv3 --kernel-trace --stats output directory (csv)."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f, "total %.1f ms" % (tot / 1e6))
    for r in rows[:n]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("nfft::", "")
        print("%-70s calls %5s avg %9.1f us  total %8.2f ms  %5.1f%%" % (name[:70], r["Calls"], float(r["AverageNs"]) / 1e3,
              float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
