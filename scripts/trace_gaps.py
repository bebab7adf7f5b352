"""Developer tool: prints the kernel sequence of the LAST step of a rocprofv3 --kernel-trace CSV with start offsets,
durations and gaps.  usage: python scripts/trace_gaps.py kernel_trace.csv [kernels per window]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-k:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:90]))
    prev_end = e
