"""Developer tool: the figures of a bench.py JSON line, one leg per row."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f %s, n_gpus %d, ms/step %.3f (median %.3f)" % (d["value"], d["unit"], d["n_gpus"], d["ms_per_step"], d.get("ms_per_step_median", 0)))
r = d["roofline"]
print("spread %.4f ms  frac %.4f  bound %s  traffic %s" % (r["avg_launch_ms"], r["frac"], r["bound"], r["traffic"]))
if "roofline_interp" in d:
    print("gather %.4f ms  frac %.4f" % (d["roofline_interp"]["avg_launch_ms"], d["roofline_interp"]["frac"]))
print("stages", {k: round(v, 4) for k, v in d["stage_ms_per_launch"].items()})
for k, v in d["configs"].items():
    print("%-14s" % k, v.get("error") or ("%.4f ms  %.1f %s" % (v["ms_per_step_median"], v["value"], v["unit"])), {a: round(b, 3) for a, b in v.get("stage_ms_per_step", {}).items()})
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
