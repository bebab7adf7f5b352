"""Developer tool: A/B of library builds at stage level on ONE box (box-to-box differences are +-3 %, as large as most
single changes).  usage: python scripts/ab_stage.py lib1.so lib2.so ...   (first = reference)
Every library is loaded in its own process (they share a soname) through bare ctypes -- no symbol / ABI check, so builds
of older commits work -- and times nfft_hip_plan_points, nfft_hip_spread and nfft_hip_interpolate at the C3 size with
GPU events; the processes are interleaved ROUNDS times so that clock drift hits all builds alike."""
import ctypes, os, subprocess, sys, json

CHILD = r'''
import ctypes, os, sys, json
import torch
lib = ctypes.CDLL(sys.argv[1])
class Problem(ctypes.Structure):
    _fields_ = [("dim", ctypes.c_int32), ("flags", ctypes.c_int32), ("num_points", ctypes.c_int64),
                ("num_columns", ctypes.c_int64), ("batch_size", ctypes.c_int64), ("N", ctypes.c_int64), ("m", ctypes.c_int64)]
vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
P = ctypes.POINTER(Problem)
lib.nfft_hip_plan_bytes.argtypes = [P]; lib.nfft_hip_plan_bytes.restype = i64
lib.nfft_hip_plan_points.argtypes = [P, vp, vp, vp, i64, vp]; lib.nfft_hip_plan_points.restype = ci
lib.nfft_hip_spread.argtypes = [P, vp, vp, i64, vp, vp, vp]; lib.nfft_hip_spread.restype = ci
lib.nfft_hip_spread_scratch_bytes.argtypes = [P, i64]; lib.nfft_hip_spread_scratch_bytes.restype = i64
lib.nfft_hip_interpolate.argtypes = [P, vp, vp, i64, vp, vp]; lib.nfft_hip_interpolate.restype = ci
lib.nfft_hip_last_error.restype = ctypes.c_char_p
d, N, m, n = 3, int(os.environ.get("NBAND", 256)), int(os.environ.get("M_CUT", 4)), int(os.environ.get("NPTS", 10_000_000))
prob = Problem(d, 0, n, 1, 1, N, m)
gen = torch.Generator(device="cuda").manual_seed(1)
if os.environ.get("CLUSTERS") == "1":
    centres = torch.rand((8, d), generator=gen, device="cuda") - 0.5
    which = torch.randint(0, 8, (n,), generator=gen, device="cuda")
    pos = centres[which] + 0.05 * torch.randn((n, d), generator=gen, device="cuda")
    pos = pos - torch.floor(pos + 0.5)
else:
    pos = torch.rand((n, d), generator=gen, device="cuda") - 0.5
x = torch.rand((n,), generator=gen, device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
grid = torch.empty((2 * N,) * 3, device="cuda")
scratch = torch.empty(lib.nfft_hip_spread_scratch_bytes(ctypes.byref(prob), 1) // 4 + 1024, device="cuda")
y = torch.empty(n, device="cuda")
def chk(rc):
    if rc: raise RuntimeError(lib.nfft_hip_last_error().decode())
def timed(fn, reps):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]
f_plan = lambda: chk(lib.nfft_hip_plan_points(ctypes.byref(prob), p(pos), None, p(plan), plan.numel(), s))
f_spread = lambda: chk(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
f_interp = lambda: chk(lib.nfft_hip_interpolate(ctypes.byref(prob), p(plan), p(grid), 1, p(y), s))
f_plan(); f_spread(); f_interp(); torch.cuda.synchronize()
reps = int(os.environ.get("REPS", 15))
out = {"plan": timed(f_plan, reps), "spread": timed(f_spread, reps), "interp": timed(f_interp, reps),
       "grid_sum": float(grid.double().sum()), "y_sum": float(y.double().sum())}
print("RESULT " + json.dumps(out))
'''

libs = sys.argv[1:]
rounds = int(os.environ.get("ROUNDS", 3))
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(l)], capture_output=True, text=True, timeout=600)
        line = [x for x in out.stdout.splitlines() if x.startswith("RESULT ")]
        if not line:
            print("FAILED", l, out.stderr[-1500:], flush=True)
            continue
        res[l].append(json.loads(line[0][7:]))
for l in libs:
    if not res[l]:
        continue
    def med(stage, k=0):
        v = sorted(x[stage][k] for x in res[l])
        return v[len(v) // 2]
    print("%-40s plan %.3f  spread(+zero+max) %.3f  interp %.3f   (min: %.3f %.3f %.3f)  grid sum %.6g  y sum %.6g"
          % (os.path.basename(l), med("plan"), med("spread"), med("interp"), med("plan", 1), med("spread", 1), med("interp", 1),
             res[l][0]["grid_sum"], res[l][0]["y_sum"]), flush=True)
