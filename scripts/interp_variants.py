"""Developer tool: times nfft_hip_interpolate at the C3 size for the library named by NFFT_HIP_LIB (variant builds from
scripts/exp_build.sh)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch_nfft_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
d, N, m, n = 3, int(os.environ.get("NBAND", 256)), int(os.environ.get("M_CUT", 4)), int(os.environ.get("NPTS", 10_000_000))
prob = _lib.Problem(d, n, 1, 1, N, m)
gen = torch.Generator(device="cuda").manual_seed(1)
pos = (torch.rand((n, d), generator=gen, device="cuda") - 0.5) * float(os.environ.get("SCALE", 1.0))
plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(pos), None, p(plan), plan.numel(), s))
grid = torch.rand((2 * N,) * 3, generator=gen, device="cuda")
y = torch.empty(n, device="cuda")
best = 1e9
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(lib.nfft_hip_interpolate(ctypes.byref(prob), p(plan), p(grid), 1, p(y), s))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    best = min(best, (t1 - t0) * 1e3)
print("%s: %.3f ms (interpolate), y sum %.6g" % (os.environ.get("NFFT_HIP_LIB", "default"), best, float(y.double().sum())), flush=True)
if hasattr(lib, "nfft_hip_debug_timing") or os.environ.get("IS_TIMING"):
    import numpy as np
    f = lib.nfft_hip_debug_timing
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 16)()
    f(None, 1)
    _lib.check(lib.nfft_hip_interpolate(ctypes.byref(prob), p(plan), p(grid), 1, p(y), s))
    torch.cuda.synchronize()
    f(buf, 0)
    v = np.array(list(buf), dtype=np.float64)
    names = ["fetch", "build", "wait", "products", "reduce", "store", "blocks", "total"]
    print("consumers: " + "  ".join("%s %.1f%%" % (names[i], 100 * v[i] / v[7]) for i in (0, 1, 2, 3, 4, 5)) + "  blocks %d  cycles/block %.0f" % (v[6], v[7] / max(v[6], 1)))
    pn = ["load wait", "poll", "convert", "planes", "total", "next_needed"]
    print("producers: " + "  ".join("%s %.1f%%" % (pn[i], 100 * v[8 + i] / v[12]) for i in (0, 1, 2, 5)) + "  planes %d  cycles/plane %.0f" % (v[11], v[12] / max(v[11], 1)))
