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
