"""Times the spreading stage at the C3 size under debug variants (developer tool)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch_nfft_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
d, N, m, n = 3, 256, 4, int(os.environ.get("NPTS", 10_000_000))
prob = _lib.Problem(d, n, 1, 1, N, m)
gen = torch.Generator(device="cuda").manual_seed(1)
pos = torch.rand((n, d), generator=gen, device="cuda") - 0.5
x = torch.rand((n,), generator=gen, device="cuda")
plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(pos), None, p(plan), plan.numel(), s))
grid = torch.empty((2 * N,) * 3, device="cuda")
scratch = torch.empty(n + 128, device="cuda")
for mode in [int(v) for v in os.environ.get("MODES", "0,1,2,3,6,9").split(",")]:
    os.environ["NFFT_HIP_DBG"] = str(mode)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print("mode %d: %.3f ms (gather+memset+spread)" % (mode, (t1 - t0) * 1e3), flush=True)
