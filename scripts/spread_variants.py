"""Developer tool: times nfft_hip_spread (gather + zero-fill + spreading) at the C3 size for the library named by
NFFT_HIP_LIB (variant builds from scripts/exp_build.sh) in the mode given by NFFT_HIP_SPREAD."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch_nfft_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
d, N, m, n = 3, int(os.environ.get("NBAND", 256)), int(os.environ.get("M_CUT", 4)), int(os.environ.get("NPTS", 10_000_000))
prob = _lib.Problem(d, n, 1, 1, N, m)
gen = torch.Generator(device="cuda").manual_seed(1)
pos = torch.rand((n, d), generator=gen, device="cuda") - 0.5
x = torch.rand((n,), generator=gen, device="cuda")
plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(pos), None, p(plan), plan.numel(), s))
grid = torch.empty((2 * N,) * 3, device="cuda")
scratch = torch.empty(n + 256, device="cuda")
best = 1e9
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    best = min(best, (t1 - t0) * 1e3)
print("%s: %.3f ms (gather+memset+spread), grid sum %.6g" % (os.environ.get("NFFT_HIP_LIB", "default"), best, float(grid.double().sum())), flush=True)
if hasattr(lib, "nfft_dbg_read"):
    buf = (ctypes.c_ulonglong * 16)()
    lib.nfft_dbg_read(buf)  # clear
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
    torch.cuda.synchronize()
    lib.nfft_dbg_read(buf)
    for role, base in (("plane owners", 0), ("builders", 8)):
        v = list(buf)[base:base + 8]
        nw, steps = v[6], v[7]
        if nw:
            print("%s: waves %d, steps/wave %.1f; per wave-step (memtime ticks): stage %.0f acc %.0f build %.0f barrier %.0f; loop/wave %.0f, final flush %.0f"
                  % (role, nw, steps / nw, v[0] / steps, v[1] / steps, v[2] / steps, v[3] / steps, v[4] / nw, v[5] / nw))
