"""Developer tool: times nfft_hip_plan_points (C3 points, and a C5-like sparse set that takes the owned tiling) for the
library named by NFFT_HIP_LIB."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch_nfft_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
gen = torch.Generator(device="cuda").manual_seed(1)
for name, n, scale, flags in (("C3", 10_000_000, 1.0, 0), ("C5-like", 1_000_000, 0.25, _lib.POINTS_IN_QUARTER_BALL if hasattr(_lib, "POINTS_IN_QUARTER_BALL") else 1)):
    prob = _lib.Problem(3, n, 1, 1, 256, 4, flags)
    pos = (torch.rand((n, 3), generator=gen, device="cuda") - 0.5) * scale
    plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
    ts = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(pos), None, p(plan), plan.numel(), s))
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("%s %s: plan %.3f ms (median of 12, min %.3f)" % (os.environ.get("NFFT_HIP_LIB", "default"), name, ts[6], ts[0]), flush=True)
