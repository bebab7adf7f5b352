"""Developer tool: where the waves of spread_ring_kernel spend their time, per role, from a trace build
(the archived experiment scripts/experiments/r04_spread_ring.hip copied back to torch_nfft_amd/csrc/spread_ring.hip, hooked into
launch_spread_mfma and built with scripts/exp_build.sh ...:"-DNFFT_HIP_TRACE=1"; NFFT_HIP_LIB=<that library>; results: profiles/r04_ring_trace.txt)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch_nfft_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
d, N, m, n = 3, 256, int(os.environ.get("M_CUT", 4)), 10_000_000
prob = _lib.Problem(d, n, 1, 1, N, m)
gen = torch.Generator(device="cuda").manual_seed(1)
pos = torch.rand((n, d), generator=gen, device="cuda") - 0.5
x = torch.rand((n,), generator=gen, device="cuda")
plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(pos), None, p(plan), plan.numel(), s))
grid = torch.empty((2 * N,) * 3, device="cuda")
scratch = torch.empty(n + 256, device="cuda")
trace = torch.zeros((16, 16, 4), dtype=torch.int64, device="cuda")
assert lib.nfft_dbg_set_ring_trace(p(trace)) == 0
phase = torch.zeros((16, 16, 8), dtype=torch.int64, device="cuda")
assert lib.nfft_dbg_set_ring_phase(p(phase)) == 0
for it in range(3):
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
torch.cuda.synchronize()
trace.zero_()
phase.zero_()
_lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.float64)
print("shader-clock cycles per wave over one work item (mean of the first 16 workgroups); K-blocks per item: see builders' units x 4")
for name, waves in (("owners   (waves 0-11)", range(0, 12)), ("stagers  (waves 12, 13)", (12, 13)), ("builders (waves 14, 15)", range(14, 16))):
    w = t[:, list(waves), :]
    print("%-24s loop %8.0f   waiting for flags %8.0f (%4.1f %%)   second wait class (owners: flush) %7.0f   units of work %6.1f   cycles per unit %7.1f"
          % (name, w[..., 0].mean(), w[..., 1].mean(), 100 * w[..., 1].mean() / max(w[..., 0].mean(), 1), w[..., 3].mean(), w[..., 2].mean(),
             (w[..., 0].mean() - w[..., 1].mean()) / max(w[..., 2].mean(), 1)))
ph = phase.cpu().numpy().astype(np.float64)[:, :12, :].mean(axis=(0, 1))
hits = t[:, :12, 2].mean()
names = ["wait raw + packed arithmetic", "look (when out of known K-blocks)", "scan for the next hit", "issue raw requests", "wait B + MFMAs",
         "issue B requests + publish progress", "pipeline empty: poll / flush", "loop top"]
print("owner phases, cycles per hit (s_memtime stamps: ~40 cycles each, included):")
for k in range(8):
    print("  %-42s %7.1f" % (names[k], ph[k] / max(hits, 1)))
