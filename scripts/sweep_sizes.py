"""Developer tool: adjoint + forward step time over point counts and cutoffs (3-D, N = 256), to look for cliffs around
the kernel-selection thresholds (owner-computes spreading below 0.03 points per cell, streamed gather + column groups
from ~5.5e6 points)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_nfft_amd as tn
N = 256
gen = torch.Generator(device="cuda").manual_seed(3)
for m, n in [(4, 500_000), (4, 1_000_000), (4, 2_000_000), (4, 4_000_000), (4, 5_000_000), (4, 5_400_000), (4, 5_600_000),
             (4, 6_000_000), (4, 8_000_000), (4, 10_000_000), (4, 20_000_000), (2, 10_000_000), (3, 10_000_000),
             (5, 10_000_000), (6, 10_000_000), (7, 10_000_000)]:
    pos = torch.rand((n, 3), generator=gen, device="cuda") - 0.5
    x = torch.rand((n,), generator=gen, device="cuda")
    for _ in range(2):
        y = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
        f = tn.nfft_forward(y, pos, None, cutoff=m, real_output=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        tn.ops.plan_cache_clear()
        t0 = time.perf_counter()
        y = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
        f = tn.nfft_forward(y, pos, None, cutoff=m, real_output=True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("m=%d n=%9d: %7.3f ms/step  %8.1f Mpoints/s" % (m, n, ts[2], n / ts[2] / 1e3), flush=True)
    del pos, x, y, f
