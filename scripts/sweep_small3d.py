"""Developer tool: adjoint + forward step time of small 3-D grids (N = 32, 48, 64) over cutoffs, point counts and point sets --
which tiling / kernel family serves them best (run once per setting of NFFT_HIP_SPREAD / NFFT_HIP_WIDE_MIN_M)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_nfft_amd as tn
gen = torch.Generator(device="cuda").manual_seed(3)
cases = [(N, m, n, B) for N in (32, 64) for m in (2, 3, 4, 6) for (n, B) in ((20_000, 1), (100_000, 1), (1_000_000, 1), (800_000, 8))]
cases += [(48, 3, 100_000, 1), (48, 4, 800_000, 8)]
if os.environ.get("SWEEP") == "fine":
    cases = [(32, m, n * B, B) for m in (4, 5) for B in (1, 4) for n in (50_000, 200_000, 500_000, 1_000_000)]
    cases += [(64, m, n * B, B) for m in (2, 3) for B in (4, 8, 16) for n in (20_000, 100_000)]
for N, m, n, B in cases:
    pos = torch.rand((n, 3), generator=gen, device="cuda") - 0.5
    x = torch.rand((n,), generator=gen, device="cuda")
    batch = None if B == 1 else (torch.arange(n, device="cuda") * B) // n
    for _ in range(2):
        y = tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
        f = tn.nfft_forward(y, pos, batch, cutoff=m, real_output=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        tn.ops.plan_cache_clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
        f = tn.nfft_forward(y, pos, batch, cutoff=m, real_output=True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("N=%d m=%d n=%8d B=%d: %7.3f ms/step" % (N, m, n, B, ts[4]), flush=True)
    del pos, x, y, f
