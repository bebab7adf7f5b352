"""Developer tool: timing of config C5 (fastsum, Gaussian kernel, 1e6 sources x 1e6 targets, 3-D N=256, m=4)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch_nfft_amd as tn
from torch_nfft_amd import ops
n, N, m, sigma = 1_000_000, 256, 4, 0.1
gen = torch.Generator(device="cuda").manual_seed(20240)
src = (torch.rand((n, 3), generator=gen, device="cuda") - 0.5) * 0.5
tgt = (torch.rand((n, 3), generator=gen, device="cuda") - 0.5) * 0.5
x = torch.rand((n,), generator=gen, device="cuda")
coeffs = tn.gaussian_analytic_coeffs(sigma, dim=3, N=N)
def step():
    ops.plan_cache_clear()
    return tn.nfft_fastsum(x, coeffs, src, tgt, cutoff=m)
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 10
for _ in range(K): y = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("C5 fastsum 1e6 x 1e6, N=%d: %.2f ms per call, %.0f M source+target points/s" % (N, dt * 1e3, 2 * n / dt / 1e6))
