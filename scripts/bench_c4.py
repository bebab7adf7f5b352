"""Developer tool: timing + stage split of a batched problem (B point sets x C coefficient columns, NPER points per set, bandwidth NB,
cutoff MC from the environment; default: a C4-shaped one) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch_nfft_amd as tn
from torch_nfft_amd import _lib, ops
N, m, B, C, n_per = int(os.environ.get("NB", 128)), int(os.environ.get("MC", 4)), int(os.environ.get("B", 4)), int(os.environ.get("C", 64)), int(os.environ.get("NPER", 250_000))
gen = torch.Generator(device="cuda").manual_seed(5)
pos = torch.rand((B * n_per, 3), generator=gen, device="cuda") - 0.5
batch = torch.arange(B * n_per, device="cuda") // n_per
x = torch.randn((B * n_per, C), generator=gen, device="cuda")
def step():
    ops.plan_cache_clear()
    y = tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
    return tn.nfft_forward(y, pos, batch, cutoff=m, real_output=True)
for _ in range(2): step()
torch.cuda.synchronize()
_lib.profile_enable(True); _lib.profile_collect()
t0 = time.perf_counter()
K = int(os.environ.get("K", 5))
for _ in range(K): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
st = _lib.profile_collect()
print("B=%d C=%d n/set=%d N=%d: %.2f ms per adjoint+forward, %.1f M point-columns/s" % (B, C, n_per, N, dt * 1e3, B * n_per * C / dt / 1e6))
print({k: round(v[0] / K, 3) for k, v in st.items() if v[1]})
