"""Developer tool: adjoint+forward timing for a dense multi-column problem (which gather kernel wins where)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch_nfft_amd as tn
from torch_nfft_amd import _lib, ops
N, m = int(os.environ.get("NBAND", 256)), 4
n, C, B = int(os.environ.get("NPTS", 10_000_000)), int(os.environ.get("C", 8)), int(os.environ.get("B", 1))
gen = torch.Generator(device="cuda").manual_seed(5)
pos = torch.rand((n, 3), generator=gen, device="cuda") - 0.5
batch = (torch.arange(n, device="cuda") // (n // B)).clamp(max=B - 1) if B > 1 else None
x = torch.randn((n, C), generator=gen, device="cuda")
def step():
    ops.plan_cache_clear()
    y = tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
    return tn.nfft_forward(y, pos, batch, cutoff=m, real_output=True)
for _ in range(2): step()
torch.cuda.synchronize()
_lib.profile_enable(True); _lib.profile_collect()
K = 5
t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
st = _lib.profile_collect()
print("GATHER=%s OWNED=%s N=%d n=%d C=%d B=%d: %.2f ms  %s" % (os.environ.get("NFFT_HIP_GATHER", "-"), os.environ.get("NFFT_HIP_OWNED", "-"), N, n, C, B, dt * 1e3,
      {k: round(v[0] / K, 3) for k, v in st.items() if v[1]}), flush=True)
