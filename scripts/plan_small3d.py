import ctypes, torch, time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_nfft_amd as tn
from torch_nfft_amd import ops
# time the plan of a 1e5-point 3-D problem (its scan has ~23 500 items)
gen = torch.Generator(device="cuda").manual_seed(1)
for n in (100_000, 300_000):
    pos = torch.rand((n, 3), generator=gen, device="cuda") - 0.5
    x = torch.rand((n,), generator=gen, device="cuda")
    from torch_nfft_amd import _lib
    for _ in range(3):
        ops.plan_cache_clear(); y = tn.nfft_adjoint(x, pos, None, bandwidth=256, cutoff=4)
    torch.cuda.synchronize()
    _lib.profile_enable(True); _lib.profile_collect()
    for _ in range(20):
        ops.plan_cache_clear(); y = tn.nfft_adjoint(x, pos, None, bandwidth=256, cutoff=4)
    torch.cuda.synchronize()
    st = _lib.profile_collect()
    print(n, {k: round(v[0] / 20, 4) for k, v in st.items() if v[1]})
