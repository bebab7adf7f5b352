"""Developer tool: the small legs of bench.py (C1: 1-D N=64 m=2 n=1e3; C2: 2-D N=128 m=4 n=1e5) on their own, for kernel
traces (rocprofv3 --kernel-trace -- python3 scripts/small_legs.py c2) and host-side timing.
usage: python scripts/small_legs.py c1|c2|"d,N,m,n,B,C" [steps]   (NFFT_HIP_SMALL_GRID=0 in the environment: general path)"""
import sys, time
import torch
import torch_nfft_amd as tn
from torch_nfft_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(777)
B, C = 1, 1
if "," in which:
    d, N, m, n, B, C = (int(v) for v in which.split(","))
else:
    d, N, m, n = (1, 64, 2, 1000) if which == "c1" else (2, 128, 4, 100_000)
pos = torch.rand((n, d), generator=g, device=dev) - 0.5
x = torch.rand((n,) if C == 1 else (n, C), generator=g, device=dev)
batch = None if B == 1 else (torch.arange(n, device=dev) * B) // n

def step(fresh=True):
    if fresh:
        ops.plan_cache_clear()
    y = tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
    return tn.nfft_forward(y, pos, batch, cutoff=m, real_output=True)

for _ in range(5):
    step()
torch.cuda.synchronize()
for fresh in (True, False):
    # one at a time (as bench.py's legs) and back to back
    ts = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); step(fresh); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(fresh)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("%s fresh_plan=%s: one at a time median %.1f us (min %.1f); back to back %.1f us per step (host enqueue %.1f us)"
          % (which, fresh, ts[len(ts) // 2] * 1e6, ts[0] * 1e6, tot / steps * 1e6, host / steps * 1e6), flush=True)
