"""Developer tool: cost of the stage timers (HIP events around every stage) on the headline step and the small legs."""
import sys, time
import torch
import torch_nfft_amd as tn
from torch_nfft_amd import ops, _lib

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for name, d, N, m, n, steps in (("c3", 3, 256, 4, 10_000_000, 30), ("c2", 2, 128, 4, 100_000, 200), ("c1", 1, 64, 2, 1000, 200)):
    pos = torch.rand((n, d), generator=g, device=dev) - 0.5
    x = torch.rand((n,), generator=g, device=dev)
    def step():
        ops.plan_cache_clear()
        y = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
        return tn.nfft_forward(y, pos, None, cutoff=m, real_output=True)
    for _ in range(5): step()
    for rnd in range(2):
        for prof in (False, True):
            _lib.profile_enable(prof); _lib.profile_collect()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps): step()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
            ts = []
            for _ in range(steps):
                torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            ts.sort()
            _lib.profile_collect(); _lib.profile_enable(False)
            print("%s stage timers %s: back to back %.1f us per step, one at a time median %.1f us" % (name, "on " if prof else "off", dt * 1e6, ts[len(ts) // 2] * 1e6), flush=True)
