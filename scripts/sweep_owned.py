"""Developer tool: step time of the scatter (NFFT_HIP_OWNED=0) and owner-computes (=1) spreading variants over point counts
(3-D, N = 256, m = 4): where is the crossover?  Run once per setting (the switch is read once per process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_nfft_amd as tn
N, m = 256, 4
gen = torch.Generator(device="cuda").manual_seed(3)
for n in [1_000_000, 2_000_000, 3_000_000, 4_000_000, 5_000_000]:
    pos = torch.rand((n, 3), generator=gen, device="cuda") - 0.5
    x = torch.rand((n,), generator=gen, device="cuda")
    for _ in range(2):
        y = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
        f = tn.nfft_forward(y, pos, None, cutoff=m, real_output=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        tn.ops.plan_cache_clear()
        t0 = time.perf_counter()
        y = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
        f = tn.nfft_forward(y, pos, None, cutoff=m, real_output=True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("OWNED=%s n=%8d: %7.3f ms/step" % (os.environ.get("NFFT_HIP_OWNED", "auto"), n, ts[3]), flush=True)
