"""Developer tool: host-side (enqueue) cost of the small legs, layer by layer: autograd wrapper -> torch operator ->
C entry point with caller-owned buffers.  usage: python scripts/host_cost.py c1|c2"""
import ctypes, sys, time
import torch
import torch_nfft_amd as tn
from torch_nfft_amd import ops, _lib

which = sys.argv[1] if len(sys.argv) > 1 else "c1"
d, N, m, n = (1, 64, 2, 1000) if which == "c1" else (2, 128, 4, 100_000)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
pos = torch.rand((n, d), generator=g, device=dev) - 0.5
x = torch.rand((n,), generator=g, device=dev)
yh = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)

def enqueue_cost(fn, iters=400, burst=10):
    """mean host time of fn() over bursts of `burst` calls into an EMPTY queue (synchronised between bursts)"""
    for _ in range(20): fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters // burst):
        t0 = time.perf_counter()
        for _ in range(burst): fn()
        tot += time.perf_counter() - t0
        torch.cuda.synchronize()
    return tot / iters * 1e6

lib = _lib.load()
Problem = _lib.Problem
vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
prob = Problem(d, n, 1, 1, N, m)
P = ctypes.byref(prob)
plan = torch.empty(lib.nfft_hip_plan_bytes(P), dtype=torch.uint8, device=dev)
wsa = torch.empty(lib.nfft_hip_adjoint_workspace_bytes(P, 0, 0) + 256, dtype=torch.uint8, device=dev)
wsf = torch.empty(lib.nfft_hip_forward_workspace_bytes(P, 1, 1) + 256, dtype=torch.uint8, device=dev)
yf = torch.empty(n, device=dev)
s = vp(torch.cuda.current_stream().cuda_stream)
p_ = lambda t: vp(t.data_ptr())
def c_plan(): assert lib.nfft_hip_plan_points(P, p_(pos), None, p_(plan), i64(plan.numel()), s) == 0
def c_adj(): assert lib.nfft_hip_adjoint_planned(P, p_(plan), p_(x), 0, 0, p_(yh), p_(wsa), i64(wsa.numel()), s) == 0
def c_fwd(): assert lib.nfft_hip_forward_planned(P, p_(plan), p_(yh), 1, 1, p_(yf), p_(wsf), i64(wsf.numel()), s) == 0
c_plan(); torch.cuda.synchronize()
rows = [
    ("python no-op", lambda: None),
    ("ctypes nfft_hip_plan_bytes (no GPU work)", lambda: lib.nfft_hip_plan_bytes(P)),
    ("C nfft_hip_plan_points", c_plan),
    ("C nfft_hip_adjoint_planned", c_adj),
    ("C nfft_hip_forward_planned", c_fwd),
    ("torch op nfft_adjoint, plan cached", lambda: ops.nfft_adjoint(pos, x, None, N, m, 0)),
    ("torch op nfft_forward, plan cached", lambda: ops.nfft_forward(pos, yh, None, m, 1)),
    ("autograd wrapper nfft_adjoint, plan cached", lambda: tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)),
    ("autograd wrapper nfft_forward, plan cached", lambda: tn.nfft_forward(yh, pos, None, cutoff=m, real_output=True)),
    ("plan_cache_clear + torch op nfft_adjoint", lambda: (ops.plan_cache_clear(), ops.nfft_adjoint(pos, x, None, N, m, 0))),
    ("torch.empty(1000)", lambda: torch.empty(1000, device=dev)),
    ("hipMemsetAsync via torch zero_ (1 KiB)", lambda: yf[:256].zero_()),
]
for name, fn in rows:
    print("%-50s %7.1f us" % (name, enqueue_cost(fn)), flush=True)
