"""Stage-by-stage comparison of the HIP path with the oracle (developer tool, GPU box)."""
import ctypes, itertools, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch_nfft_amd import _lib
import torch_nfft_amd as tn
from oracle import nfft_ref, ndft

lib = _lib.load()
def rel(a, b): return float(np.linalg.norm((a-b).ravel())/max(np.linalg.norm(b.ravel()),1e-30))
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None

def run(d, N, m, n, B, Cr, seed=0):
    rng = np.random.default_rng(seed)
    pos = (rng.random((n, d)) - 0.5).astype(np.float32)
    batch = None
    if B > 1:
        batch = np.sort(rng.integers(0, B, n)).astype(np.int64); batch[0]=0; batch[-1]=B-1
    x = rng.standard_normal((n, Cr)).astype(np.float32)
    prob = _lib.Problem(d, n, Cr, B, N, m)
    plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
    post = torch.from_numpy(pos).cuda(); xt = torch.from_numpy(x).cuda()
    bt = torch.from_numpy(batch).cuda() if batch is not None else None
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(post), p(bt), p(plan), plan.numel(), s))
    M = 2*N
    grid = torch.full((B*Cr,)+(M,)*d, float('nan'), device='cuda')
    scratch = torch.empty(n*Cr + 128, device='cuda')
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(xt), Cr, p(grid), p(scratch), s))
    torch.cuda.synchronize()
    ref = nfft_ref.spread(x, pos, batch, N, m).real.reshape((B*Cr,)+(M,)*d)
    g = grid.cpu().numpy()
    print(f"d={d} N={N} m={m} n={n} B={B} Cr={Cr}: spread rel={rel(g, ref):.3e} sum got={g.sum():.6f} ref={ref.sum():.6f} nan={np.isnan(g).sum()}")
    ya = tn.nfft_adjoint(xt, post, bt, bandwidth=N, cutoff=m)
    r = nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)
    print(f"    adjoint rel={rel(ya.cpu().numpy(), r):.3e}")
    xh = torch.from_numpy(r.astype(np.complex64)).cuda()
    yf = tn.nfft_forward(xh, post, bt, cutoff=m)
    rf = nfft_ref.nfft_forward(r.astype(np.complex64), pos, batch, m=m)
    print(f"    forward rel={rel(yf.cpu().numpy(), rf):.3e}")
    yfr = tn.nfft_forward(xh, post, bt, cutoff=m, real_output=True)
    print(f"    forward(real) rel={rel(yfr.cpu().numpy(), rf.real):.3e}")

if __name__ == "__main__":
    for (d, N, m, n, B, Cr) in [(1,16,2,5,1,1),(1,64,4,300,1,1),(2,16,3,1,1,1),(2,16,3,200,1,1),(2,16,4,300,2,3),
                                (3,16,4,1,1,1),(3,16,4,300,1,1),(3,32,4,3000,2,2),(3,64,4,20000,1,1)]:
        run(d, N, m, n, B, Cr)
