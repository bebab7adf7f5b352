import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_nfft_amd as tn
from oracle import nfft_ref
rng = np.random.default_rng(17)
for (n, N, B) in ((5000, 32, 2), (60000, 32, 2), (5000, 64, 1)):
    m = 4
    pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
    batch = np.sort(rng.integers(0, B, n)).astype(np.int64)
    batch[0], batch[-1] = 0, B - 1
    x = rng.standard_normal((n, 2)).astype(np.float32)
    pt, bt, xt = torch.from_numpy(pos).cuda(), torch.from_numpy(batch).cuda(), torch.from_numpy(x).cuda()
    y = tn.nfft_adjoint(xt, pt, bt, bandwidth=N, cutoff=m)
    ref = nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)
    e1 = np.linalg.norm(y.cpu().numpy() - ref) / np.linalg.norm(ref)
    f = tn.nfft_forward(y, pt, bt, cutoff=m, real_output=True)
    rf = nfft_ref.nfft_forward(y.cpu().numpy(), pos, batch, m=m, real_output=True)
    e2 = np.linalg.norm(f.cpu().numpy() - rf) / np.linalg.norm(rf)
    print("n=%d N=%d B=%d adjoint err %.3e forward err %.3e" % (n, N, B, e1, e2), flush=True)
