"""Developer tool: 1-D transforms on big grids -- where do library and oracle differ?"""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_nfft_amd as tn
from oracle import nfft_ref
for logN, n in ((17, 6000), (17, 1), (18, 1), (18, 6000)):
    rng = np.random.default_rng(2024)
    N, m = 1 << logN, 4
    pos = (rng.random((n, 1)) - 0.5).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    y = tn.nfft_adjoint(torch.from_numpy(x).cuda(), torch.from_numpy(pos).cuda(), None, bandwidth=N, cutoff=m).cpu().numpy()[0]
    ref = nfft_ref.nfft_adjoint(x, pos, None, N=N, m=m)[0]
    d = np.abs(y - ref)
    top = np.argsort(-d)[:6]
    print("N=2^%d n=%d: rel %.2e; max |diff| %.3e at k=%s (|ref| there %s); count(diff > 1e-3 max|ref|) = %d; ratio y/ref at top %s" % (
        logN, n, np.linalg.norm(y - ref) / np.linalg.norm(ref), d.max(), (top - N // 2).tolist(), np.round(np.abs(ref[top]), 3).tolist(),
        int((d > 1e-3 * np.abs(ref).max()).sum()), np.round(y[top] / ref[top], 4).tolist()), flush=True)
