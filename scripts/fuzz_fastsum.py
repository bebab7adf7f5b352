"""Developer tool: seeded random configurations of nfft_fastsum against the oracle (oracle/nfft_ref.py): dimension, N, m,
point sets, columns, real / complex coefficients and kernel coefficients, shared or distinct targets.
usage: python scripts/fuzz_fastsum.py <cases> [first seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch_nfft_amd as tn
from torch_nfft_amd import ops
from oracle import nfft_ref
rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0; worst = 0.0; t0 = time.time()
for seed in range(first, first + int(sys.argv[1])):
    rng = np.random.default_rng(seed)
    d = int(rng.integers(1, 4))
    N = int(rng.choice({1: [8, 32, 64, 100, 512], 2: [8, 16, 32, 48, 64], 3: [4, 8, 16, 32]}[d]))
    m = min(int(rng.integers(1, 8)), N - 1)
    B = int(rng.choice([1, 1, 2, 3]))
    C = int(rng.choice([1, 1, 2, 5]))
    ns = int(rng.integers(1, 1500)); shared = bool(rng.integers(0, 2)); nt = ns if shared else int(rng.integers(1, 1500))
    def pts(n):
        pos = ((rng.random((n, d)) - 0.5) * 0.5).astype(np.float32)   # radius-1/4 box (fastsum geometry)
        if B == 1: return pos, None
        b = np.sort(rng.integers(0, B, n)).astype(np.int64); b[-1] = B - 1; b[0] = 0 if n > 1 else B - 1
        return pos, np.sort(b)
    src, sb = pts(ns)
    tgt, tb = (src, sb) if shared else pts(nt)
    if B > 1 and (sb[-1] != B - 1 or tb[-1] != B - 1): continue
    x = rng.standard_normal((ns, C)).astype(np.float32)
    if rng.integers(0, 2): x = (x + 1j * rng.standard_normal((ns, C))).astype(np.complex64)
    co = rng.standard_normal((N,) * d).astype(np.float32)
    if rng.integers(0, 2): co = (co + 1j * rng.standard_normal((N,) * d)).astype(np.complex64)
    y = tn.nfft_fastsum(dev(x), dev(co), dev(src), None if shared else dev(tgt), dev(sb), None if shared else dev(tb), cutoff=m)
    ref = nfft_ref.nfft_fastsum(x, co, src, None if shared else tgt, sb, None if shared else tb, m=m)
    if not np.iscomplexobj(x): ref = ref.real
    e = rel(y.cpu().numpy(), ref)
    worst = max(worst, e)
    if not e < 2e-5:
        bad += 1
        print("FAIL seed", seed, "d=%d N=%d m=%d B=%d C=%d ns=%d nt=%d shared=%s" % (d, N, m, B, C, ns, nt, shared), e, flush=True)
    if time.time() - t0 > 400: print("time limit at seed", seed); break
ops.check_status()
print("cases", seed - first + 1, "failures", bad, "worst", worst)
