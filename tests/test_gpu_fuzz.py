"""Seeded random configurations of the operators against the oracle (oracle/nfft_ref.py): dimension, bandwidth, cutoff,
point sets (some EMPTY, ragged sizes), columns, dtypes, real_output and the point distribution are drawn per case, so
that the dispatch seams -- fused 1-D kernels / LDS-tile kernels with point splits / matrix-core kernels with one workgroup
per range or the plan's work list, one or several planes per launch -- are crossed in combinations the hand-written cases
do not list.  Every case is reproducible from its seed."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import nfft_ref

pytestmark = pytest.mark.gpu

TOL = 2e-5  # the T1 of test_gpu_parity.py (fp32 accumulation of up to n terms, atomics reorder the sums)


@pytest.fixture(scope="module")
def tn():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import torch_nfft_amd
    return torch_nfft_amd


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def draw_case(seed):
    rng = np.random.default_rng(seed)
    d = int(rng.integers(1, 4))
    if d == 1:
        N = int(rng.choice([2, 6, 16, 50, 64, 256, 1000, 2048, 4096]))
        n = int(rng.integers(1, 5000))
    elif d == 2:
        N = int(rng.choice([4, 10, 16, 32, 48, 64, 128]))
        n = int(rng.integers(1, 4000))
    else:
        N = int(rng.choice([4, 8, 12, 16, 32, 32, 40]))  # 32, 40: grids of 64^3 / 80^3 = the matrix-core kernels
        n = int(rng.integers(1, 2500)) if N < 32 else int(rng.integers(500, 12000))
    m = min(int(rng.integers(1, 9)), N - 1)  # (the library rejects a window of 2m+2 taps on a grid of fewer cells)
    B = int(rng.choice([1, 1, 2, 3, 5]))
    cols = [(), (1,), (2,), (3,), (2, 2), (5,)][int(rng.integers(0, 6))]
    complex_x = bool(rng.integers(0, 2))
    dist = ["uniform", "clusters", "corner"][int(rng.integers(0, 3))]
    if dist == "uniform":
        pos = rng.random((n, d)) - 0.5
    elif dist == "clusters":  # two tight clusters: dense slab ranges, cut ranges, most work items empty
        centres = rng.random((2, d)) - 0.5
        pos = centres[rng.integers(0, 2, n)] + 0.01 * rng.standard_normal((n, d))
    else:  # everything within a few cells of the periodic corner
        pos = 0.5 + 0.02 * rng.standard_normal((n, d))
    pos = (pos - np.floor(pos + 0.5)).astype(np.float32)
    pos = np.clip(pos, -0.5, np.nextafter(np.float32(0.5), np.float32(0)))
    if B > 1:
        sizes = rng.multinomial(n, rng.dirichlet(np.ones(B)))
        if rng.integers(0, 2):
            sizes[int(rng.integers(0, B - 1))] = 0  # an empty point set (never the last: batch[-1] defines B)
        if sizes[-1] == 0:
            sizes[-1] = 1
        n = int(sizes.sum())
        pos = pos[:n] if n <= pos.shape[0] else np.concatenate([pos, pos[: n - pos.shape[0]]])
        batch = np.repeat(np.arange(B), sizes).astype(np.int64)
    else:
        batch = None
    x = rng.standard_normal((n,) + cols).astype(np.float32)
    if complex_x:
        x = (x + 1j * rng.standard_normal(x.shape)).astype(np.complex64)
    xh = rng.standard_normal((B,) + (N,) * d + cols).astype(np.float32)
    if bool(rng.integers(0, 2)):
        xh = (xh + 1j * rng.standard_normal(xh.shape)).astype(np.complex64)
    return dict(d=d, N=N, m=m, B=B, n=n, cols=cols, pos=pos, batch=batch, x=x, xh=xh,
                real_adj=bool(rng.integers(0, 2)), real_fwd=bool(rng.integers(0, 2)))


@pytest.mark.parametrize("seed", range(60))
def test_random_configuration_vs_oracle(tn, seed):
    c = draw_case(20260000 + seed)
    label = "d=%d N=%d m=%d B=%d n=%d cols=%s" % (c["d"], c["N"], c["m"], c["B"], c["n"], c["cols"])
    pos, batch = dev(c["pos"]), dev(c["batch"])
    ya = tn.nfft_adjoint(dev(c["x"]), pos, batch, bandwidth=c["N"], cutoff=c["m"], real_output=c["real_adj"])
    ref = nfft_ref.nfft_adjoint(c["x"], c["pos"], c["batch"], N=c["N"], m=c["m"], real_output=c["real_adj"])
    assert ya.shape == ref.shape, label
    assert rel_l2(host(ya), ref) < TOL, label
    yf = tn.nfft_forward(dev(c["xh"]), pos, batch, cutoff=c["m"], real_output=c["real_fwd"])
    ref = nfft_ref.nfft_forward(c["xh"], c["pos"], c["batch"], m=c["m"], real_output=c["real_fwd"])
    assert yf.shape == ref.shape, label
    assert rel_l2(host(yf), ref) < TOL, label
    from torch_nfft_amd import ops
    ops.check_status()
