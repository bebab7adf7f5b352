"""GPU parity tests: the HIP path (through the C ABI, via torch_nfft_amd) against the oracle.

Tolerances (fp32, stated here as the contract):
  T1  HIP vs the float64 restatement of the reference's algorithm (oracle/nfft_ref.py):
      relative L2 <= 2e-5  (fp32 accumulation of up to n terms; atomics reorder the sums)
  T2  HIP vs the exact NDFT (golden vectors frozen from the reference's torch_nfft/ndft.py, or
      oracle/ndft.py): relative L2 <= 2e-2 (m=2), 3e-3 (m=3), 5e-4 (m=4), 2e-5 (m=8)
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import ndft, nfft_ref

pytestmark = pytest.mark.gpu

T1 = 2e-5
T1N = 2e-6  # narrow-tiling kernels on the sweep sizes: they accumulate in 8-byte LDS cells, observed ~1e-7
T1W = 2e-6  # matrix-core kernels (3-D grids of 64^3 and up, m <= 7): ~22-bit operands, observed 2e-7; a slip in the f16
            # split (dropping a term costs ~5e-4, a mis-rounded hi part ~1e-5) must not pass
T2 = {1: 2e-1, 2: 2e-2, 3: 3e-3, 4: 5e-4, 5: 1e-4, 6: 5e-5, 7: 3e-5, 8: 2e-5}


@pytest.fixture(scope="module")
def tn():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import torch_nfft_amd
    return torch_nfft_amd


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- golden vectors

def test_golden_g1_adjoint_2d_batched(tn):
    g = load_golden("g1_adjoint_2d_batched")
    y = tn.nfft_adjoint(dev(g["x"]), dev(g["pos"]), dev(g["batch"]), bandwidth=16, cutoff=4)
    assert y.shape == (3, 16, 16, 10) and y.dtype == torch.complex64
    assert rel_l2(host(y), g["y_adjoint"]) < T2[4]
    ref = nfft_ref.nfft_adjoint(g["x"], g["pos"], g["batch"], N=16, m=4)
    assert rel_l2(host(y), ref) < T1


def test_golden_g2_forward_2d(tn):
    g = load_golden("g2_forward_2d")
    y = tn.nfft_forward(dev(g["x"]), dev(g["pos"]), None, cutoff=4)
    assert y.shape == (10, 1) and y.dtype == torch.complex64
    assert rel_l2(host(y), g["y_forward"]) < T2[4]
    assert rel_l2(host(y), nfft_ref.nfft_forward(g["x"], g["pos"], None, m=4)) < T1


@pytest.mark.parametrize("m", [2, 4, 8])
def test_golden_g3_1d(tn, m):
    g = load_golden("g3_1d_n64")
    ya = tn.nfft_adjoint(dev(g["x"]), dev(g["pos"]), None, bandwidth=64, cutoff=m)
    assert ya.shape == (1, 64)
    assert rel_l2(host(ya), g["y_adjoint"]) < T2[m]
    assert rel_l2(host(ya), nfft_ref.nfft_adjoint(g["x"], g["pos"], None, N=64, m=m)) < T1
    yf = tn.nfft_forward(dev(g["xhat"]), dev(g["pos"]), None, cutoff=m)
    assert yf.shape == (1000,)
    assert rel_l2(host(yf), g["y_forward"]) < T2[m]
    assert rel_l2(host(yf), nfft_ref.nfft_forward(g["xhat"], g["pos"], None, m=m)) < T1


def test_golden_g4_3d_ragged(tn):
    g = load_golden("g4_3d_ragged")
    pos, batch = dev(g["pos"]), dev(g["batch"])
    for key in ("real", "complex"):
        y = tn.nfft_adjoint(dev(g["x_" + key]), pos, batch, bandwidth=16, cutoff=4)
        assert y.shape == (3, 16, 16, 16, 2)
        assert rel_l2(host(y), g["y_adjoint_" + key]) < T2[4]
        assert rel_l2(host(y), nfft_ref.nfft_adjoint(g["x_" + key], g["pos"], g["batch"], N=16, m=4)) < T1
    yf = tn.nfft_forward(dev(g["xhat"]), pos, batch, cutoff=4)
    assert yf.shape == (200, 2)
    assert rel_l2(host(yf), g["y_forward"]) < T2[4]
    assert rel_l2(host(yf), nfft_ref.nfft_forward(g["xhat"], g["pos"], g["batch"], m=4)) < T1


def test_golden_g5_grad_shapes(tn):
    g = load_golden("g5_grad_shapes")
    pos, batch = dev(g["pos"]), dev(g["batch"])
    ya = tn.nfft_adjoint(dev(g["x"]), pos, batch, bandwidth=16, cutoff=3)
    assert rel_l2(host(ya), g["y_adjoint"]) < T2[3]
    yf = tn.nfft_forward(dev(g["xhat"]), pos, batch, cutoff=3)
    assert rel_l2(host(yf), g["y_forward"]) < T2[3]


# ----------------------------------------------------------------------------- oracle sweeps

def _random_problem(rng, d, n, B, cols, complex_x):
    pos = (rng.random((n, d)) - 0.5).astype(np.float32)
    if B > 1:
        batch = np.sort(rng.integers(0, B, n)).astype(np.int64)
        batch[0], batch[-1] = 0, B - 1
    else:
        batch = None
    shape = (n,) + cols
    x = rng.standard_normal(shape).astype(np.float32)
    if complex_x:
        x = (x + 1j * rng.standard_normal(shape)).astype(np.complex64)
    return pos, batch, x


@pytest.mark.parametrize("d,N,m", [(1, 64, 2), (1, 32, 3), (1, 512, 4), (2, 16, 3), (2, 32, 4), (2, 64, 2),
                                   (3, 16, 4), (3, 16, 2), (3, 24, 3), (3, 32, 4), (2, 16, 6), (3, 16, 5),
                                   (1, 64, 8), (2, 32, 8), (3, 20, 7), (2, 8, 1),
                                   (1, 48, 3), (1, 100, 4), (1, 4096, 2),  # (1-D outside the fused path: smallgrid.hip)
                                   # large grids that are not a power of two: the cell / fraction split of a coordinate
                                   # must stay exact there (a contracted multiply once cost 6e-9 M in relative error)
                                   (1, 1000, 4), (1, 3000, 8), (2, 100, 4), (3, 40, 3), (3, 48, 4)])  # (96^3: matrix cores)
@pytest.mark.parametrize("complex_x", [False, True])
def test_adjoint_and_forward_vs_oracle(tn, d, N, m, complex_x):
    rng = np.random.default_rng(1000 * d + N + m)
    n, B, cols = 700, 3, (2,)
    pos, batch, x = _random_problem(rng, d, n, B, cols, complex_x)
    ya = tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=N, cutoff=m)
    assert ya.shape == (B,) + (N,) * d + cols and ya.dtype == torch.complex64
    ref = nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)
    assert rel_l2(host(ya), ref) < T1N
    assert rel_l2(host(ya), ndft.ndft_adjoint(x, pos, batch, N=N)) < T2[m]
    # forward of a random spectrum
    xh = rng.standard_normal((B,) + (N,) * d + cols).astype(np.float32)
    if complex_x:
        xh = (xh + 1j * rng.standard_normal(xh.shape)).astype(np.complex64)
    yf = tn.nfft_forward(dev(xh), dev(pos), dev(batch), cutoff=m)
    assert yf.shape == (n,) + cols and yf.dtype == torch.complex64
    assert rel_l2(host(yf), nfft_ref.nfft_forward(xh, pos, batch, m=m)) < T1N
    assert rel_l2(host(yf), ndft.ndft_forward(xh, pos, batch)) < T2[m]


@pytest.mark.parametrize("N,m", [(8, 2), (64, 4), (64, 2)])
def test_column_fft_sizes_3d(tn, N, m):
    """Grids of 16^3 and 128^3 exercise the radix sequences (4,4) and (8,4,4) of the fused column passes (the other
    tests cover 32, 64, 256 and 512; 1024 is in test_gpu_large.py); 128^3 also takes the matrix-core kernels."""
    rng = np.random.default_rng(300 + N + m)
    n = 3000
    pos, batch, x = _random_problem(rng, 3, n, 1, (), True)
    tol = T1W if N >= 32 else T1
    ya = tn.nfft_adjoint(dev(x), dev(pos), None, bandwidth=N, cutoff=m)
    assert rel_l2(host(ya), nfft_ref.nfft_adjoint(x, pos, None, N=N, m=m)) < tol
    xh = (rng.standard_normal((1, N, N, N)) + 1j * rng.standard_normal((1, N, N, N))).astype(np.complex64)
    yf = tn.nfft_forward(dev(xh), dev(pos), None, cutoff=m)
    assert rel_l2(host(yf), nfft_ref.nfft_forward(xh, pos, None, m=m)) < tol


@pytest.mark.parametrize("N,m,cols,complex_x", [(64, 4, (), False), (64, 2, (3,), True), (128, 4, (), True), (128, 3, (2,), False),
                                               (256, 4, (), False), (512, 2, (), True)])
def test_column_fft_sizes_2d(tn, N, m, cols, complex_x):
    """2-D grids of 128^2 ... 1024^2 (round 4): own row pass + ONE pruned column pass that carries the roll-off instead of
    rocFFT's 2-D transform and the roll-off kernel (column tiles of 4 ... 16 spectrum columns, by the plane count) -- both directions, both real_output settings, one and several columns (planar passes + layout
    transpose), four point sets, against the float64 algorithm restatement."""
    rng = np.random.default_rng(2000 + N + m)
    n, B = 2500, 4
    pos, batch, x = _random_problem(rng, 2, n, B, cols, complex_x)
    for real_output in (False, True):
        ya = tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=N, cutoff=m, real_output=real_output)
        assert ya.shape == (B, N, N) + cols and ya.dtype == (torch.float32 if real_output else torch.complex64)
        assert rel_l2(host(ya), nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m, real_output=real_output)) < T1N
    xh = rng.standard_normal((B, N, N) + cols).astype(np.float32)
    if complex_x:
        xh = (xh + 1j * rng.standard_normal(xh.shape)).astype(np.complex64)
    for real_output in (False, True):
        yf = tn.nfft_forward(dev(xh), dev(pos), dev(batch), cutoff=m, real_output=real_output)
        assert yf.shape == (n,) + cols and yf.dtype == (torch.float32 if real_output else torch.complex64)
        assert rel_l2(host(yf), nfft_ref.nfft_forward(xh, pos, batch, m=m, real_output=real_output)) < T1N


@pytest.mark.parametrize("d", [1, 2, 3])
def test_real_output_variants(tn, d):
    rng = np.random.default_rng(77 + d)
    N, m, n = 16, 4, 300
    pos, batch, x = _random_problem(rng, d, n, 2, (3,), True)
    ya = tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=N, cutoff=m, real_output=True)
    assert ya.dtype == torch.float32
    assert rel_l2(host(ya), nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m, real_output=True)) < T1
    xh = (rng.standard_normal((2,) + (N,) * d + (3,)) + 1j * rng.standard_normal((2,) + (N,) * d + (3,))).astype(np.complex64)
    yf = tn.nfft_forward(dev(xh), dev(pos), dev(batch), cutoff=m, real_output=True)
    assert yf.dtype == torch.float32 and yf.shape == (n, 3)
    assert rel_l2(host(yf), nfft_ref.nfft_forward(xh, pos, batch, m=m, real_output=True)) < T1


def test_trailing_shapes_and_no_columns(tn):
    rng = np.random.default_rng(5)
    pos, batch, x = _random_problem(rng, 2, 200, 1, (2, 3), False)
    ya = tn.nfft_adjoint(dev(x), dev(pos), None, bandwidth=16, cutoff=3)
    assert ya.shape == (1, 16, 16, 2, 3)
    assert rel_l2(host(ya), nfft_ref.nfft_adjoint(x, pos, None, N=16, m=3)) < T1
    x1 = x[:, 0, 0].copy()  # 1-D x: no trailing dimension
    y1 = tn.nfft_adjoint(dev(x1), dev(pos), None, bandwidth=16, cutoff=3)
    assert y1.shape == (1, 16, 16)
    assert rel_l2(host(y1), host(ya)[..., 0, 0]) < 1e-5
    yf = tn.nfft_forward(ya, dev(pos), None, cutoff=3)
    assert yf.shape == (200, 2, 3)
    assert rel_l2(host(yf), nfft_ref.nfft_forward(host(ya), pos, None, m=3)) < T1


def test_edge_points_periodic_wrap(tn):
    # torus boundary, exact grid nodes, the largest float below 1/2, and a point outside [-1/2,1/2)
    pos = np.array([[-0.5, -0.5, -0.5], [0.49999997, 0.49999997, 0.49999997], [0.0, 0.0, 0.0],
                    [-0.5, 0.25, 0.125], [0.25, -0.5, 0.49999997], [0.46875, -0.46875, 0.0],
                    [0.75, -0.75, 1.25]], np.float32)
    rng = np.random.default_rng(11)
    x = rng.standard_normal((7, 1)).astype(np.float32)
    ya = tn.nfft_adjoint(dev(x), dev(pos), None, bandwidth=16, cutoff=4)
    assert rel_l2(host(ya), ndft.ndft_adjoint(x, pos, None, N=16)) < T2[4]
    yf = tn.nfft_forward(ya, dev(pos), None, cutoff=4)
    assert rel_l2(host(yf), ndft.ndft_forward(host(ya), pos, None)) < T2[4]


@pytest.mark.parametrize("N,m", [(32, 4), (32, 2), (40, 7), (64, 5)])
def test_edge_cases_wide_tiling(tn, N, m):
    """The same edge cases on grids that take the matrix-core kernels (3-D, 2N >= 64, m <= 7): torus boundary, exact
    grid nodes, out-of-range points, a single point, empty point sets between populated ones, every point in one cell,
    several columns, complex coefficients."""
    rng = np.random.default_rng(900 + N + m)
    edge = np.array([[-0.5, -0.5, -0.5], [0.49999997, 0.49999997, 0.49999997], [0.0, 0.0, 0.0],
                     [-0.5, 0.25, 0.125], [0.25, -0.5, 0.49999997], [0.46875, -0.46875, 0.0],
                     [0.75, -0.75, 1.25]], np.float32)
    same = np.tile(np.array([[0.1234, -0.3456, 0.4999]], np.float32), (40, 1))  # 40 points in one cell
    rnd = (rng.random((300, 3)) - 0.5).astype(np.float32)
    pos = np.concatenate([edge, same, rnd]).astype(np.float32)
    n = pos.shape[0]
    # point sets 0 and 3 are populated, 1, 2 and 4 are empty (batch_size = 5 comes from the last index + 1 ... so the
    # last set has one point)
    batch = np.zeros(n, np.int64)
    batch[n // 2:] = 3
    batch[-1] = 4
    x = (rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))).astype(np.complex64)
    ya = tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=N, cutoff=m)
    assert ya.shape == (5, N, N, N, 2)
    assert float(ya[1].abs().max()) == 0.0 and float(ya[2].abs().max()) == 0.0
    assert rel_l2(host(ya), nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)) < T1W
    yf = tn.nfft_forward(ya, dev(pos), dev(batch), cutoff=m)
    assert rel_l2(host(yf), nfft_ref.nfft_forward(host(ya), pos, batch, m=m)) < T1W
    # a single point
    y1 = tn.nfft_adjoint(dev(np.ones((1,), np.float32)), dev(pos[:1]), None, bandwidth=N, cutoff=m)
    assert rel_l2(host(y1), nfft_ref.nfft_adjoint(np.ones((1,), np.float32), pos[:1], None, N=N, m=m)) < T1W


def test_many_small_point_sets_wide_tiling(tn):
    """1 500 point sets of a few points each on a 64^3 grid: more first-level sort bins than fit the LDS histogram, so
    the plan takes the one-level (global-atomic) binning path -- its work list is built from the one-level bins -- under the matrix-core
    kernels; checked on a sample of the point sets against the oracle."""
    rng = np.random.default_rng(123)
    B, N, m = 1500, 32, 3
    counts = rng.integers(0, 9, size=B)
    counts[-1] = 3  # batch_size comes from the last index
    batch = np.repeat(np.arange(B), counts).astype(np.int64)
    n = batch.shape[0]
    pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    y = tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=N, cutoff=m)
    assert y.shape == (B, N, N, N)
    xh = (rng.standard_normal((B, N, N, N)) + 1j * rng.standard_normal((B, N, N, N))).astype(np.complex64)
    yf = tn.nfft_forward(dev(xh), dev(pos), dev(batch), cutoff=m)
    yh, yfh = host(y), host(yf)
    for b in (0, 1, 2, 700, 1498, 1499):
        sel = batch == b
        if sel.sum() == 0:
            assert float(np.abs(yh[b]).max()) == 0.0
            continue
        ref = nfft_ref.nfft_adjoint(x[sel], pos[sel], None, N=N, m=m)[0]
        assert rel_l2(yh[b], ref) < T1
        reff = nfft_ref.nfft_forward(xh[b:b + 1], pos[sel], None, m=m)
        assert rel_l2(yfh[sel], reff) < T1


def test_empty_and_tiny_inputs(tn):
    pos = torch.zeros((0, 2), dtype=torch.float32, device="cuda")
    x = torch.zeros((0, 3), dtype=torch.float32, device="cuda")
    y = tn.nfft_adjoint(x, pos, None, bandwidth=8, cutoff=2)
    assert y.shape == (1, 8, 8, 3) and float(y.abs().max()) == 0.0
    xh = torch.randn((1, 8, 8, 3), dtype=torch.complex64, device="cuda")
    assert tn.nfft_forward(xh, pos, None, cutoff=2).shape == (0, 3)
    # one point, empty point sets in the middle of the batch
    pos1 = np.array([[0.1, -0.2]], np.float32)
    ya = tn.nfft_adjoint(dev(np.ones((1,), np.float32)), dev(pos1), dev(np.array([2], np.int64)), bandwidth=8, cutoff=3)
    assert ya.shape == (3, 8, 8)
    assert float(ya[:2].abs().max()) == 0.0
    assert rel_l2(host(ya[2:]), ndft.ndft_adjoint(np.ones((1,)), pos1, None, N=8)) < T2[3]


def test_non_contiguous_inputs(tn):
    rng = np.random.default_rng(21)
    pos, _, x = _random_problem(rng, 2, 150, 1, (4,), False)
    xd = dev(x)[:, ::2]  # non-contiguous view
    posd = dev(np.concatenate([pos, pos], 1))[:, :2]
    ya = tn.nfft_adjoint(xd, posd, None, bandwidth=16, cutoff=4)
    assert rel_l2(host(ya), nfft_ref.nfft_adjoint(x[:, ::2], pos, None, N=16, m=4)) < T1


def test_input_checks(tn):
    pos = torch.zeros((4, 2), device="cuda")
    x = torch.zeros((4,), device="cuda")
    with pytest.raises(RuntimeError, match="Input mismatch"):
        tn.nfft_adjoint(torch.zeros((5,), device="cuda"), pos)
    with pytest.raises(RuntimeError, match="Input mismatch"):
        tn.nfft_adjoint(x, torch.zeros((4, 4), device="cuda"))
    with pytest.raises(RuntimeError, match="Input mismatch"):
        tn.nfft_adjoint(x.double(), pos)
    with pytest.raises(RuntimeError, match="Input mismatch"):
        tn.nfft_adjoint(x, pos, cutoff=9)
    with pytest.raises(RuntimeError, match="Input mismatch"):
        tn.nfft_forward(torch.zeros((2, 8, 8), device="cuda"), pos)  # batch_size mismatch
    with pytest.raises(RuntimeError, match="Input mismatch"):
        tn.nfft_forward(torch.zeros((1, 8, 6), device="cuda"), pos)
    with pytest.raises(RuntimeError, match="only implemented for GPU tensors"):
        tn.nfft_adjoint(x.cpu(), pos.cpu())


def test_operator_namespace(tn):
    """torch.ops.torch_nfft.* keeps the reference's positional schema (csrc/core.cpp:176-184)."""
    rng = np.random.default_rng(31)
    pos, batch, x = _random_problem(rng, 2, 100, 2, (), False)
    y1 = torch.ops.torch_nfft.nfft_adjoint(dev(pos), dev(x), dev(batch), 16, 3, 0)
    y2 = tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=16, cutoff=3)
    assert rel_l2(host(y1), host(y2)) < 1e-5
    z1 = torch.ops.torch_nfft.nfft_forward(dev(pos), y1, dev(batch), 3, 1)
    assert z1.dtype == torch.float32 and z1.shape == (100,)


# ----------------------------------------------------------------------------- autograd (test_grad.py scenarios)

@pytest.mark.parametrize("d", [1, 2, 3])
def test_autograd_adjointness(tn, d):
    """The reference's test_grad.py compares backward() with finite differences; for a linear operator that is
    the statement <A x, y> = <x, A^H y>, checked here exactly (up to fp32) through autograd."""
    rng = np.random.default_rng(41 + d)
    N, m, n, B = 16, 3, 40, 2
    pos, batch, x = _random_problem(rng, d, n, B, (3,), False)
    xt = dev(x).requires_grad_(True)
    ya = tn.nfft_adjoint(xt, dev(pos), dev(batch), bandwidth=N, cutoff=m)
    w = torch.randn_like(ya)
    loss = (ya * w.conj()).real.sum()
    loss.backward()
    # d/dx Re<A x, w> = Re(A^H w) where A^H = forward transform
    expect = tn.nfft_forward(w, dev(pos), dev(batch), cutoff=m, real_output=True)
    assert rel_l2(host(xt.grad), host(expect)) < 1e-5
    assert rel_l2(host(xt.grad), nfft_ref.nfft_forward(host(w), pos, batch, m=m, real_output=True)) < T1
    # forward op: gradient is the adjoint transform
    xh = torch.randn((B,) + (N,) * d + (3,), dtype=torch.complex64, device="cuda", requires_grad=True)
    yf = tn.nfft_forward(xh, dev(pos), dev(batch), cutoff=m)
    v = torch.randn_like(yf)
    (yf * v.conj()).real.sum().backward()
    ref = nfft_ref.nfft_adjoint(host(v), pos, batch, N=N, m=m)
    assert rel_l2(host(xh.grad), ref) < T1
    # finite-difference spot check as in test_grad.py:25-46 (|op(x)|.sum())
    x0 = dev(x)
    f = lambda t: tn.nfft_adjoint(t, dev(pos), dev(batch), bandwidth=N, cutoff=m).abs().sum()
    x0.requires_grad_(True)
    f(x0).backward()
    eps = 1e-2
    i, c = 3, 1
    xp = x0.detach().clone(); xp[i, c] += eps
    xm = x0.detach().clone(); xm[i, c] -= eps
    fd = (f(xp) - f(xm)).item() / (2 * eps)
    assert abs(fd - x0.grad[i, c].item()) <= 2e-2 * max(1.0, abs(fd))


# ----------------------------------------------------------------------------- stage level + chunking

def test_spread_stage_matches_oracle(tn):
    """nfft_hip_plan_points + nfft_hip_spread against the oracle's gridding (real planes)."""
    import ctypes
    from torch_nfft_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(51)
    d, N, m, n, B, Cr = 3, 16, 4, 500, 2, 2
    pos, batch, x = _random_problem(rng, d, n, B, (Cr,), False)
    prob = _lib.Problem(d, n, Cr, B, N, m)
    plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
    post, batcht, xt = dev(pos), dev(batch), dev(x)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(post), p(batcht), p(plan), plan.numel(), s))
    M = 2 * N
    grid = torch.full((B * Cr,) + (M,) * d, float("nan"), device="cuda")
    scratch = torch.empty(lib.nfft_hip_spread_scratch_bytes(ctypes.byref(prob), Cr) // 4, device="cuda")
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(xt), Cr, p(grid), p(scratch), s))
    ref = nfft_ref.spread(x, pos, batch, N, m).real.reshape((B * Cr,) + (M,) * d)
    assert rel_l2(host(grid), ref) < T1
    # gather back: interpolation of the oracle grid at the same points
    yr = torch.empty((n, Cr), device="cuda")
    gt = dev(ref.astype(np.float32))
    _lib.check(lib.nfft_hip_interpolate(ctypes.byref(prob), p(plan), p(gt), Cr, p(yr), s))
    shift, psi = nfft_ref.window_taps(pos, N, m)
    exp = np.zeros((n, Cr))
    import itertools
    for ls in itertools.product(range(2 * m + 2), repeat=d):
        w = np.ones(n)
        idx = []
        for a, l in enumerate(ls):
            w = w * psi[:, a, l]
            idx.append((shift[:, a] + l + M) % M)
        for c in range(Cr):
            exp[:, c] += w * ref[(batch * Cr + c,) + tuple(idx)]
    assert rel_l2(host(yr), exp) < T1


def test_chunked_planes_path(tn, monkeypatch):
    """Force the (batch, column) plane chunking used for grids that exceed the chunk budget."""
    rng = np.random.default_rng(61)
    d, N, m = 2, 16, 3
    pos, batch, x = _random_problem(rng, d, 300, 2, (3,), True)
    # budget for ~2 planes of (real grid + half spectrum)
    monkeypatch.setenv("NFFT_HIP_CHUNK_BYTES", str(2 * (32 * 32 * 4 + 32 * 17 * 8) + 8))
    from torch_nfft_amd import ops
    ya = tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=N, cutoff=m)
    assert rel_l2(host(ya), nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)) < T1
    yf = tn.nfft_forward(ya, dev(pos), dev(batch), cutoff=m)
    assert rel_l2(host(yf), nfft_ref.nfft_forward(host(ya), pos, batch, m=m)) < T1


def test_full_rocfft_path_when_column_passes_disabled(tn, monkeypatch):
    """3-D power-of-two sizes normally take the pruned column passes (colfft.hip); the full rocFFT R2C/C2R +
    separate roll-off path (used for other sizes) must give the same answer on the same input."""
    rng = np.random.default_rng(81)
    pos, batch, x = _random_problem(rng, 3, 400, 2, (2,), True)
    from torch_nfft_amd import ops
    ya = tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=16, cutoff=4)
    yf = tn.nfft_forward(ya, dev(pos), dev(batch), cutoff=4)
    monkeypatch.setenv("NFFT_HIP_NO_COLFFT", "1")
    ya2 = tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=16, cutoff=4)
    yf2 = tn.nfft_forward(ya, dev(pos), dev(batch), cutoff=4)
    assert rel_l2(host(ya2), host(ya)) < 2e-6
    assert rel_l2(host(yf2), host(yf)) < 2e-6
    assert rel_l2(host(ya2), nfft_ref.nfft_adjoint(x, pos, batch, N=16, m=4)) < T1


def test_plan_cache_follows_in_place_updates(tn):
    """The host keeps one point plan keyed on tensor identity + version; editing pos in place must re-plan.  (A 128^2 grid:
    smaller ones run without a plan, smallgrid.hip.)"""
    from torch_nfft_amd import ops
    rng = np.random.default_rng(91)
    pos, _, x = _random_problem(rng, 2, 300, 1, (), False)
    post, xt = dev(pos), dev(x)
    ops.plan_cache_clear()
    y1 = tn.nfft_adjoint(xt, post, None, bandwidth=64, cutoff=3)
    h0 = ops.plan_cache_stats()["hits"]
    y1b = tn.nfft_adjoint(xt, post, None, bandwidth=64, cutoff=3)
    assert ops.plan_cache_stats()["hits"] == h0 + 1
    assert rel_l2(host(y1b), host(y1)) < 1e-6
    post.mul_(0.5)  # in-place: version counter changes
    y2 = tn.nfft_adjoint(xt, post, None, bandwidth=64, cutoff=3)
    assert rel_l2(host(y2), nfft_ref.nfft_adjoint(x, pos * 0.5, None, N=64, m=3)) < T1
    ops.plan_cache_enabled(False)
    y3 = tn.nfft_adjoint(xt, post, None, bandwidth=64, cutoff=3)
    ops.plan_cache_enabled(True)
    assert rel_l2(host(y3), host(y2)) < 1e-6


def test_register_tile_spreading_mode_is_deterministic():
    """NFFT_HIP_SPREAD=reg (read once per process) selects the atomics-free register-tile spreading kernel:
    same numbers as the oracle, and bitwise identical results run to run."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
import torch_nfft_amd as tn
from oracle import nfft_ref
rng = np.random.default_rng(3)
n, N, m = 3000, 32, 4
pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
pos[:500] = (0.02 * rng.standard_normal((500, 3)) + 0.4999).astype(np.float32)   # cluster on the periodic corner
batch = np.sort(rng.integers(0, 2, n)).astype(np.int64); batch[0], batch[-1] = 0, 1
x = rng.standard_normal((n, 2)).astype(np.float32)
args = [torch.from_numpy(a).cuda() for a in (x, pos, batch)]
y1 = tn.nfft_adjoint(*args, bandwidth=N, cutoff=m)
y2 = tn.nfft_adjoint(*args, bandwidth=N, cutoff=m)
ref = nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)
err = np.linalg.norm(y1.cpu().numpy() - ref) / np.linalg.norm(ref)
print("RESULT", err, bool(torch.equal(y1, y2)))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NFFT_HIP_SPREAD="reg")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) < T1 and line[2] == "True"


def test_package_import_before_torch():
    """Importing the package before torch must not bind the library to a second HIP runtime (every call then failed
    with "hipGetDevice failed"): a fresh interpreter imports torch_nfft_amd first and runs a transform."""
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, %r)
import torch_nfft_amd as tn
import torch
pos = torch.rand((500, 3), device="cuda") - 0.5
x = torch.rand((500,), device="cuda")
y = tn.nfft_adjoint(x, pos, None, bandwidth=16, cutoff=3)
print("RESULT", tuple(y.shape), bool(torch.isfinite(y.abs()).all()))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RESULT (1, 16, 16, 16) True" in out.stdout


@pytest.mark.parametrize("env_extra", [{"NFFT_HIP_SPREAD": "lds"},
                                       {"NFFT_HIP_GATHER": "lds", "NFFT_HIP_ROCFFT_ROWS": "1"},
                                       {"NFFT_HIP_COL_XCD": "0"}])
def test_fallback_kernels_match_oracle(env_extra):
    """The kernels the defaults replaced stay selectable and correct: NFFT_HIP_SPREAD=lds (f64-LDS-atomic spreading,
    narrow pencil tiling), NFFT_HIP_GATHER=lds (lane-per-point interpolation on the wide tiling),
    NFFT_HIP_ROCFFT_ROWS=1 (rocFFT instead of the own row passes) and NFFT_HIP_COL_XCD=0 (column passes in plain
    workgroup order) on a 128^3 grid: adjoint and forward vs the oracle."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
import torch_nfft_amd as tn
from oracle import nfft_ref
rng = np.random.default_rng(4)
n, N, m = 4000, 64, 4
pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
pos[:500] = (0.02 * rng.standard_normal((500, 3)) + 0.4999).astype(np.float32)
x = rng.standard_normal((n, 2)).astype(np.float32)
xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(pos).cuda()
y = tn.nfft_adjoint(xt, pt, None, bandwidth=N, cutoff=m)
ref = nfft_ref.nfft_adjoint(x, pos, None, N=N, m=m)
e1 = np.linalg.norm(y.cpu().numpy() - ref) / np.linalg.norm(ref)
f = tn.nfft_forward(y, pt, None, cutoff=m)
reff = nfft_ref.nfft_forward(y.cpu().numpy(), pos, None, m=m)
e2 = np.linalg.norm(f.cpu().numpy() - reff) / np.linalg.norm(reff)
print("RESULT", e1, e2)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) < T1 and float(line[2]) < T1


@pytest.mark.parametrize("env_extra", [{}, {"NFFT_HIP_SMALL_NARROW": "0"}, {"NFFT_HIP_OWNED_PAIR": "0"}, {"NFFT_HIP_COLFFT_2D": "0"}],
                         ids=["defaults", "wide-64^3", "unpaired-owned", "rocfft-2d"])
def test_round4_selection_switches_match_oracle(env_extra):
    """What round 4 made the default, and the path each switch brings back, against the oracle: the 64^3 grid (N = 32) with a
    narrow window (narrow tiling by default / NFFT_HIP_SMALL_NARROW=0: the matrix-core kernels on 3 x 2 pencils), a sparse
    128^3 problem with three columns (paired owner-computes spreading / NFFT_HIP_OWNED_PAIR=0: one sweep per column), a 2-D
    256^2 grid with two point sets (own row + column passes / NFFT_HIP_COLFFT_2D=0: rocFFT + the roll-off kernel)."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
import torch_nfft_amd as tn
from torch_nfft_amd import ops
from oracle import nfft_ref
rng = np.random.default_rng(44)
worst = 0.0
for d, N, m, n, B, C in ((3, 32, 3, 6000, 2, 1), (3, 32, 2, 3000, 1, 2), (3, 64, 4, 5000, 2, 3), (2, 128, 4, 8000, 2, 1)):
    pos = (rng.random((n, d)) - 0.5).astype(np.float32)
    batch = np.sort(rng.integers(0, B, n)).astype(np.int64); batch[0], batch[-1] = 0, B - 1
    x = rng.standard_normal((n, C)).astype(np.float32)
    pt, bt = torch.from_numpy(pos).cuda(), torch.from_numpy(batch).cuda()
    y = tn.nfft_adjoint(torch.from_numpy(x).cuda(), pt, bt, bandwidth=N, cutoff=m)
    ref = nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)
    worst = max(worst, np.linalg.norm(y.cpu().numpy() - ref) / np.linalg.norm(ref))
    f = tn.nfft_forward(y, pt, bt, cutoff=m)
    reff = nfft_ref.nfft_forward(y.cpu().numpy(), pos, batch, m=m)
    worst = max(worst, np.linalg.norm(f.cpu().numpy() - reff) / np.linalg.norm(reff))
ops.check_status()
print("RESULT", worst)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) < T1N, out.stdout


@pytest.mark.parametrize("env_extra", [{"NFFT_HIP_WORK_LIST": "1"}, {"NFFT_HIP_WORK_LIST": "1", "NFFT_HIP_STREAM_MIN": "1"},
                                       {"NFFT_HIP_WORK_LIST": "1", "NFFT_HIP_OWNED": "0"}])
def test_work_list_forced_on_a_uniform_input(env_extra):
    """NFFT_HIP_WORK_LIST=1 runs every wide plan from its work list -- the persistent form of the matrix-core kernels that
    otherwise only unbalanced inputs reach: spreading (scatter and owner-computes), the lock-step and the streamed gather, the
    wave-per-column gather (6 columns), two point sets of different size, on a 128^3 grid against the oracle."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
import torch_nfft_amd as tn
from torch_nfft_amd import ops
from oracle import nfft_ref
rng = np.random.default_rng(41)
n, N, m = 9000, 64, 4
pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
batch = (np.arange(n) >= n // 3).astype(np.int64)
errs = []
for cols in ((), (6,)):
    x = rng.standard_normal((n,) + cols).astype(np.float32)
    xt, pt, bt = torch.from_numpy(x).cuda(), torch.from_numpy(pos).cuda(), torch.from_numpy(batch).cuda()
    y = tn.nfft_adjoint(xt, pt, bt, bandwidth=N, cutoff=m)
    ref = nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)
    errs.append(np.linalg.norm(y.cpu().numpy() - ref) / np.linalg.norm(ref))
    f = tn.nfft_forward(y, pt, bt, cutoff=m)
    reff = nfft_ref.nfft_forward(y.cpu().numpy(), pos, batch, m=m)
    errs.append(np.linalg.norm(f.cpu().numpy() - reff) / np.linalg.norm(reff))
ops.check_status()
print("RESULT", max(errs))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) < T1W


@pytest.mark.parametrize("scale", [1e-25, 1e-8, 1e12, 1e30])
def test_input_scale_invariance(tn, scale):
    """The matrix-core kernels rescale their f16 operands by powers of two (max |x| for spreading, max |G| per plane
    tile for the gather): results must scale exactly with the input over the fp32 range, and a mixed-magnitude input
    must keep its small entries accurate relative to the total."""
    rng = np.random.default_rng(77)
    n, N, m = 6000, 32, 4
    pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    y1 = tn.nfft_adjoint(dev(x), dev(pos), None, bandwidth=N, cutoff=m)
    ys = tn.nfft_adjoint(dev((x * np.float32(scale)).astype(np.float32)), dev(pos), None, bandwidth=N, cutoff=m)
    assert rel_l2(host(ys) / scale, host(y1)) < 2e-6
    f1 = tn.nfft_forward(y1, dev(pos), None, cutoff=m)
    fs = tn.nfft_forward(y1 * scale, dev(pos), None, cutoff=m)
    assert rel_l2(host(fs) / scale, host(f1)) < 2e-6
    # mixed magnitudes: one huge coefficient among ordinary ones
    xm = x.copy()
    xm[0] = 1e6
    ym = tn.nfft_adjoint(dev(xm), dev(pos), None, bandwidth=N, cutoff=m)
    assert rel_l2(host(ym), nfft_ref.nfft_adjoint(xm, pos, None, N=N, m=m)) < T1


def test_clustered_points_many_per_tile(tn):
    """All points inside one grid cell neighbourhood: stresses LDS accumulation order and the chunk sweep."""
    rng = np.random.default_rng(71)
    n = 5000
    pos = (0.013 * rng.standard_normal((n, 3)) + np.array([0.31, -0.07, 0.49])).astype(np.float32)
    x = rng.standard_normal((n,)).astype(np.float32)
    ya = tn.nfft_adjoint(dev(x), dev(pos), None, bandwidth=32, cutoff=4)
    sub = np.stack(np.meshgrid(*[np.arange(-4, 4)] * 3, indexing="ij"), -1).reshape(-1, 3)
    ex = ndft.ndft_adjoint_subset(x, pos, sub)[:, 0]
    got = host(ya)[0][tuple((sub + 16).T)]
    assert rel_l2(got, ex) < T2[4]


def test_columns_of_very_different_magnitude(tn):
    """Columns (and point sets) whose magnitudes differ by many orders: the matrix-core spreading kernel scales its f16
    operands per work item and column, so every column keeps fp32-grade accuracy RELATIVE TO ITSELF (a single global
    scale would leave ~12 bits at 1e-6 and flush 1e-12 to zero; the reference spreads each column independently in
    fp32, csrc/cuda/spatial_window_operations.cu:103-171)."""
    rng = np.random.default_rng(4321)
    n, N, m = 30000, 64, 4
    pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
    batch = (np.arange(n) >= n // 2).astype(np.int64)
    scales = np.array([1.0, 1e-6, 1e-12, 1e9], np.float32)
    x = (rng.standard_normal((n, 4)) * scales[None, :]).astype(np.float32)
    x[n // 2:] *= np.float32(1e-5)  # the second point set is another 1e-5 smaller
    y = host(tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=N, cutoff=m))
    ref = nfft_ref.nfft_adjoint(x.astype(np.float64), pos, batch, N=N, m=m)
    for b in range(2):
        for c in range(4):
            assert rel_l2(y[b, ..., c], ref[b, ..., c]) < 2e-6, (b, c)
    # complex coefficients: real and imaginary planes of different magnitude
    xc = (x[:, 0] + 1j * x[:, 2]).astype(np.complex64)
    yc = host(tn.nfft_adjoint(dev(xc), dev(pos), None, bandwidth=N, cutoff=m))
    ri = nfft_ref.nfft_adjoint(1j * x[:, 2].astype(np.float64), pos, None, N=N, m=m)
    yi = host(tn.nfft_adjoint(dev((1j * x[:, 2]).astype(np.complex64)), dev(pos), None, bandwidth=N, cutoff=m))
    assert rel_l2(yi, ri) < 2e-6
    assert rel_l2(yc, nfft_ref.nfft_adjoint(xc.astype(np.complex128), pos, None, N=N, m=m)) < 2e-6


# ----------------------------------------------------------------------------- owner-computes spreading (sparse inputs)

def _owned_stage_case(m, Cr):
    import ctypes
    from torch_nfft_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(510 + m)
    d, N, B = 3, 64, 3
    M = 2 * N
    cells = np.array([0, 1, 30, 31, 32, 33, 62, 63, 64, 65, 95, 96, 126, 127])
    corner = np.stack(np.meshgrid(cells, cells[::3], cells, indexing="ij"), -1).reshape(-1, 3)
    edge = ((corner + rng.random(corner.shape) * 0.999) / M - 0.5).astype(np.float32)
    rnd = (rng.random((1500, 3)) - 0.5).astype(np.float32)
    pos = np.concatenate([edge, rnd]).astype(np.float32)
    n = pos.shape[0]
    batch = np.sort(rng.integers(0, B, n)).astype(np.int64)
    batch[0], batch[-1] = 0, B - 1
    perm = rng.permutation(n)
    pos = pos[perm]  # (batch stays sorted, the points of a set are shuffled)
    x = rng.standard_normal((n, Cr)).astype(np.float32)
    prob = _lib.Problem(d, n, Cr, B, N, m)
    plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
    post, batcht, xt = dev(pos), dev(batch), dev(x)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(post), p(batcht), p(plan), plan.numel(), s))
    sb = lib.nfft_hip_spread_scratch_bytes(ctypes.byref(prob), Cr)
    assert sb >= 4 * n * Cr * 4  # the owned plan has room for four entries per point
    scratch = torch.empty(sb // 4, device="cuda")
    grid = torch.full((B * Cr,) + (M,) * d, float("nan"), device="cuda")
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(xt), Cr, p(grid), p(scratch), s))
    ref = nfft_ref.spread(x, pos, batch, N, m).real.reshape((B * Cr,) + (M,) * d)
    got = host(grid)
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) < 2e-6
    grid2 = torch.full_like(grid, float("nan"))
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(xt), Cr, p(grid2), p(scratch), s))
    assert torch.equal(grid, grid2)


@pytest.mark.parametrize("m,Cr", [(1, 2), (4, 2), (7, 2), (4, 1), (4, 3), (2, 5), (7, 3)])
def test_owned_spreading_stage_tile_borders(tn, m, Cr):
    """Sparse 3-D problems spread by owner-computes (a plan entry per touched tile, plain stores; one column: 32 x 64
    tiles, one sweep of the points per column; two or more: 32 x 32 tiles, one sweep per PAIR of columns -- an odd column
    count leaves every point set's last column to a sweep of its own): points on tile corners and edges, on the torus
    boundary, in neighbouring point sets -- the spread grid against the oracle's gridding, written completely (the grid is
    pre-filled with NaN) and bitwise reproducible.  (A single column on a 128^3 grid takes the scatter variant by default
    since round 4: that case runs in a process that forces the owned one.)"""
    if Cr >= 2:
        return _owned_stage_case(m, Cr)
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path[:0] = [%r, %r]; import test_gpu_parity as t; t._owned_stage_case(%d, %d); print('CASE OK')" % (
        root, os.path.join(root, "tests"), m, Cr)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NFFT_HIP_OWNED="1"), capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "CASE OK" in out.stdout, out.stderr[-3000:] + out.stdout[-1000:]


@pytest.mark.parametrize("env_extra", [{"NFFT_HIP_OWNED": "0"}, {"NFFT_HIP_OWNED": "0", "NFFT_HIP_STREAM_MIN": "1"}],
                         ids=["scatter", "scatter+groups"])
def test_scatter_spreading_stage_dense_128_cubed(env_extra):
    """The scatter variant of the matrix-core spreading kernel (the C3 kernel) at stage level: nfft_hip_plan_points +
    nfft_hip_spread on a dense 128^3 problem (120 000 points = 0.057 per cell, two real columns, a cluster, points on
    the torus boundary), the spread grid against the oracle's gridding at 2e-6 -- an FFT-side error cannot cancel a
    spreading-side one here.  Second case: the plan ordered by column groups (K-blocks that skip a half tile), as at C3.
    Semantics: csrc/cuda/spatial_window_operations.cu:103-171."""
    import subprocess
    import sys
    code = r'''
import ctypes, numpy as np, torch, sys
sys.path.insert(0, %r)
from torch_nfft_amd import _lib
from oracle import nfft_ref
lib = _lib.load()
rng = np.random.default_rng(77)
d, N, m, n, Cr = 3, 64, 4, 120000, 2
M = 2 * N
pos = (rng.random((n, d)) - 0.5).astype(np.float32)
pos[:20000] = (0.03 * rng.standard_normal((20000, 3)) + 0.1).astype(np.float32)
pos[20000:20600] = (np.abs(0.004 * rng.standard_normal((600, 3))) - 0.5).astype(np.float32)  # at the periodic corner
pos = np.clip(pos, -0.5, np.nextafter(np.float32(0.5), np.float32(0))).astype(np.float32)
x = rng.standard_normal((n, Cr)).astype(np.float32)
prob = _lib.Problem(d, n, Cr, 1, N, m)
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
post, xt = torch.from_numpy(pos).cuda(), torch.from_numpy(x).cuda()
plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
_lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(post), None, p(plan), plan.numel(), s))
scratch = torch.empty(lib.nfft_hip_spread_scratch_bytes(ctypes.byref(prob), Cr) // 4, device="cuda")
grid = torch.full((Cr, M, M, M), float("nan"), device="cuda")
_lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(xt), Cr, p(grid), p(scratch), s))
_lib.check_status()
ref = nfft_ref.spread(x, pos, None, N, m).real.reshape((Cr, M, M, M))
got = grid.cpu().numpy()
print("FINITE", bool(np.isfinite(got).all()))
print("RESULT", np.linalg.norm((got - ref).ravel()) / np.linalg.norm(ref.ravel()))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    assert "FINITE True" in out.stdout, out.stdout
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) < T1W, out.stdout


@pytest.mark.parametrize("owned", ["0", "1"])
def test_owned_and_scatter_spreading_agree_dense_clustered(owned):
    """NFFT_HIP_OWNED forces either spreading variant whatever the density: 600 000 points, half of them in two tight
    clusters (dense slab ranges are cut into pieces: the work list) and one cluster on the periodic corner, two point sets,
    128^3 grid -- adjoint on a frequency subset vs the exact NDFT and vs the float64 algorithm restatement."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
import torch_nfft_amd as tn
from oracle import ndft
rng = np.random.default_rng(8)
n, N, m = 600_000, 64, 4
pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
pos[:200_000] = (0.01 * rng.standard_normal((200_000, 3)) + np.array([0.1, -0.2, 0.3])).astype(np.float32)
pos[200_000:300_000] = (0.02 * rng.standard_normal((100_000, 3)) + 0.4999).astype(np.float32)
pos = (pos - np.floor(pos + 0.5)).astype(np.float32)
batch = (np.arange(n) >= 350_000).astype(np.int64)
x = rng.standard_normal((n, 2)).astype(np.float32)
xt, pt, bt = (torch.from_numpy(a).cuda() for a in (x, pos, batch))
y = tn.nfft_adjoint(xt, pt, bt, bandwidth=N, cutoff=m).cpu().numpy()
y2 = tn.nfft_adjoint(xt, pt, bt, bandwidth=N, cutoff=m).cpu().numpy()
freqs = rng.integers(-N // 2, N // 2, size=(40, 3))
err = 0.0
for b in range(2):
    sel = batch == b
    ex = ndft.ndft_adjoint_subset(x[sel], pos[sel], freqs)
    got = y[b][tuple((freqs + N // 2).T)]
    err = max(err, np.linalg.norm(got - ex) / np.linalg.norm(ex))
print("RESULT", err, bool((y == y2).all()))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NFFT_HIP_OWNED=owned)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) < T2[4]
    if owned == "1":
        assert line[2] == "True"  # no atomics: bitwise reproducible


# ----------------------------------------------------------------------------- wave-per-column interpolation

@pytest.mark.parametrize("C,complex_out,chunk_planes,m", [(5, True, None, 4), (3, False, None, 4), (9, False, 7, 4),
                                                          (4, True, 5, 4), (6, False, None, 2), (4, False, None, 6),
                                                          (2, True, None, 7)])
def test_forward_many_columns_wave_per_column(tn, monkeypatch, C, complex_out, chunk_planes, m):
    """From 4 real planes per point set the forward gather runs one wave per column (interp_cols.hip): column counts
    that do not fill the last group of 8, chunks of planes that start in the middle of a group and of a point set,
    a pencil holding far more than one group of 384 points (several plane sweeps), empty point sets, points on the
    torus boundary -- forward transform against the float64 algorithm restatement, per column."""
    rng = np.random.default_rng(600 + C + m)
    N, B = 32, 4
    n_dense, n_rest = 6000, 3000
    dense = (0.02 * rng.standard_normal((n_dense, 3)) + np.array([0.11, -0.23, 0.37])).astype(np.float32)
    edge = np.array([[-0.5, -0.5, -0.5], [0.49999997, 0.49999997, 0.49999997], [0.0, 0.0, 0.0]], np.float32)
    pos = np.concatenate([dense, edge, (rng.random((n_rest, 3)) - 0.5).astype(np.float32)]).astype(np.float32)
    n = pos.shape[0]
    batch = np.sort(rng.choice([0, 1, 3], size=n)).astype(np.int64)  # point set 2 is empty
    batch[0], batch[-1] = 0, 3
    pos = pos[rng.permutation(n)]
    shape = (B, N, N, N, C)
    xh = rng.standard_normal(shape).astype(np.float32)
    if complex_out:
        xh = (xh + 1j * rng.standard_normal(shape)).astype(np.complex64)
    if chunk_planes is not None:
        monkeypatch.setenv("NFFT_HIP_CHUNK_BYTES", str(chunk_planes * (64 ** 3 * 4 + 64 * 64 * 33 * 8) + 8))
    y = host(tn.nfft_forward(dev(xh), dev(pos), dev(batch), cutoff=m, real_output=not complex_out))
    ref = nfft_ref.nfft_forward(xh, pos, batch, m=m, real_output=not complex_out)
    assert y.shape == (n, C)
    for c in range(C):
        assert rel_l2(y[:, c], ref[:, c]) < T1W, c
    assert rel_l2(y, ref) < 2e-6


@pytest.mark.parametrize("env_extra", [{"NFFT_HIP_STREAM_MIN": "1"}, {"NFFT_HIP_STREAM_MIN": "1", "NFFT_HIP_COLGROUPS": "0"},
                                       {"NFFT_HIP_XGATHER": "1"}],
                         ids=["stream+groups", "stream-nogroups", "separate-permutation"])
def test_streamed_gather_and_column_groups_on_small_problems(env_extra):
    """The streamed gather and the column-group order of the plan are chosen for big work items only; with
    NFFT_HIP_STREAM_MIN=1 they run on problems the oracle can check.  Every cutoff of the wide tiling (chunks of 13 ... 1
    slabs, windows of 4 ... 16 taps), grids whose last pencil is partial, points on pencil and group boundaries, a dense
    cluster (cut pieces), empty regions, two point sets: adjoint and forward vs the float64 oracle.  The third case
    runs the coefficient permutation as a pass of its own (the default does it inside the spreading kernel)."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
import torch_nfft_amd as tn
from oracle import nfft_ref
worst = 0.0
for case, (N, m, n, nsets) in enumerate([(32, 1, 3000, 1), (32, 2, 3000, 1), (64, 3, 5000, 1), (64, 4, 6000, 2), (32, 5, 3000, 1),
                                         (64, 6, 4000, 1), (32, 7, 2500, 1), (80, 4, 5000, 1)]):
    rng = np.random.default_rng(100 + case)
    M = 2 * N
    pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
    # a dense cluster, and points exactly on cell / pencil / column-group boundaries of the first pencils
    pos[:n // 4] = (0.01 * rng.standard_normal((n // 4, 3)) + 0.2).astype(np.float32)
    W = 2 * m + 2
    T2 = 65 - W
    edges = np.array([0, 32 - W, 33 - W, 31, 32, T2 - 1, T2, T2 + 32 - W], dtype=np.float64)
    k = min(len(edges) * 8, n // 8)
    pos[n // 4:n // 4 + k, 2] = ((np.resize(edges, k) + rng.integers(0, 2, k) * 0.999) / M - 0.5).astype(np.float32)
    pos[n // 4:n // 4 + k, 1] = ((rng.integers(0, M, k)) / M - 0.5).astype(np.float32)
    pos = np.clip(pos, -0.5, np.nextafter(np.float32(0.5), np.float32(0))).astype(np.float32)
    batch = None
    bt = None
    if nsets == 2:
        batch = (np.arange(n) >= n // 3).astype(np.int64)
        bt = torch.from_numpy(batch).cuda()
    x = rng.standard_normal((n, 2)).astype(np.float32)
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(pos).cuda()
    y = tn.nfft_adjoint(xt, pt, bt, bandwidth=N, cutoff=m)
    ref = nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)
    e1 = np.linalg.norm(y.cpu().numpy() - ref) / np.linalg.norm(ref)
    f = tn.nfft_forward(y, pt, bt, cutoff=m)
    reff = nfft_ref.nfft_forward(y.cpu().numpy(), pos, batch, m=m)
    e2 = np.linalg.norm(f.cpu().numpy() - reff) / np.linalg.norm(reff)
    print("CASE", N, m, n, nsets, e1, e2)
    worst = max(worst, e1, e2)
print("RESULT", worst)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) < T1W, out.stdout


@pytest.mark.parametrize("log_n", [17, 18, 20])
def test_large_1d_grid(tn, log_n):
    """1-D bandwidths of 2^17 ... 2^20 (the largest the library accepts): cell indices beyond the range in which the
    point sort derives tile indices from a float reciprocal (common.h div_small), more first-level bins than the
    two-level sort takes (fallback plan), and frequencies whose square exceeds 2^31 (the roll-off factor once formed
    k * k in 32-bit integers: every coefficient with |k| >= 46341 was wrong) -- points near both ends of the grid and a
    cluster, adjoint and forward against the oracle."""
    rng = np.random.default_rng(2024)
    n, N, m = 6000, 1 << log_n, 4
    pos = (rng.random((n, 1)) - 0.5).astype(np.float32)
    pos[:200, 0] = np.float32(0.5) - rng.random(200).astype(np.float32) * np.float32(1e-4)
    pos[200:400, 0] = np.float32(-0.5) + rng.random(200).astype(np.float32) * np.float32(1e-4)
    pos[400:1000, 0] = (0.3 + 1e-3 * rng.standard_normal(600)).astype(np.float32)
    pos = np.clip(pos, -0.5, np.nextafter(np.float32(0.5), np.float32(0))).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    y = tn.nfft_adjoint(dev(x), dev(pos), None, bandwidth=N, cutoff=m)
    ref = nfft_ref.nfft_adjoint(x, pos, None, N=N, m=m)
    assert rel_l2(host(y), ref) < T1
    f = tn.nfft_forward(y, dev(pos), None, cutoff=m)
    assert rel_l2(host(f), nfft_ref.nfft_forward(host(y), pos, None, m=m)) < T1


# ----------------------------------------------------------------------------- several columns: column-innermost passes

@pytest.mark.parametrize("N,B,C,complex_x,real_output", [(64, 1, 40, False, False), (64, 2, 24, False, True),
                                                          (64, 1, 17, True, False), (64, 3, 11, True, True),
                                                          (128, 1, 33, False, True)])
def test_many_columns_column_innermost_passes(tn, N, B, C, complex_x, real_output):
    """Chunks of >= 32 planes with several coefficient columns take the column-innermost FFT passes (groups of 16 planes,
    plane index innermost; the last adjoint pass writes / the first forward pass reads the reference's [B, N^3, C] layout
    in place: docs/source/theory/dataformat.rst:43-63, spectral_window_operations.cu:51-111).  Partial last groups, groups
    that straddle point sets, (re, im) plane pairs, real output: every column against the single-column transform of the
    same data (the plane-by-plane pipeline), two columns against the oracle."""
    rng = np.random.default_rng(7 * N + C)
    m, n = 3, 6000
    pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
    batch = None
    if B > 1:
        batch = np.sort(rng.integers(0, B, n)).astype(np.int64)
        batch[0], batch[-1] = 0, B - 1
    x = rng.standard_normal((n, C)).astype(np.float32)
    if complex_x:
        x = (x + 1j * rng.standard_normal((n, C))).astype(np.complex64)
    pt, bt, xt = dev(pos), dev(batch), dev(x)
    y = tn.nfft_adjoint(xt, pt, bt, bandwidth=N, cutoff=m, real_output=real_output)
    assert y.shape == (B, N, N, N, C)
    for c in rng.choice(C, size=5, replace=False):
        yc = tn.nfft_adjoint(xt[:, c].contiguous(), pt, bt, bandwidth=N, cutoff=m, real_output=real_output)
        assert rel_l2(host(y[..., c]), host(yc)) < 1e-6, c
    for c in (0, C - 1):
        ref = nfft_ref.nfft_adjoint(x[:, c], pos, batch, N=N, m=m, real_output=real_output)
        assert rel_l2(host(y[..., c]), ref) < T1W
    # forward of a random spectrum
    xh = torch.randn((B, N, N, N, C), device="cuda")
    if complex_x:
        xh = torch.complex(xh, torch.randn((B, N, N, N, C), device="cuda"))
    f = tn.nfft_forward(xh, pt, bt, cutoff=m, real_output=real_output)
    assert f.shape == (n, C)
    for c in rng.choice(C, size=5, replace=False):
        fc = tn.nfft_forward(xh[..., c].contiguous(), pt, bt, cutoff=m, real_output=real_output)
        assert rel_l2(host(f[:, c]), host(fc)) < 1e-6, c
    c = C // 2
    reff = nfft_ref.nfft_forward(host(xh[..., c]), pos, batch, m=m, real_output=real_output)
    assert rel_l2(host(f[:, c]), reff) < T1W
