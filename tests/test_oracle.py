"""CPU tests: the oracle (oracle/) against the golden vectors frozen from the
reference's own exact transform (torch_nfft/ndft.py, see oracle/make_golden.py)."""
import numpy as np
import pytest

from conftest import load_golden, rel_l2
from oracle import ndft, ndft_cpu, nfft_ref

# fp32 tolerance of the reference's own ground truth (complex64 tensordot): ~1e-6 relative
TOL_EXACT = 5e-6
# NFFT approximation error vs exact NDFT, by cutoff m
TOL_NFFT = {2: 2e-2, 3: 3e-3, 4: 5e-4, 8: 1e-5}

IMPLS = [pytest.param(ndft, id="numpy"), pytest.param(ndft_cpu, id="c-openmp")]


@pytest.mark.parametrize("impl", IMPLS)
def test_g1_adjoint_2d_batched(impl):
    g = load_golden("g1_adjoint_2d_batched")
    y = impl.ndft_adjoint(g["x"], g["pos"], g["batch"], N=int(g["N"]))
    assert y.shape == g["y_adjoint"].shape == (3, 16, 16, 10)
    assert rel_l2(y, g["y_adjoint"]) < TOL_EXACT


@pytest.mark.parametrize("impl", IMPLS)
def test_g2_forward_2d(impl):
    g = load_golden("g2_forward_2d")
    y = impl.ndft_forward(g["x"], g["pos"], None)
    assert y.shape == g["y_forward"].shape == (10, 1)
    assert rel_l2(y, g["y_forward"]) < TOL_EXACT


@pytest.mark.parametrize("impl", IMPLS)
def test_g3_1d(impl):
    g = load_golden("g3_1d_n64")
    ya = impl.ndft_adjoint(g["x"], g["pos"], None, N=64)
    assert ya.shape == (1, 64)
    assert rel_l2(ya, g["y_adjoint"]) < TOL_EXACT
    yf = impl.ndft_forward(g["xhat"], g["pos"], None)
    assert yf.shape == (1000,)
    assert rel_l2(yf, g["y_forward"]) < TOL_EXACT


@pytest.mark.parametrize("impl", IMPLS)
def test_g4_3d_ragged(impl):
    g = load_golden("g4_3d_ragged")
    for key in ("real", "complex"):
        y = impl.ndft_adjoint(g["x_" + key], g["pos"], g["batch"], N=16)
        assert y.shape == (3, 16, 16, 16, 2)
        assert rel_l2(y, g["y_adjoint_" + key]) < TOL_EXACT
    yf = impl.ndft_forward(g["xhat"], g["pos"], g["batch"])
    assert yf.shape == (200, 2)
    assert rel_l2(yf, g["y_forward"]) < TOL_EXACT


@pytest.mark.parametrize("impl", IMPLS)
def test_g5_grad_shapes(impl):
    g = load_golden("g5_grad_shapes")
    assert rel_l2(impl.ndft_adjoint(g["x"], g["pos"], g["batch"], N=16), g["y_adjoint"]) < TOL_EXACT
    assert rel_l2(impl.ndft_forward(g["xhat"], g["pos"], g["batch"]), g["y_forward"]) < TOL_EXACT


def test_g6_fastsum():
    g = load_golden("g6_fastsum_2d")
    y = ndft.ndft_fastsum(g["x"], g["coeffs"], g["pos"])
    assert rel_l2(y, g["y_fastsum"]) < TOL_EXACT
    # fastsum == multiplication with the exact trigonometric matrix (ndft.py:66-95)
    assert rel_l2(g["exact_trig"].real @ g["x"], g["y_fastsum"]) < 1e-5


@pytest.mark.parametrize("d,N,m", [(1, 64, 2), (1, 64, 4), (2, 16, 3), (2, 16, 4), (3, 16, 4), (1, 64, 8)])
def test_algorithm_restatement_vs_exact(d, N, m):
    """The float64 restatement of the reference's NFFT algorithm reproduces the exact
    NDFT up to the window's approximation error (tolerance ladder of SURVEY.md 8c)."""
    rng = np.random.default_rng(1234)
    n = 300
    pos = (rng.random((n, d)) - 0.5).astype(np.float32)
    x = rng.standard_normal((n, 2))
    batch = np.sort(rng.integers(0, 2, n))
    batch[0], batch[-1] = 0, 1
    ya = nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)
    ye = ndft.ndft_adjoint(x, pos, batch, N=N)
    assert rel_l2(ya, ye) < TOL_NFFT[m]
    yf = nfft_ref.nfft_forward(ye, pos, batch, m=m)
    assert rel_l2(yf, ndft.ndft_forward(ye, pos, batch)) < TOL_NFFT[m]


def test_algorithm_real_output_and_edges():
    rng = np.random.default_rng(7)
    # points on the torus boundary and at exact grid nodes exercise the periodic wrap
    pos = np.array([[-0.5, -0.5], [0.49999997, 0.49999997], [0.0, 0.0], [-0.5, 0.25], [0.25, -0.5]], np.float32)
    x = rng.standard_normal((5,))
    ya = nfft_ref.nfft_adjoint(x, pos, None, N=16, m=4)
    assert rel_l2(ya, ndft.ndft_adjoint(x, pos, None, N=16)) < TOL_NFFT[4]
    yr = nfft_ref.nfft_adjoint(x, pos, None, N=16, m=4, real_output=True)
    assert np.array_equal(yr, ya.real)
    yf = nfft_ref.nfft_forward(ya, pos, None, m=4, real_output=True)
    assert yf.dtype == np.float64 and yf.shape == (5,)


def test_adjoint_subset_matches_full():
    rng = np.random.default_rng(3)
    pos = (rng.random((500, 3)) - 0.5).astype(np.float32)
    x = rng.standard_normal((500, 1))
    full = ndft.ndft_adjoint(x, pos, None, N=8)[0]
    freqs = rng.integers(-4, 4, size=(20, 3))
    sub = ndft.ndft_adjoint_subset(x, pos, freqs)
    ref = np.array([full[tuple(f + 4)] for f in freqs])
    assert rel_l2(sub, ref) < 1e-12
