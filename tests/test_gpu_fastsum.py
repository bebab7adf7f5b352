"""GPU tests of the widened surface: nfft_fastsum, the coefficient operators and the
matrix-free convenience layer, against the oracle and the golden vectors frozen from the reference's ndft.py.
Scenarios follow the reference's test/test_fastsum.py, test/test_kernel.py and test/test_grad.py."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import coeffs_ref, ndft, nfft_ref

pytestmark = pytest.mark.gpu
T1 = 2e-5


@pytest.fixture(scope="module")
def tn():
    import torch_nfft_amd
    return torch_nfft_amd


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("dim,N", [(1, 32), (2, 16), (3, 8), (2, 12)])
def test_coefficient_operators(tn, dim, N):
    sigma = 0.17
    a = tn.gaussian_analytic_coeffs(sigma, dim=dim, N=N)
    assert a.shape == (N,) * dim and a.dtype == torch.float32
    assert rel_l2(host(a), coeffs_ref.gaussian_analytic_coeffs(sigma, dim, N)) < 1e-6
    for p in (-1, 0):
        c = tn.gaussian_interpolated_coeffs(sigma, dim=dim, N=N, p=p)
        assert c.shape == (N,) * dim and c.dtype == torch.complex64
        assert rel_l2(host(c), coeffs_ref.gaussian_interpolated_coeffs(sigma, dim, N, p)) < 2e-6
    g = tn.interpolation_grid(dim=dim, N=N)
    assert g.shape == (N,) * dim + (dim,)
    assert np.abs(host(g) - coeffs_ref.interpolation_grid(dim, N)).max() < 1e-7
    r = tn.radial_interpolation_grid(dim=dim, N=N)
    assert np.abs(host(r) - coeffs_ref.radial_interpolation_grid(dim, N)).max() < 1e-6
    # user-sampled kernel: real and complex samples
    rng = np.random.default_rng(3)
    vals = rng.standard_normal((N,) * dim).astype(np.float32)
    assert rel_l2(host(tn.interpolated_kernel_coeffs(dev(vals))), coeffs_ref.interpolated_kernel_coeffs(vals)) < 2e-6
    valc = (vals + 1j * rng.standard_normal((N,) * dim)).astype(np.complex64)
    assert rel_l2(host(tn.interpolated_kernel_coeffs(dev(valc))), coeffs_ref.interpolated_kernel_coeffs(valc)) < 2e-6
    # sampling the Gaussian on the grid and interpolating reproduces gaussian_interpolated_coeffs(p=-1)
    samp = torch.exp(-(r ** 2) / sigma ** 2)
    assert rel_l2(host(tn.interpolated_kernel_coeffs(samp)),
                  host(tn.gaussian_interpolated_coeffs(sigma, dim=dim, N=N, p=-1))) < 2e-6
    with pytest.raises(RuntimeError, match="only implemented for p<=0"):
        tn.gaussian_interpolated_coeffs(sigma, dim=dim, N=N, p=2)
    with pytest.raises(RuntimeError, match="only implemented for eps=0"):
        tn.gaussian_interpolated_coeffs(sigma, dim=dim, N=N, p=0, eps=0.1)


def test_fastsum_golden_g6(tn):
    """ndft_fastsum of the reference (golden) vs nfft_fastsum, d=2, N=8, test_fastsum.py shapes."""
    g = load_golden("g6_fastsum_2d")
    y = tn.nfft_fastsum(dev(g["x"]), dev(g["coeffs"]), dev(g["pos"]), cutoff=4)
    assert y.shape == (200, 2) and y.dtype == torch.float32
    assert rel_l2(host(y), g["y_fastsum"]) < 1e-3            # NFFT approximation at m=4, N=8
    assert rel_l2(host(y), nfft_ref.nfft_fastsum(g["x"], g["coeffs"], g["pos"], m=4)) < T1
    # fastsum with analytic coefficients approximates the Gaussian kernel matrix (test_fastsum.py:20-34)
    A = tn.nfft_fastsum(torch.eye(200, device="cuda"), dev(g["coeffs"]), dev(g["pos"]), cutoff=4)
    assert np.abs(host(A) - g["exact_trig"].real).max() < 2e-3
    assert np.abs(host(A) - g["exact_gauss"]).max() < 0.1     # N=8 truncation of the kernel's Fourier series


@pytest.mark.parametrize("d,N,m", [(1, 32, 3), (2, 16, 3), (3, 16, 4), (3, 12, 2),
                                   (2, 64, 4), (2, 128, 3)])  # (2-D grids of 128^2 up: own row + column passes, the kernel
                                                             # coefficients ride on the column pass of the adjoint)
@pytest.mark.parametrize("complex_x", [False, True])
def test_fastsum_vs_oracle(tn, d, N, m, complex_x):
    rng = np.random.default_rng(10 * d + N)
    ns, nt, B = 300, 200, 2
    src = (0.5 * (rng.random((ns, d)) - 0.5)).astype(np.float32)
    tgt = (0.5 * (rng.random((nt, d)) - 0.5)).astype(np.float32)
    sb = np.sort(rng.integers(0, B, ns)).astype(np.int64); sb[0], sb[-1] = 0, B - 1
    tb = np.sort(rng.integers(0, B, nt)).astype(np.int64); tb[0], tb[-1] = 0, B - 1
    x = rng.standard_normal((ns, 3)).astype(np.float32)
    if complex_x:
        x = (x + 1j * rng.standard_normal((ns, 3))).astype(np.complex64)
    coeffs = coeffs_ref.gaussian_interpolated_coeffs(0.2, d, N, p=0).astype(np.complex64)
    y = tn.nfft_fastsum(dev(x), dev(coeffs), dev(src), dev(tgt), dev(sb), dev(tb), cutoff=m)
    assert y.shape == (nt, 3) and y.dtype == (torch.complex64 if complex_x else torch.float32)
    assert rel_l2(host(y), nfft_ref.nfft_fastsum(x, coeffs, src, tgt, sb, tb, m=m)) < T1
    # exact trigonometric sum (ndft_fastsum) within the window's approximation error
    tol = {2: 3e-2, 3: 5e-3, 4: 1e-3}[m]
    assert rel_l2(host(y), ndft.ndft_fastsum(x, coeffs, src, tgt, sb, tb)) < tol
    # symmetric variant with a shared batch vector and real coefficients
    ca = coeffs_ref.gaussian_analytic_coeffs(0.2, d, N).astype(np.float32)
    ys = tn.nfft_fastsum(dev(x), dev(ca), dev(src), batch=dev(sb), cutoff=m)
    assert rel_l2(host(ys), nfft_ref.nfft_fastsum(x, ca, src, batch=sb, m=m)) < T1


def test_fastsum_autograd_is_transpose(tn):
    """backward of fastsum = fastsum with sources and targets swapped (nfft.py:82-88; test_grad.py:79-102)."""
    rng = np.random.default_rng(5)
    src = dev((0.5 * (rng.random((40, 2)) - 0.5)).astype(np.float32))
    tgt = dev((0.5 * (rng.random((30, 2)) - 0.5)).astype(np.float32))
    coeffs = tn.gaussian_interpolated_coeffs(0.2, dim=2, N=16, p=0)
    x = torch.randn((40, 3), device="cuda", requires_grad=True)
    y = tn.nfft_fastsum(x, coeffs, src, tgt, cutoff=3)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    expect = tn.nfft_fastsum(w, coeffs, tgt, src, cutoff=3)
    assert rel_l2(host(x.grad), host(expect)) < 1e-5
    # <K x, w> = <x, K^T w>
    assert abs(float((y * w).sum()) - float((x.detach() * expect).sum())) < 1e-3 * float(y.abs().sum())


def test_gaussian_kernel_matrices(tn):
    """test_kernel.py: dense Gram matrix from GaussianKernel vs the exact Gaussian matrix, absolute and relative
    sigma, batched (the reference needs torch_scatter for the batched radius; this build does not)."""
    rng = np.random.default_rng(9)
    n, b, dim, diameter, N, m = 6, 2, 2, 10.0, 32, 4
    pos = (diameter * (rng.random((n * b, dim)) - 0.5)).astype(np.float32)
    batch = np.repeat(np.arange(b), n).astype(np.int64)
    post, batcht = dev(pos), dev(batch)

    def exact(sig, p):
        out = np.zeros((n * b, n * b))
        for k in range(b):
            q = p[batch == k]
            d2 = ((q[:, None, :] - q[None, :, :]) ** 2).sum(-1)
            out[k * n:(k + 1) * n, k * n:(k + 1) * n] = np.exp(-d2 / sig ** 2)
        return out

    kern = tn.GaussianKernel(1.0 * diameter, dim, N, m, shift_by_center=True, max_infinity_norm=diameter / 2,
                             reg_degree=0)
    dense = host(kern(post, batch=batcht).to_dense())
    assert np.abs(dense - exact(diameter, pos)).max() < 5e-3
    kern = tn.GaussianKernel(1.0, dim, N, m, shift_by_center=True, reg_degree=0)
    dense = host(kern(post, batch=batcht).to_dense())
    shifted = tn.utils.shift_points_by_center(post, batch=batcht)[0]
    scaled = host(tn.utils.scale_points_by_norm(shifted, batch=batcht, norm="euclidean")[0])
    assert np.abs(dense - exact(1.0, scaled)).max() < 5e-3
    # analytic coefficients, un-batched, L-infinity scaling
    kern = tn.GaussianKernel(0.8, dim, N, m, analytic=True)
    K = kern(post[:n])
    dense = host(K.to_dense())
    p1 = host(tn.utils.scale_points_by_norm(tn.utils.shift_points_by_center(post[:n])[0], norm="infinity")[0])
    d2 = ((p1[:, None, :] - p1[None, :, :]) ** 2).sum(-1)
    assert np.abs(dense - np.exp(-d2 / 0.8 ** 2)).max() < 5e-3
    assert K.is_symmetric() and K.T is K
    # adjacency matrix with symmetric normalisation and Laplacian shift against the dense formulas
    W = dense - np.eye(n)                       # loop_weight = 0 removes the unit diagonal
    deg = W.sum(1)
    A = kern.adjacency_matrix(post[:n], loop_weight=0, normalization="sym", shift="laplacian")
    x = rng.standard_normal((n, 2)).astype(np.float32)
    ref = x - (W / np.sqrt(deg)[:, None] / np.sqrt(deg)[None, :]) @ x
    assert np.abs(host(A @ dev(x)) - ref).max() < 2e-2
    A2 = kern.adjacency_matrix(post[:n], loop_weight=0, normalization="rw")
    assert np.abs(host(A2 @ dev(x)) - (W / deg[:, None]) @ x).max() < 2e-2
    assert np.abs(host(A2.T @ dev(x)) - (W / deg[None, :]) @ x).max() < 2e-2
    A3 = kern.adjacency_matrix(post[:n], loop_weight=0, shift="signless")
    assert np.abs(host(A3 @ dev(x)) - (deg[:, None] * x + W @ x)).max() < 2e-2
    # the attribute / method names code written against the reference reads (matrices.py:120-151)
    assert np.abs(host(A3.degrees) - deg).max() < 2e-2 and A3.d_inv is None and A3.d_inv_sqrt is None
    assert np.abs(host(A.d_inv_sqrt) - 1 / np.sqrt(deg)).max() < 2e-2 and A.degrees is None
    assert np.abs(host(A2.d_inv) - 1 / deg).max() < 2e-2 and np.abs(host(A2.T.d_inv) - 1 / deg).max() < 2e-2
    xt = dev(x)
    assert torch.equal(A3.apply_shift(xt, xt), A3.degrees[:, None] * xt + xt) and A2.apply_shift(xt, xt) is xt


def test_fastsum_operator_and_checks(tn):
    rng = np.random.default_rng(13)
    src = dev((0.4 * (rng.random((50, 2)) - 0.5)).astype(np.float32))
    x = torch.randn(50, device="cuda")
    coeffs = tn.gaussian_analytic_coeffs(0.3, dim=2, N=16)
    y1 = torch.ops.torch_nfft.nfft_fastsum(src, src, x, coeffs, None, None, 3)
    y2 = tn.nfft_fastsum(x, coeffs, src, cutoff=3)
    assert rel_l2(host(y1), host(y2)) < 1e-6
    with pytest.raises(RuntimeError, match="Input mismatch"):
        tn.nfft_fastsum(x, tn.gaussian_analytic_coeffs(0.3, dim=3, N=16), src)
    with pytest.raises(RuntimeError, match="only implemented for GPU tensors"):
        tn.nfft_fastsum(x.cpu(), coeffs, src)
    s = str(torch.ops.torch_nfft.gaussian_interpolated_coeffs.default._schema)
    assert s.startswith("torch_nfft::gaussian_interpolated_coeffs(float sigma, int N, int dim, int p, float eps)")
