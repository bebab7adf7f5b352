"""GPU parity at BASELINE.json's full sizes, through size-independent checks (a dense oracle is impossible:
the exact NDFT of config C3 is 1.7e14 point-frequency pairs):

  * adjoint: a random subset of frequencies against the chunked float64 NDFT (oracle.ndft.ndft_adjoint_subset);
  * forward: a spectrum with a handful of non-zero frequencies, for which the exact result is a short sum;
  * adjointness <A x, y> = <x, A^H y> between the two transforms at full size;
  * linearity in x;
  * fastsum (config C5): a subset of targets against the exact Gaussian kernel sums.
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import ndft

pytestmark = pytest.mark.gpu
T2_M4 = 5e-4


@pytest.fixture(scope="module")
def tn():
    import torch_nfft_amd
    return torch_nfft_amd


def _subset_check_adjoint(tn, d, N, m, n, nfreq, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    pos = torch.rand((n, d), generator=gen, device="cuda") - 0.5
    x = torch.rand((n,), generator=gen, device="cuda")
    y = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
    rng = np.random.default_rng(seed)
    freqs = rng.integers(-N // 2, N // 2, size=(nfreq, d))
    exact = ndft.ndft_adjoint_subset(x.cpu().numpy()[:, None], pos.cpu().numpy(), freqs)[:, 0]
    got = y[0].cpu().numpy()[tuple((freqs + N // 2).T)]
    return pos, x, y, rel_l2(got, exact)


def _sparse_forward_check(tn, pos, N, m, nnz, seed):
    d = pos.shape[1]
    rng = np.random.default_rng(seed)
    freqs = rng.integers(-N // 2, N // 2, size=(nnz, d))
    vals = (rng.standard_normal(nnz) + 1j * rng.standard_normal(nnz)).astype(np.complex64)
    xh = torch.zeros((1,) + (N,) * d, dtype=torch.complex64, device="cuda")
    for f, v in zip(freqs, vals):
        xh[(0,) + tuple(f + N // 2)] += complex(v)
    y = tn.nfft_forward(xh, pos, None, cutoff=m)
    sel = rng.integers(0, pos.shape[0], size=4096)
    p = pos[sel].cpu().numpy().astype(np.float64)
    exact = (np.exp(-2j * np.pi * (p @ freqs.T.astype(np.float64))) * vals[None, :].astype(np.complex128)).sum(1)
    return xh, y, rel_l2(y[sel].cpu().numpy(), exact)


def test_config_c2_2d_n128_m4_100k(tn):
    pos, x, y, err = _subset_check_adjoint(tn, 2, 128, 4, 100_000, 256, 2)
    assert y.shape == (1, 128, 128) and err < T2_M4
    _, yf, errf = _sparse_forward_check(tn, pos, 128, 4, 12, 3)
    assert errf < T2_M4


def test_grid_1024_cubed(tn):
    """N = 512 (oversampled grid 1024^3, the largest size of the fused column passes: radix sequence 8, 8, 4, 4)."""
    pos, x, y, err = _subset_check_adjoint(tn, 3, 512, 4, 200_000, 48, 9)
    assert y.shape == (1, 512, 512, 512) and err < T2_M4
    del y
    _, yf, errf = _sparse_forward_check(tn, pos, 512, 4, 8, 10)
    assert errf < T2_M4


def test_config_c3_3d_n256_m4_10m(tn):
    n, N, m = 10_000_000, 256, 4
    pos, x, y, err = _subset_check_adjoint(tn, 3, N, m, n, 48, 4)
    assert y.shape == (1, N, N, N) and err < T2_M4
    xh, yf, errf = _sparse_forward_check(tn, pos, N, m, 8, 5)
    assert yf.shape == (n,) and errf < T2_M4
    # adjointness at full size: <A x, xh> = <x, A^H xh>  (A = adjoint transform, A^H = forward transform)
    lhs = torch.sum(y * xh.conj())
    rhs = torch.sum(x.to(torch.complex64) * yf.conj())
    assert abs(complex(lhs) - complex(rhs)) < 1e-4 * abs(complex(lhs)) + 1e-2
    # linearity
    x2 = torch.rand_like(x)
    y2 = tn.nfft_adjoint(x2, pos, None, bandwidth=N, cutoff=m)
    y12 = tn.nfft_adjoint(x + 2 * x2, pos, None, bandwidth=N, cutoff=m)
    assert rel_l2((y + 2 * y2).cpu().numpy(), y12.cpu().numpy()) < 2e-6


def test_clustered_points_work_list(tn):
    """2e6 points in 8 Gaussian clusters (sigma 0.05; SURVEY 8(d)'s robustness distribution) on a 256^3 grid, two point
    sets: dense slab ranges are cut at plan time and all pieces run from the plan's work list in the persistent launch of the
    matrix-core kernels.  Adjoint on a frequency subset vs the exact NDFT, forward of a sparse spectrum vs the exact
    sum, adjointness between the two."""
    N, m, n = 128, 4, 2_000_000
    gen = torch.Generator(device="cuda").manual_seed(31)
    centres = torch.rand((8, 3), generator=gen, device="cuda") - 0.5
    which = torch.randint(0, 8, (n,), generator=gen, device="cuda")
    pos = centres[which] + 0.05 * torch.randn((n, 3), generator=gen, device="cuda")
    pos = pos - torch.floor(pos + 0.5)
    batch = (torch.arange(n, device="cuda") >= n // 3).to(torch.int64)  # two point sets of different size
    x = torch.randn((n,), generator=gen, device="cuda")
    y = tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
    assert y.shape == (2, N, N, N)
    rng = np.random.default_rng(32)
    freqs = rng.integers(-N // 2, N // 2, size=(32, 3))
    for b in range(2):
        sel = (batch == b).cpu().numpy()
        exact = ndft.ndft_adjoint_subset(x.cpu().numpy()[sel][:, None], pos.cpu().numpy()[sel], freqs)[:, 0]
        got = y[b].cpu().numpy()[tuple((freqs + N // 2).T)]
        assert rel_l2(got, exact) < T2_M4
    xh = torch.zeros((2, N, N, N), dtype=torch.complex64, device="cuda")
    f = rng.integers(-N // 2, N // 2, size=(6, 3))
    vals = (rng.standard_normal((2, 6)) + 1j * rng.standard_normal((2, 6))).astype(np.complex64)
    for b in range(2):
        for fr, v in zip(f, vals[b]):
            xh[(b,) + tuple(fr + N // 2)] += complex(v)
    yf = tn.nfft_forward(xh, pos, batch, cutoff=m)
    idx = rng.integers(0, n, size=4096)
    p = pos[idx].cpu().numpy().astype(np.float64)
    bsel = batch[idx].cpu().numpy()
    exact = (np.exp(-2j * np.pi * (p @ f.T.astype(np.float64))) * vals[bsel].astype(np.complex128)).sum(1)
    assert rel_l2(yf[idx].cpu().numpy(), exact) < T2_M4
    lhs = torch.sum(y * xh.conj())
    rhs = torch.sum(x.to(torch.complex64) * yf.conj())
    assert abs(complex(lhs) - complex(rhs)) < 1e-4 * abs(complex(lhs)) + 1e-2


def test_config_c3_clustered_10m(tn):
    """bench.py's `C3-clustered` leg at its full size: 3-D N=256, m=4, 10^7 points in 8 Gaussian clusters (sigma 0.05,
    SURVEY.md 8(d)'s second distribution) -- dense slab ranges cut into pieces (the plan's work list), over-full slabs, the streamed
    gather on ragged work items.  Adjoint on a frequency subset vs the exact NDFT, forward of a sparse spectrum vs the
    exact sum, adjointness between the two, and no device fault left behind."""
    from torch_nfft_amd import ops
    N, m, n = 256, 4, 10_000_000
    gen = torch.Generator(device="cuda").manual_seed(777)
    centres = torch.rand((8, 3), generator=gen, device="cuda") - 0.5
    which = torch.randint(0, 8, (n,), generator=gen, device="cuda")
    pos = centres[which] + 0.05 * torch.randn((n, 3), generator=gen, device="cuda")
    pos = pos - torch.floor(pos + 0.5)
    del which
    x = torch.rand((n,), generator=gen, device="cuda")
    y = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
    assert y.shape == (1, N, N, N)
    rng = np.random.default_rng(778)
    freqs = rng.integers(-N // 2, N // 2, size=(32, 3))
    exact = ndft.ndft_adjoint_subset(x.cpu().numpy()[:, None], pos.cpu().numpy(), freqs)[:, 0]
    got = y[0].cpu().numpy()[tuple((freqs + N // 2).T)]
    assert rel_l2(got, exact) < T2_M4
    xh, yf, errf = _sparse_forward_check(tn, pos, N, m, 8, 779)
    assert yf.shape == (n,) and errf < T2_M4
    ops.check_status()
    lhs = torch.sum(y * xh.conj())
    rhs = torch.sum(x.to(torch.complex64) * yf.conj())
    assert abs(complex(lhs) - complex(rhs)) < 1e-4 * abs(complex(lhs)) + 1e-2


@pytest.mark.parametrize("complex_x,real_output,chunk_planes,C", [(False, True, 5, 4), (True, False, 5, 4), (True, True, 7, 4),
                                                                   (False, False, 3, 4), (False, True, 2, 3), (False, False, 4, 5)])
def test_config_c4_shape_batched_columns_chunked(tn, monkeypatch, complex_x, real_output, chunk_planes, C):
    """Config C4's structure (3-D N=128, m=4, several point sets x several columns) at a size that runs in seconds,
    with an odd chunk budget that forces the plane loop of the planar-copy column passes (chunks that start in the
    middle of a point set, (re, im) plane pairs for complex data): subset of frequencies vs the exact NDFT per
    (batch, column), sparse-spectrum forward vs the exact sums, adjointness."""
    # (round 4: the owner-computes spreading kernel sweeps the points once per PAIR of columns -- the chunks here start and
    # end inside pairs, and the odd column counts leave every point set's last column to a sweep of its own)
    N, m, B, n_per = 128, 4, 3, 20_000
    gen = torch.Generator(device="cuda").manual_seed(7)
    pos = torch.rand((B * n_per, 3), generator=gen, device="cuda") - 0.5
    batch = torch.arange(B * n_per, device="cuda") // n_per
    x = torch.randn((B * n_per, C), generator=gen, device="cuda")
    if complex_x:
        x = torch.complex(x, torch.randn((B * n_per, C), generator=gen, device="cuda"))
    monkeypatch.setenv("NFFT_HIP_CHUNK_BYTES",
                       str(chunk_planes * (256 ** 3 * 4 + 256 * 256 * 129 * 8 + 300_000_000)))
    y = tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
    assert y.shape == (B, N, N, N, C)
    rng = np.random.default_rng(8)
    freqs = rng.integers(-N // 2, N // 2, size=(64, 3))
    for b in range(B):
        sel = slice(b * n_per, (b + 1) * n_per)
        exact = ndft.ndft_adjoint_subset(x[sel].cpu().numpy(), pos[sel].cpu().numpy(), freqs)
        got = y[b].cpu().numpy()[tuple((freqs + N // 2).T)]
        assert rel_l2(got, exact) < T2_M4
    z = tn.nfft_forward(y, pos, batch, cutoff=m, real_output=real_output)
    assert z.shape == (B * n_per, C) and z.dtype == (torch.float32 if real_output else torch.complex64)
    # <A x, A x> = Re <x, A^H A x>  (a complex x needs the complex result of A^H: with real_output only its real part
    # exists, which is then compared with the real part of the complex call instead)
    lhs = float((y.abs() ** 2).sum())
    if complex_x and real_output:
        zc = tn.nfft_forward(y, pos, batch, cutoff=m, real_output=False)
        assert rel_l2(z.cpu().numpy(), zc.real.cpu().numpy()) < 2e-6
        z = zc
    rhs = float((x.conj() * z).real.sum()) if complex_x else float((x * z.real).sum())
    assert abs(lhs - rhs) < 1e-4 * lhs
    # sparse spectrum: a few non-zero (set, frequency, column) entries, exact result is a short sum per point
    xh = torch.zeros((B, N, N, N, C), dtype=torch.complex64, device="cuda")
    f = rng.integers(-N // 2, N // 2, size=(5, 3))
    vals = (rng.standard_normal((B, 5, C)) + 1j * rng.standard_normal((B, 5, C))).astype(np.complex64)
    for b in range(B):
        for k in range(5):
            xh[(b,) + tuple(f[k] + N // 2)] += torch.from_numpy(vals[b, k]).cuda()
    zf = tn.nfft_forward(xh, pos, batch, cutoff=m, real_output=real_output)
    idx = rng.integers(0, B * n_per, size=2048)
    ph = np.exp(-2j * np.pi * (pos[idx].cpu().numpy().astype(np.float64) @ f.T.astype(np.float64)))  # [pts, 5]
    exact = np.einsum("pk,pkc->pc", ph, vals[batch[idx].cpu().numpy()].astype(np.complex128))
    got = zf[idx].cpu().numpy()
    assert rel_l2(got, exact.real if real_output else exact) < T2_M4


@pytest.mark.parametrize("chunk_gib", [None, 3])
def test_config_c4_stated_shape_one_gpu_share(tn, monkeypatch, chunk_gib):
    """Config C4 at its stated shape, one GPU's share of the 8-way sharded batch: 3-D N=128, m=4, C=64 real columns,
    B=4 point sets of 10^5 points (n per set assumed, BASELINE.json gives none).  Reference
    layout [B, N, N, N, C] (docs/source/theory/dataformat.rst:43-63; column loop of the reference:
    csrc/cuda/spatial_window_operations.cu:103-171).  With the default chunk budget and with a small one (3 GiB:
    the plane loop runs ~10 times and chunks start in the middle of a point set's columns).
      * adjoint: EVERY (set, column) pair on a frequency subset vs the exact NDFT (oracle.ndft.ndft_adjoint_subset);
      * forward: sparse spectrum vs the exact sums on a sample of points, all 64 columns;
      * adjointness <A x, xh> = <x, A^H xh> at full size."""
    N, m, B, C, n_per = 128, 4, 4, 64, 100_000
    if chunk_gib is not None:
        monkeypatch.setenv("NFFT_HIP_CHUNK_BYTES", str(chunk_gib << 30))
    gen = torch.Generator(device="cuda").manual_seed(64)
    n = B * n_per
    pos = torch.rand((n, 3), generator=gen, device="cuda") - 0.5
    batch = torch.arange(n, device="cuda") // n_per
    x = torch.randn((n, C), generator=gen, device="cuda")
    y = tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
    assert y.shape == (B, N, N, N, C) and y.dtype == torch.complex64
    rng = np.random.default_rng(65)
    freqs = rng.integers(-N // 2, N // 2, size=(24, 3))
    worst = 0.0
    for b in range(B):
        sel = slice(b * n_per, (b + 1) * n_per)
        exact = ndft.ndft_adjoint_subset(x[sel].cpu().numpy(), pos[sel].cpu().numpy(), freqs)  # [24, 64]
        got = y[b].cpu().numpy()[tuple((freqs + N // 2).T)]
        assert got.shape == exact.shape == (24, C)
        for c in range(C):
            worst = max(worst, rel_l2(got[:, c], exact[:, c]))
    assert worst < T2_M4
    # forward of a sparse spectrum
    xh = torch.zeros((B, N, N, N, C), dtype=torch.complex64, device="cuda")
    f = rng.integers(-N // 2, N // 2, size=(4, 3))
    vals = (rng.standard_normal((B, 4, C)) + 1j * rng.standard_normal((B, 4, C))).astype(np.complex64)
    for b in range(B):
        for k in range(4):
            xh[(b,) + tuple(f[k] + N // 2)] += torch.from_numpy(vals[b, k]).cuda()
    z = tn.nfft_forward(xh, pos, batch, cutoff=m, real_output=True)
    assert z.shape == (n, C) and z.dtype == torch.float32
    idx = rng.integers(0, n, size=1024)
    ph = np.exp(-2j * np.pi * (pos[idx].cpu().numpy().astype(np.float64) @ f.T.astype(np.float64)))
    exact = np.einsum("pk,pkc->pc", ph, vals[batch[idx].cpu().numpy()].astype(np.complex128)).real
    got = z[idx].cpu().numpy()
    assert max(rel_l2(got[:, c], exact[:, c]) for c in range(C)) < T2_M4
    # adjointness: <A x, xh> = <x, A^H xh>; with a real x only Re(A^H xh) enters
    lhs = complex(torch.sum(y * xh.conj()))
    rhs = float(torch.sum(x * z))
    assert abs(lhs.real - rhs) < 1e-4 * abs(lhs.real) + 1e-2


def test_config_c5_fastsum_1m_x_1m(tn):
    """Gaussian kernel sums, 1e6 sources x 1e6 targets, 3-D N=256, m=4 (test_fastsum.py geometry: points in the
    ball of radius 1/4): 256 targets against the exact sums."""
    ns = nt = 1_000_000
    N, m, sigma = 256, 4, 0.1
    gen = torch.Generator(device="cuda").manual_seed(11)

    def ball(k):
        p = torch.rand((k, 3), generator=gen, device="cuda") - 0.5
        return p * (0.25 / torch.linalg.norm(p, dim=1).max())

    src, tgt = ball(ns), ball(nt)
    x = torch.rand((ns,), generator=gen, device="cuda")
    coeffs = tn.gaussian_analytic_coeffs(sigma, dim=3, N=N)
    y = tn.nfft_fastsum(x, coeffs, src, tgt, cutoff=m)
    assert y.shape == (nt,) and y.dtype == torch.float32
    sel = torch.arange(0, nt, nt // 256, device="cuda")[:256]
    d2 = ((tgt[sel, None, :].double() - src[None, :, :].double()) ** 2).sum(-1)
    exact = (torch.exp(-d2 / sigma ** 2) * x.double()[None, :]).sum(1)
    assert rel_l2(y[sel].cpu().numpy(), exact.cpu().numpy()) < 1e-3


def test_streamed_interpolation_clustered_8m(tn):
    """8e6 points in 8 tight Gaussian clusters on a 256^3 grid: the plan's work items are big enough for the streamed
    (producer / consumer) interpolation kernel, most of them are cut pieces, many chunks between clusters are
    empty (the producers skip planes nobody needs).  Forward of a sparse spectrum vs the exact sums on a sample of
    points, run twice (the kernel has no atomics: the two runs must agree bitwise), and adjointness with the adjoint."""
    N, m, n = 128, 4, 8_000_000
    gen = torch.Generator(device="cuda").manual_seed(77)
    centres = torch.rand((8, 3), generator=gen, device="cuda") - 0.5
    which = torch.randint(0, 8, (n,), generator=gen, device="cuda")
    pos = centres[which] + 0.03 * torch.randn((n, 3), generator=gen, device="cuda")
    pos = pos - torch.floor(pos + 0.5)
    rng = np.random.default_rng(78)
    f = rng.integers(-N // 2, N // 2, size=(6, 3))
    vals = (rng.standard_normal(6) + 1j * rng.standard_normal(6)).astype(np.complex64)
    xh = torch.zeros((1, N, N, N), dtype=torch.complex64, device="cuda")
    for fr, v in zip(f, vals):
        xh[(0,) + tuple(fr + N // 2)] += complex(v)
    y1 = tn.nfft_forward(xh, pos, None, cutoff=m)
    y2 = tn.nfft_forward(xh, pos, None, cutoff=m)
    assert torch.equal(y1, y2)
    idx = rng.integers(0, n, size=8192)
    p = pos[idx].cpu().numpy().astype(np.float64)
    exact = (np.exp(-2j * np.pi * (p @ f.T.astype(np.float64))) * vals[None, :].astype(np.complex128)).sum(1)
    assert rel_l2(y1[idx].cpu().numpy(), exact) < T2_M4
    x = torch.randn((n,), generator=gen, device="cuda")
    ya = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
    lhs = torch.sum(ya * xh.conj())
    rhs = torch.sum(x.to(torch.complex64) * y1.conj())
    assert abs(complex(lhs) - complex(rhs)) < 1e-4 * abs(complex(lhs)) + 1e-2


@pytest.mark.parametrize("m,nsets", [(2, 1), (5, 1), (4, 2)])
def test_streamed_interpolation_other_cutoffs(tn, m, nsets):
    """The streamed interpolation kernel with other window widths (chunks of 11 and 5 slabs instead of 7) and with two
    point sets: 6e6 / 12e6 uniform points on a 256^3 grid (big enough work items to select it); forward of a sparse
    spectrum vs the exact sums on a sample of points."""
    N = 128
    n = 6_000_000 * nsets
    gen = torch.Generator(device="cuda").manual_seed(90 + m + nsets)
    pos = torch.rand((n, 3), generator=gen, device="cuda") - 0.5
    batch = None if nsets == 1 else (torch.arange(n, device="cuda") >= n // 2).to(torch.int64)
    rng = np.random.default_rng(91 + m)
    f = rng.integers(-N // 2, N // 2, size=(5, 3))
    vals = (rng.standard_normal((nsets, 5)) + 1j * rng.standard_normal((nsets, 5))).astype(np.complex64)
    xh = torch.zeros((nsets, N, N, N), dtype=torch.complex64, device="cuda")
    for b in range(nsets):
        for fr, v in zip(f, vals[b]):
            xh[(b,) + tuple(fr + N // 2)] += complex(v)
    y = tn.nfft_forward(xh, pos, batch, cutoff=m)
    idx = rng.integers(0, n, size=8192)
    p = pos[idx].cpu().numpy().astype(np.float64)
    bsel = np.zeros(idx.shape[0], np.int64) if batch is None else batch[idx].cpu().numpy()
    exact = (np.exp(-2j * np.pi * (p @ f.T.astype(np.float64))) * vals[bsel].astype(np.complex128)).sum(1)
    tol = {2: 2e-2, 4: 5e-4, 5: 1e-4}[m]
    assert rel_l2(y[idx].cpu().numpy(), exact) < tol


def test_streamed_interpolation_is_bitwise_reproducible(tn):
    """The streamed interpolation kernel hands blocks of points to consumer waves through a queue while producer waves
    recycle a ring of 16 grid planes: whatever the timing, every block must see the planes it waits for.  The kernel has
    no atomics, so ten forward transforms of a dense random spectrum on 10^7 points (column groups active, 4 producers +
    12 consumers per workgroup) must agree bit for bit -- a slot overwritten too early or a block started too soon shows
    up as a difference -- and a sample must match the exact sums."""
    N, m, n = 256, 4, 10_000_000
    gen = torch.Generator(device="cuda").manual_seed(311)
    pos = torch.rand((n, 3), generator=gen, device="cuda") - 0.5
    xh = torch.zeros((1, N, N, N), dtype=torch.complex64, device="cuda")
    rng = np.random.default_rng(312)
    f = rng.integers(-N // 2, N // 2, size=(7, 3))
    vals = (rng.standard_normal(7) + 1j * rng.standard_normal(7)).astype(np.complex64)
    for fr, v in zip(f, vals):
        xh[(0,) + tuple(fr + N // 2)] += complex(v)
    # (a dense background on top of the seven checked frequencies would need the full NDFT as reference: the background
    # here is white noise eight orders of magnitude below them)
    noise = 1e-8 * (torch.randn((1, N, N, N), generator=gen, device="cuda") + 1j * torch.randn((1, N, N, N), generator=gen, device="cuda"))
    xh = xh + noise.to(torch.complex64)
    y0 = tn.nfft_forward(xh, pos, None, cutoff=m)
    for _ in range(9):
        assert torch.equal(tn.nfft_forward(xh, pos, None, cutoff=m), y0)
    idx = rng.integers(0, n, size=4096)
    p = pos[idx].cpu().numpy().astype(np.float64)
    exact = (np.exp(-2j * np.pi * (p @ f.T.astype(np.float64))) * vals[None, :].astype(np.complex128)).sum(1)
    assert rel_l2(y0[idx].cpu().numpy(), exact) < T2_M4 + 1e-4  # (the background adds ~1e-8 sqrt(N^3) = 4e-5)


def test_grid_2d_16384_squared(tn):
    """2-D, N = 8192 (oversampled grid of 16384^2 = 2^28 cells, 1 GiB per real plane): the largest 2-D bandwidth whose
    grid, spectra and rocFFT work area fit comfortably -- index arithmetic beyond 2^27 cells on the narrow tiling.  Adjoint
    on a frequency subset (with the band's corners) against the exact sums, forward of a sparse spectrum, adjointness."""
    N, m, n = 8192, 4, 200_000
    pos, x, y, err = _subset_check_adjoint(tn, 2, N, m, n, 96, seed=611)
    assert err < T2_M4
    corners = np.array([[-N // 2, -N // 2], [-N // 2, N // 2 - 1], [N // 2 - 1, -N // 2], [N // 2 - 1, N // 2 - 1], [0, 0]])
    exact = ndft.ndft_adjoint_subset(x.cpu().numpy()[:, None], pos.cpu().numpy(), corners)[:, 0]
    got = y[0].cpu().numpy()[tuple((corners + N // 2).T)]
    assert rel_l2(got, exact) < T2_M4
    del y
    xh, yf, errf = _sparse_forward_check(tn, pos, N, m, 6, seed=612)
    assert errf < T2_M4
    ya = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
    lhs = torch.sum(ya * xh.conj())
    rhs = torch.sum(x.to(torch.complex64) * yf.conj())
    assert abs(complex(lhs) - complex(rhs)) < 1e-4 * abs(complex(lhs)) + 1e-2


def test_work_list_tickets_on_concurrent_streams(tn):
    """Two streams run the persistent (work-list) kernels at the same time on the SAME cached plan: the ticket words that
    hand out the list live in a ring owned by the library, one set per launch -- the plan itself is read-only -- so the
    launches must not disturb each other.  4e6 clustered points (most work items are cut pieces), three rounds, results
    compared with the same transforms run one after the other."""
    N, m, n = 128, 4, 4_000_000
    gen = torch.Generator(device="cuda").manual_seed(77)
    centres = torch.rand((4, 3), generator=gen, device="cuda") - 0.5
    pos = centres[torch.randint(0, 4, (n,), generator=gen, device="cuda")] + 0.03 * torch.randn((n, 3), generator=gen, device="cuda")
    pos = pos - torch.floor(pos + 0.5)
    xa = torch.randn((n,), generator=gen, device="cuda")
    xb = torch.randn((n,), generator=gen, device="cuda")
    ya = tn.nfft_adjoint(xa, pos, None, bandwidth=N, cutoff=m)      # (also builds and caches the plan)
    yb = tn.nfft_adjoint(xb, pos, None, bandwidth=N, cutoff=m)
    fa = tn.nfft_forward(ya, pos, None, cutoff=m)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            ya2 = tn.nfft_adjoint(xa, pos, None, bandwidth=N, cutoff=m)
            fa2 = tn.nfft_forward(ya, pos, None, cutoff=m)
        with torch.cuda.stream(s2):
            yb2 = tn.nfft_adjoint(xb, pos, None, bandwidth=N, cutoff=m)
        torch.cuda.synchronize()
        # (the scatter variant's float atomics are not bitwise reproducible: compare at the parity tolerance)
        assert rel_l2(ya2.cpu().numpy(), ya.cpu().numpy()) < 2e-6
        assert rel_l2(yb2.cpu().numpy(), yb.cpu().numpy()) < 2e-6
        assert torch.equal(fa2, fa)  # the gather has no atomics
    from torch_nfft_amd import ops
    ops.check_status()


# ----------------------------------------------------------------------------- stage level at full C3 size
# The matrix-core kernels' arithmetic (f16-split operands, fp32 accumulation on the matrix cores) is pinned at 2e-6 against
# the float64 gridding on SUB-VOLUMES of the full-size problem: an MFMA truncation bias this build once had appeared only
# at 10^7 points and at 2e-5 -- below the approximation tolerance of the frequency-subset checks above.

T_STAGE = 2e-6


def _stage_lib():
    import ctypes
    from torch_nfft_amd import _lib
    return ctypes, _lib, _lib.load()


def _points_c3(kind, n, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    if kind == "uniform":
        pos = torch.rand((n, 3), generator=gen, device="cuda") - 0.5
    else:  # 8 Gaussian clusters, sigma 0.05
        centres = torch.rand((8, 3), generator=gen, device="cuda") - 0.5
        which = torch.randint(0, 8, (n,), generator=gen, device="cuda")
        pos = centres[which] + 0.05 * torch.randn((n, 3), generator=gen, device="cuda")
        pos = pos - torch.floor(pos + 0.5)
        pos = pos.clamp(-0.5, 0.49999997)
    return pos.contiguous(), torch.rand((n,), generator=gen, device="cuda") - 0.25


def _box_reference(pos_h, x_h, N, m, origin, size):
    """float64 gridding (oracle.nfft_ref.window_taps: spatial_window_operations.cu:38-97) of the points whose window
    touches the periodic box [origin, origin + size)^3 of the (2N)^3 grid, accumulated into that box only."""
    import itertools
    from oracle import nfft_ref
    M, W = 2 * N, 2 * m + 2
    cell = np.floor(pos_h.astype(np.float64) * M).astype(np.int64)
    touch = np.ones(pos_h.shape[0], dtype=bool)
    for a in range(3):
        rel = np.mod(cell[:, a] - m - origin[a], M)
        touch &= (rel < size) | (rel > M - W)
    idx = np.nonzero(touch)[0]
    shift, psi = nfft_ref.window_taps(pos_h[idx], N, m)
    box = np.zeros((size,) * 3)
    xv = x_h[idx].astype(np.float64)
    for ls in itertools.product(range(W), repeat=3):
        w = xv.copy()
        inside = np.ones(idx.size, dtype=bool)
        rel = []
        for a, l in enumerate(ls):
            w = w * psi[:, a, l]
            r = np.mod(shift[:, a] + l - origin[a], M)
            inside &= r < size
            rel.append(r)
        np.add.at(box, tuple(r[inside] for r in rel), w[inside])
    return box, idx.size


def _box_of(grid, origin, size, M):
    ix = [torch.remainder(torch.arange(o, o + size, device=grid.device), M) for o in origin]
    return grid[ix[0][:, None, None], ix[1][None, :, None], ix[2][None, None, :]].cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("kind", ["uniform", "clusters"])
def test_c3_spread_stage_subvolumes_match_float64_gridding(kind):
    """nfft_hip_plan_points + nfft_hip_spread at N=256, m=4, n=10^7: three 40^3 sub-volumes of the 512^3 grid -- interior,
    across a pencil corner of the 23 x 55 tiling, and across the periodic corner -- against the float64 gridding."""
    ctypes, _lib, lib = _stage_lib()
    N, m, n, size = 256, 4, 10_000_000, 40
    M = 2 * N
    pos, x = _points_c3(kind, n, 99)
    prob = _lib.Problem(3, n, 1, 1, N, m)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
    _lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(pos), None, p(plan), plan.numel(), s))
    grid = torch.full((M, M, M), float("nan"), device="cuda")
    scratch = torch.empty(lib.nfft_hip_spread_scratch_bytes(ctypes.byref(prob), 1) // 4 + 64, device="cuda")
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
    _lib.check(lib.nfft_hip_check_status(s, 1))
    assert bool(torch.isfinite(grid).all())
    pos_h, x_h = pos.cpu().numpy(), x.cpu().numpy()
    if kind == "clusters":  # put the interior box where the points are
        c = np.floor(np.median(pos_h[:200_000], axis=0).astype(np.float64) * M).astype(np.int64) % M
        interior = tuple(int(v) for v in c)
    else:
        interior = (200, 300, 100)
    boxes = {"interior": interior, "pencil corner": (77, 23 * 5 - 20, 55 * 4 - 20), "periodic corner": (M - 20, M - 20, M - 20)}
    for name, origin in boxes.items():
        ref, npts = _box_reference(pos_h, x_h, N, m, origin, size)
        got = _box_of(grid, origin, size, M)
        scale = np.linalg.norm(ref)
        if scale == 0.0:  # (a clustered input leaves parts of the torus empty: those cells must be exactly zero)
            assert np.abs(got).max() == 0.0, name
            continue
        err = np.linalg.norm(got - ref) / scale
        assert err < T_STAGE, (kind, name, npts, err)


def test_c3_interpolation_stage_matches_float64_gather():
    """nfft_hip_interpolate (the streamed matrix-core gather) at N=256, m=4, n=10^7 on a random grid: 4 096 of the 10^7
    results against the float64 tap sums (spatial_window_operations.cu:214-282)."""
    import itertools
    from oracle import nfft_ref
    ctypes, _lib, lib = _stage_lib()
    N, m, n = 256, 4, 10_000_000
    M, W = 2 * N, 2 * m + 2
    pos, _ = _points_c3("uniform", n, 7)
    gen = torch.Generator(device="cuda").manual_seed(8)
    grid = torch.randn((M, M, M), generator=gen, device="cuda")
    prob = _lib.Problem(3, n, 1, 1, N, m)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
    _lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(pos), None, p(plan), plan.numel(), s))
    y = torch.full((n,), float("nan"), device="cuda")
    _lib.check(lib.nfft_hip_interpolate(ctypes.byref(prob), p(plan), p(grid), 1, p(y), s))
    _lib.check(lib.nfft_hip_check_status(s, 1))
    assert bool(torch.isfinite(y).all())
    rng = np.random.default_rng(5)
    sel = np.sort(rng.choice(n, 4096, replace=False))
    shift, psi = nfft_ref.window_taps(pos[torch.from_numpy(sel).cuda()].cpu().numpy(), N, m)
    exp = np.zeros(sel.size)
    sh = torch.from_numpy(shift).cuda()
    for l0 in range(W):
        i0 = torch.remainder(sh[:, 0] + l0, M)
        for l1 in range(W):
            i1 = torch.remainder(sh[:, 1] + l1, M)
            cols = torch.remainder(sh[:, 2:3] + torch.arange(W, device="cuda")[None, :], M)
            vals = grid[i0[:, None], i1[:, None], cols].cpu().numpy().astype(np.float64)  # [4096, W] grid values, summed on the host
            exp += psi[:, 0, l0] * psi[:, 1, l1] * (vals * psi[:, 2, :]).sum(axis=1)
    got = y[torch.from_numpy(sel).cuda()].cpu().numpy().astype(np.float64)
    err = np.linalg.norm(got - exp) / np.linalg.norm(exp)
    assert err < T_STAGE, err


def test_dynamic_range_inside_one_column():
    """One coefficient column whose |x| spans 10^6 between two well-separated regions.  The operand scale of the spreading
    kernel is one power of two per (point set, column) plane, so the small region's f16 lo parts are subnormal: it keeps
    ~15 significant bits LOCALLY (relative to its own cells) where the reference's fp32 atomics keep 24; relative to the
    plane the error stays at the 2e-6 of every other test.  This test states both figures."""
    ctypes, _lib, lib = _stage_lib()
    N, m, n, size = 128, 4, 2_000_000, 32
    M = 2 * N
    gen = torch.Generator(device="cuda").manual_seed(3)
    pos = torch.rand((n, 3), generator=gen, device="cuda") - 0.5
    x = torch.rand((n,), generator=gen, device="cuda") + 0.5
    small = pos[:, 0] < 0.0                      # half of the torus along axis 0 (cells [M/2, M)) carries coefficients 10^6 times smaller
    x = torch.where(small, x * 1e-6, x).contiguous()
    prob = _lib.Problem(3, n, 1, 1, N, m)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
    _lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(pos), None, p(plan), plan.numel(), s))
    grid = torch.full((M, M, M), float("nan"), device="cuda")
    scratch = torch.empty(lib.nfft_hip_spread_scratch_bytes(ctypes.byref(prob), 1) // 4 + 64, device="cuda")
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
    _lib.check(lib.nfft_hip_check_status(s, 1))
    pos_h, x_h = pos.cpu().numpy(), x.cpu().numpy()
    # cells [M/2, M) of axis 0 hold the small region; boxes well inside each region
    big_box, small_box = (M // 4 - 16, 50, 60), (3 * M // 4 - 16, 50, 60)
    ref_b, _ = _box_reference(pos_h, x_h, N, m, big_box, size)
    ref_s, _ = _box_reference(pos_h, x_h, N, m, small_box, size)
    got_b, got_s = _box_of(grid, big_box, size, M), _box_of(grid, small_box, size, M)
    err_big = np.linalg.norm(got_b - ref_b) / np.linalg.norm(ref_b)
    err_small_local = np.linalg.norm(got_s - ref_s) / np.linalg.norm(ref_s)
    err_small_vs_plane = np.linalg.norm(got_s - ref_s) / np.linalg.norm(ref_b)
    print("in-column dynamic range 1e6: large region %.2e, small region %.2e relative to itself, %.2e relative to the plane"
          % (err_big, err_small_local, err_small_vs_plane))
    assert err_big < T_STAGE
    assert err_small_vs_plane < T_STAGE * 1e-3          # invisible next to the large region
    assert err_small_local < 2e-4                       # ~15 bits locally (reference: fp32, ~1e-7)
