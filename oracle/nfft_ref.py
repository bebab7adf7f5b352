"""float64 restatement of the reference's NFFT *algorithm* -- TEST INFRASTRUCTURE ONLY.

Same window (Gaussian, b = 4m/(3 pi)), oversampling 2 (M = 2N), 2m+2 taps per
axis, same index/sign/scale conventions as the reference's CUDA path, so that
the HIP kernels can be compared at ~1e-6 (fp32 accumulation noise) instead of
at the ~1e-4 approximation error that separates any NFFT from the exact NDFT.

Reference lines restated:
  shifts        csrc/cuda/spatial_window_operations.cu:38-61   shift = floor(pos*M) - m
  psi           csrc/cuda/spatial_window_operations.cu:24-28, 68-97
  spreading     csrc/cuda/spatial_window_operations.cu:103-211
  interpolation csrc/cuda/spatial_window_operations.cu:214-332
  phi_hat_inv   csrc/cuda/spectral_window_operations.cu:14-43
  roll-off      csrc/cuda/spectral_window_operations.cu:51-265
  FFT           csrc/cuda/core_cuda.cu:254-272 (INVERSE = e^{+}), :432-450 (FORWARD = e^{-}), unnormalised
  drivers       csrc/cuda/core_cuda.cu:144-336 (adjoint), :340-531 (forward)
"""
import itertools
import numpy as np


def _batch_info(batch, n):
    if batch is None:
        return np.zeros(n, dtype=np.int64), 1  # core_cuda.cu:62-65
    batch = np.asarray(batch).astype(np.int64)
    return batch, int(batch[-1]) + 1  # core_cuda.cu:60


def window_taps(pos, N, m):
    """shift [n,d] int64 and psi [n,d,2m+2] float64 (spatial_window_operations.cu:38-97)."""
    M = 2 * N
    W = 2 * m + 2
    p = np.asarray(pos, dtype=np.float32).astype(np.float64)  # pos*M is exact in fp32 (M power of two)
    shift = np.floor(p * M).astype(np.int64) - m
    l = np.arange(W, dtype=np.float64)
    t = (p * M - shift)[:, :, None] - l[None, None, :]
    psi = np.exp(-(t * t) * (0.75 * np.pi / m)) * np.sqrt(0.75 / m)
    return shift, psi


def phi_hat_inv(N, m):
    """exp(k^2 * pi*m/(3 N^2)), k = 0..N/2 (spectral_window_operations.cu:2-3, 14-43)."""
    k = np.arange(N // 2 + 1, dtype=np.float64)
    return np.exp(k * k * (np.pi / 3.0) * m / (N * N))


def _rolloff(N, m, d):
    """prod_k phi_hat_inv[|i_k - N/2|] on the centred [N]^d index block."""
    ph = phi_hat_inv(N, m)
    f1 = ph[np.abs(np.arange(N) - N // 2)]
    fac = np.ones((N,) * d)
    for a in range(d):
        shape = [1] * d
        shape[a] = N
        fac = fac * f1.reshape(shape)
    return fac


def _band_index(N):
    """kappa = (i - N/2) mod 2N for i = 0..N-1 (spectral_window_operations.cu:78-96)."""
    return (np.arange(N) - N // 2) % (2 * N)


def spread(x, pos, batch, N, m):
    """Adjoint gridding: g [B, C, M..M] complex128 (spatial_window_operations.cu:103-211)."""
    pos = np.asarray(pos)
    n, d = pos.shape
    M, W = 2 * N, 2 * m + 2
    x2 = np.asarray(x).reshape(n, -1).astype(np.complex128)
    C = x2.shape[1]
    bvec, B = _batch_info(batch, n)
    shift, psi = window_taps(pos, N, m)
    g = np.zeros((B, C) + (M,) * d, dtype=np.complex128)
    for ls in itertools.product(range(W), repeat=d):
        w = np.ones(n)
        idx = []
        for a, l in enumerate(ls):
            w = w * psi[:, a, l]
            idx.append((shift[:, a] + l + M) % M)
        vals = x2 * w[:, None]  # [n, C]
        for c in range(C):
            np.add.at(g, (bvec, c) + tuple(idx), vals[:, c])
    return g


def nfft_adjoint(x, pos, batch=None, N=16, m=3, real_output=False):
    """Restates nfft_adjoint_cuda (core_cuda.cu:144-336).  Returns [B, N..N, *cols]."""
    pos = np.asarray(pos)
    x = np.asarray(x)
    n, d = pos.shape
    cols = x.shape[1:]
    M = 2 * N
    g = spread(x, pos, batch, N, m)
    axes = tuple(range(2, 2 + d))
    g_hat = np.fft.ifftn(g, axes=axes) * float(M) ** d  # unnormalised e^{+}
    kap = _band_index(N)
    sub = g_hat[(slice(None), slice(None)) + np.ix_(*([kap] * d))]
    sub = sub * _rolloff(N, m, d)[None, None]
    y = np.moveaxis(sub, 1, -1)  # [B, N..N, C]
    y = y.reshape((y.shape[0],) + (N,) * d + cols)
    return y.real.copy() if real_output else y


def nfft_forward(x, pos, batch=None, m=3, real_output=False):
    """Restates nfft_forward_cuda (core_cuda.cu:340-531).  x [B, N..N, *cols] -> [n, *cols]."""
    pos = np.asarray(pos)
    x = np.asarray(x)
    n, d = pos.shape
    N = x.shape[1]
    M, W = 2 * N, 2 * m + 2
    B = x.shape[0]
    cols = x.shape[1 + d:]
    xr = x.reshape((B,) + (N,) * d + (-1,)).astype(np.complex128)
    C = xr.shape[-1]
    bvec, B2 = _batch_info(batch, n)
    assert B2 == B, "Input mismatch"
    g_hat = np.zeros((B, C) + (M,) * d, dtype=np.complex128)
    kap = _band_index(N)
    vals = np.moveaxis(xr, -1, 1) * _rolloff(N, m, d)[None, None]
    g_hat[(slice(None), slice(None)) + np.ix_(*([kap] * d))] = vals
    g = np.fft.fftn(g_hat, axes=tuple(range(2, 2 + d)))  # unnormalised e^{-}
    shift, psi = window_taps(pos, N, m)
    y = np.zeros((n, C), dtype=np.complex128)
    for ls in itertools.product(range(W), repeat=d):
        w = np.ones(n)
        idx = []
        for a, l in enumerate(ls):
            w = w * psi[:, a, l]
            idx.append((shift[:, a] + l + M) % M)
        for c in range(C):
            y[:, c] += w * g[(bvec, c) + tuple(idx)]
    y = y.reshape((n,) + cols)
    return y.real.copy() if real_output else y


def nfft_fastsum(x, coeffs, sources, targets=None, source_batch=None, target_batch=None, batch=None, m=3):
    """Restates nfft_fastsum_cuda (core_cuda.cu:535-852): spreading of the sources, inverse FFT, spectral
    multiply g_hat *= coeffs * phi_hat_inv^2 on the band / 0 elsewhere (spectral_window_operations.cu:269-402),
    forward FFT, interpolation at the targets; real part when x is real."""
    if targets is None:
        targets, target_batch = sources, source_batch
    if batch is not None:
        source_batch = target_batch = batch
    x = np.asarray(x)
    coeffs = np.asarray(coeffs)
    N = coeffs.shape[0]
    d = coeffs.ndim
    y = nfft_adjoint(x, sources, source_batch, N=N, m=m)                     # band spectrum incl. one phi_hat_inv
    y = y * coeffs.reshape((1,) + coeffs.shape + (1,) * (y.ndim - 1 - d))
    out = nfft_forward(y, targets, target_batch, m=m)                         # applies the second phi_hat_inv
    return out if np.iscomplexobj(x) else out.real
