"""float64 restatement of the reference's kernel-coefficient recipes -- TEST INFRASTRUCTURE ONLY.

Reference (CUDA only, so restated from the source, not importable):
  gaussian_analytic_coeffs      csrc/cuda/kernel_coeffs.cu:6-30
  gaussian_interpolated_coeffs  csrc/cuda/kernel_coeffs.cu:33-73, 179-202; driver core_cuda.cu:880-941
  interpolation_grid            csrc/cuda/kernel_coeffs.cu:76-97
  radial_interpolation_grid     csrc/cuda/kernel_coeffs.cu:99-123
  interpolated_kernel_coeffs    csrc/cuda/kernel_coeffs.cu:126-202; driver core_cuda.cu:994-1064
Pinned indirectly: the reference's own test (test/test_fastsum.py) checks that fastsum with these coefficients
reproduces the Gaussian kernel matrix, which tests/test_gpu_fastsum.py asserts against
oracle.ndft / the golden exact matrices.
"""
import numpy as np


def _axes(N, dim):
    l = np.arange(-N // 2, N // 2, dtype=np.float64)
    return np.meshgrid(*([l] * dim), indexing="ij")


def gaussian_analytic_coeffs(sigma, dim=3, N=16):
    out = np.ones((N,) * dim)
    for l in _axes(N, dim):
        out = out * (np.sqrt(np.pi) * sigma * np.exp(-(sigma * np.pi * l) ** 2))
    return out


def interpolation_grid(dim=3, N=16):
    return np.stack([l / N for l in _axes(N, dim)], axis=-1)  # (k_c / N - 1/2) with k = l + N/2


def radial_interpolation_grid(dim=3, N=16):
    return np.sqrt((interpolation_grid(dim, N) ** 2).sum(-1))


def interpolated_kernel_coeffs(grid_values):
    """fftshift(FFT(ifftshift(values))) / N^d  (kernel_coeffs.cu:126-202: b[(k + N/2) % N] = values[k])."""
    v = np.asarray(grid_values)
    return np.fft.fftshift(np.fft.fftn(np.fft.ifftshift(v))) / v.size


def gaussian_interpolated_coeffs(sigma, dim=3, N=16, p=-1, eps=0.0):
    assert p <= 0 and eps == 0.0
    r2 = radial_interpolation_grid(dim, N) ** 2
    if p < 0:
        vals = np.exp(-r2 / sigma ** 2)
    else:
        vals = np.where(r2 <= 0.25, np.exp(-r2 / sigma ** 2), np.exp(-0.25 / sigma ** 2))
    return interpolated_kernel_coeffs(vals)
