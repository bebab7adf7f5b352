"""Kernel coefficient helpers with the reference's Python signatures (``torch_nfft/coeffs.py:10-27``).
The work happens in ``libnfft_hip.so`` (``csrc/coeffs.hip``)."""
from . import ops


def gaussian_analytic_coeffs(sigma, dim=3, N=16):
    """b_l = prod_d sqrt(pi) sigma exp(-sigma^2 pi^2 l_d^2): Fourier coefficients of exp(-|z|^2 / sigma^2)."""
    return ops.gaussian_analytic_coeffs(sigma, N, dim)


def gaussian_interpolated_coeffs(sigma, dim=3, N=16, p=-1, eps=0.0):
    """Coefficients of the trigonometric interpolant of the Gaussian on the grid k/N - 1/2 (p < 0), or of the
    Gaussian clipped to a constant outside radius 1/2 (p == 0)."""
    return ops.gaussian_interpolated_coeffs(sigma, N, dim, p, eps)


def interpolation_grid(dim=3, N=16):
    return ops.interpolation_grid(N, dim)


def radial_interpolation_grid(dim=3, N=16):
    return ops.radial_interpolation_grid(N, dim)


def interpolated_kernel_coeffs(grid_values):
    """Coefficients of the trigonometric interpolant of user-supplied kernel samples on ``interpolation_grid``."""
    return ops.interpolated_kernel_coeffs(grid_values)
