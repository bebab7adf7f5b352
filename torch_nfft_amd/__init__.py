"""torch_nfft_amd -- MI355X-native NFFT behind the torch_nfft API.

Importing the package loads the HIP C-ABI library ``libnfft_hip.so`` and, with ``torch.ops.load_library``, the
native operator registry ``core.so`` that exports ``torch.ops.torch_nfft.*`` (the reference loads its ``core.so``
the same way, ``torch_nfft/__init__.py:11``).  There is no CPU fallback: a missing library is an ImportError.
"""
from . import _lib

_lib.load()
_lib.load_core()

from . import ops  # noqa: E402
from .nfft import (nfft_adjoint, nfft_forward, nfft_fastsum, NfftAdjointFunction, NfftForwardFunction,  # noqa: E402
                   NfftFastsumFunction)
from .ndft import ndft_forward, ndft_adjoint, ndft_fastsum, exact_trigonometric_matrix, exact_gaussian_matrix  # noqa: E402
from .coeffs import (gaussian_analytic_coeffs, gaussian_interpolated_coeffs, interpolation_grid,  # noqa: E402
                     radial_interpolation_grid, interpolated_kernel_coeffs)
from .matrices import GramMatrix, AdjacencyMatrix  # noqa: E402
from .kernel import GaussianKernel  # noqa: E402
from . import utils  # noqa: E402


__all__ = ["nfft_adjoint", "nfft_forward", "nfft_fastsum", "ndft_forward", "ndft_adjoint", "ndft_fastsum",
           "exact_trigonometric_matrix", "exact_gaussian_matrix", "NfftAdjointFunction", "NfftForwardFunction",
           "NfftFastsumFunction", "gaussian_analytic_coeffs", "gaussian_interpolated_coeffs", "interpolation_grid",
           "radial_interpolation_grid", "interpolated_kernel_coeffs", "GramMatrix", "AdjacencyMatrix",
           "GaussianKernel", "utils"]
