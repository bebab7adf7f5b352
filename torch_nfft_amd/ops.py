"""Operator layer: thin Python names for the native ``torch.ops.torch_nfft.*`` operators.

The eight operators of the reference (``csrc/core.cpp:43-121, 176-184``: names, positional argument order
``(pos, x, batch, ...)``, error messages, the input checks of ``csrc/cuda/core_cuda.cu:38-115``) are registered
from C++ by ``core.so`` (``csrc/core.cpp`` of this package), which the package ``__init__`` loads with
``torch.ops.load_library`` exactly like the reference's ``torch_nfft/__init__.py:11``.  This module only gives
them Python names, adds the ``device=`` convenience of the coefficient helpers and exposes the point-plan cache
controls.  All arithmetic happens in ``libnfft_hip.so`` behind the C ABI of ``include/nfft_hip.h``.

Point-plan reuse.  The tile-sorted copy of the points depends only on
``(pos, batch, N, m)``; adjoint <-> forward pairs on the same points (autograd backward, a forward fed by an
adjoint, fastsum) reuse it instead of re-binning.  The cache lives in ``core.so``: two entries, keyed on tensor
identity + version counter, stream-aware (a plan built on one stream is waited for and recorded on the consuming
stream).  Writes that bypass the version counter -- ``pos.data.copy_()``, kernels of other libraries, DLPack aliases --
are invisible to the KEY, so every plan carries a checksum of the ``pos`` / ``batch`` it was built from and every hit
re-checks it (one streaming pass, ~25 us for 10^7 points): a stale plan raises "stale point plan" at the next operator or
``check_status()`` instead of returning a transform of points that are no longer there (``plan_cache_verify(False)`` for
callers who never write that way).  The same switches govern the remembered ENDS of the batch vector
(``B = batch[-1] + 1`` otherwise costs a blocking read-back of ~35 us in every operator call, as in the reference's
``check_point_input``, core_cuda.cu:60): same key (identity + version counter), same limitation.
"""
import torch

_ops = torch.ops.torch_nfft


def plan_cache_enabled(flag):
    _ops._plan_cache(1 if flag else 2)


def plan_cache_clear():
    _ops._plan_cache(0)


def plan_cache_verify(flag):
    """Verify a cached plan's seal (the checksum of the points it was built from) on every hit (default on)."""
    _ops._plan_cache(5 if flag else 6)


def plan_cache_stats():
    return {"hits": int(_ops._plan_cache(3)), "misses": int(_ops._plan_cache(4))}


def check_status(synchronize=True):
    """Raises ``RuntimeError`` if a kernel on the current device has reported a fault since the last look (a bounded
    wait of the streamed interpolation kernel ran out; a batch index outside ``[0, B)``).  The operators are
    asynchronous, so such a fault otherwise surfaces in the NEXT operator call on the device; with ``synchronize``
    (default) the current stream is drained first and the work enqueued so far is covered.  The reference aborts the
    process on a device error (``csrc/cuda/cuda_utils.cu:5-16``)."""
    _ops._check_status(1 if synchronize else 0)


def nfft_adjoint(pos, x, batch, N, m, real_output):
    """torch_nfft::nfft_adjoint(Tensor pos, Tensor x, Tensor? batch, int N, int m, int real_output) -> Tensor
    (csrc/core.cpp:43-55; driver core_cuda.cu:144-336)."""
    return _ops.nfft_adjoint(pos, x, batch, int(N), int(m), 1 if real_output else 0)


def nfft_forward(pos, x, batch, m, real_output):
    """torch_nfft::nfft_forward(Tensor pos, Tensor x, Tensor? batch, int m, int real_output) -> Tensor
    (csrc/core.cpp:94-105; driver core_cuda.cu:340-531)."""
    return _ops.nfft_forward(pos, x, batch, int(m), 1 if real_output else 0)


def nfft_fastsum(sources, targets, x, coeffs, source_batch, target_batch, m):
    """torch_nfft::nfft_fastsum(Tensor sources, Tensor targets, Tensor x, Tensor coeffs, Tensor? source_batch,
    Tensor? target_batch, int m) -> Tensor   (csrc/core.cpp:108-121; driver core_cuda.cu:535-852).
    One native call (``nfft_hip_fastsum_planned``): adjoint at the sources with the kernel coefficients folded
    into its last spectral pass, forward at the targets; shared points share one plan (core_cuda.cu:552-564)."""
    return _ops.nfft_fastsum(sources, targets, x, coeffs, source_batch, target_batch, int(m))


class _on_device:
    """The coefficient operators create their output on the current device (like the reference, which has no
    device argument); ``device=`` selects it for the duration of the call."""

    def __init__(self, device):
        self.ctx = torch.cuda.device(torch.device(device)) if device is not None else None

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def gaussian_analytic_coeffs(sigma, N, dim, device=None):
    """torch_nfft::gaussian_analytic_coeffs(float sigma, int N, int dim) -> Tensor  (core_cuda.cu:855-877)."""
    with _on_device(device):
        return _ops.gaussian_analytic_coeffs(float(sigma), int(N), int(dim))


def gaussian_interpolated_coeffs(sigma, N, dim, p, eps, device=None):
    """torch_nfft::gaussian_interpolated_coeffs(float sigma, int N, int dim, int p, float eps) -> Tensor
    (core_cuda.cu:880-941)."""
    with _on_device(device):
        return _ops.gaussian_interpolated_coeffs(float(sigma), int(N), int(dim), int(p), float(eps))


def interpolation_grid(N, dim, device=None):
    """torch_nfft::interpolation_grid(int N, int dim) -> Tensor [N]*dim + [dim]  (core_cuda.cu:944-966)."""
    with _on_device(device):
        return _ops.interpolation_grid(int(N), int(dim))


def radial_interpolation_grid(N, dim, device=None):
    """torch_nfft::radial_interpolation_grid(int N, int dim) -> Tensor [N]*dim  (core_cuda.cu:969-991)."""
    with _on_device(device):
        return _ops.radial_interpolation_grid(int(N), int(dim))


def interpolated_kernel_coeffs(grid_values):
    """torch_nfft::interpolated_kernel_coeffs(Tensor grid_values) -> Tensor  (core_cuda.cu:994-1064)."""
    return _ops.interpolated_kernel_coeffs(grid_values)
