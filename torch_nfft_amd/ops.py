"""Operator layer: the reference's ``torch.ops.torch_nfft.*`` schemas on top of the C ABI.

Mirrors ``csrc/core.cpp:43-121, 176-184`` of the reference (operator names, positional argument
order ``(pos, x, batch, ...)``, error messages) and the input checks of
``csrc/cuda/core_cuda.cu:38-115``.  PyTorch is plumbing here: it owns the device memory (outputs and
the workspace come from its caching allocator) and the stream; all arithmetic happens in
``libnfft_hip.so``.
"""
import ctypes

import torch

from . import _lib

_ws_bytes_cache = {}

# Point-plan reuse.  The tile-sorted copy of the points depends only on
# (pos, batch, N, m); adjoint <-> forward pairs (autograd backward, a forward fed by an adjoint, fastsum with
# shared points -- the reference exploits sources.is_same(targets), core_cuda.cu:552-564) reuse it instead of
# re-binning.  The cache holds ONE plan, keyed on tensor identity + version counter, so in-place edits of pos
# invalidate it.  `plan_cache_enabled(False)` turns it off; `plan_cache_clear()` drops the held plan.
_plan_cache = {"key": None, "plan": None, "enabled": True, "hits": 0, "misses": 0}


def plan_cache_enabled(flag):
    _plan_cache["enabled"] = bool(flag)
    if not flag:
        plan_cache_clear()


def plan_cache_clear():
    _plan_cache["key"] = None
    _plan_cache["plan"] = None


def plan_cache_stats():
    return {"hits": _plan_cache["hits"], "misses": _plan_cache["misses"]}


def _get_plan(prob, pos, batch, stream):
    """Returns (plan_tensor, fresh) -- the cached plan for these points or a newly built one."""
    lib = _lib.load()
    key = (pos.data_ptr(), pos._version, tuple(pos.shape), pos.device.index,
           None if batch is None else (batch.data_ptr(), batch._version), prob.batch_size, prob.N, prob.m)
    if _plan_cache["enabled"] and _plan_cache["key"] == key:
        _plan_cache["hits"] += 1
        return _plan_cache["plan"]
    nbytes = lib.nfft_hip_plan_bytes(ctypes.byref(prob))
    if nbytes < 0:
        _lib.check(_lib.EINVAL)
    plan = torch.empty(int(nbytes), dtype=torch.uint8, device=pos.device)
    _lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), _ptr(pos), _ptr(batch), _ptr(plan), plan.numel(),
                                        ctypes.c_void_p(stream)))
    _plan_cache["misses"] += 1
    if _plan_cache["enabled"]:
        # keep the tensors alive so that data_ptr identity cannot be recycled while the plan is cached
        _plan_cache["key"] = key
        _plan_cache["plan"] = plan
        _plan_cache["refs"] = (pos, batch)
    return plan


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _assert_input(cond):
    if not cond:
        raise RuntimeError("Input mismatch")  # CHECK_INPUT, csrc/cuda/cuda_utils.cu:3


def _check_points(pos, batch):
    """check_point_input (core_cuda.cu:38-66): returns (dim, n, batch_size)."""
    if not pos.is_cuda:
        raise RuntimeError("pos must be CUDA tensor")
    _assert_input(pos.dim() == 2)
    _assert_input(pos.dtype == torch.float32)
    n, dim = pos.shape
    _assert_input(1 <= dim <= 3)
    if batch is not None:
        if not batch.is_cuda:
            raise RuntimeError("(*out_batch) must be CUDA tensor")
        _assert_input(batch.dim() == 1)
        _assert_input(batch.dtype == torch.int64)
        _assert_input(batch.numel() == n)
        # the one blocking read-back of the reference (core_cuda.cu:60)
        batch_size = int(batch[-1].item()) + 1 if n > 0 else 1
        _assert_input(batch_size >= 1)
    else:
        batch_size = 1
    return dim, n, batch_size


def _is_real(x):
    if x.dtype == torch.float32:
        return True
    _assert_input(x.dtype == torch.complex64)
    return False


def _workspace(kind, prob, x_is_complex, real_output, device):
    lib = _lib.load()
    key = (kind, device.index, prob.dim, prob.num_points, prob.num_columns, prob.batch_size, prob.N, prob.m,
           x_is_complex, real_output)
    nbytes = _ws_bytes_cache.get(key)
    if nbytes is None:
        fn = lib.nfft_hip_adjoint_workspace_bytes if kind == "adjoint" else lib.nfft_hip_forward_workspace_bytes
        nbytes = fn(ctypes.byref(prob), x_is_complex, real_output)
        if nbytes < 0:
            _lib.check(_lib.EINVAL if _lib.last_error().startswith("Input mismatch") else _lib.EFFT)
        if len(_ws_bytes_cache) > 256:
            _ws_bytes_cache.clear()
        _ws_bytes_cache[key] = nbytes
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def nfft_adjoint(pos, x, batch, N, m, real_output):
    """torch_nfft::nfft_adjoint(Tensor pos, Tensor x, Tensor? batch, int N, int m, int real_output) -> Tensor
    (csrc/core.cpp:43-55; driver core_cuda.cu:144-336)."""
    if not x.is_cuda:
        raise RuntimeError("torch_nfft.nfft_adjoint is currently only implemented for GPU tensors")
    dim, n, B = _check_points(pos, batch)
    real_input = _is_real(x)  # check_spatial_coeffs_input, core_cuda.cu:69-86
    _assert_input(x.dim() >= 1)
    _assert_input(x.size(0) == n)
    C = x.numel() // n if n > 0 else int(torch.Size(x.shape[1:]).numel())
    real_output = 1 if real_output else 0
    N, m = int(N), int(m)
    y_shape = (B,) + (N,) * dim + tuple(x.shape[1:])  # core_cuda.cu:298-304
    y = torch.empty(y_shape, dtype=torch.float32 if real_output else torch.complex64, device=x.device)
    if y.numel() == 0:
        return y
    pos_c, x_c = pos.contiguous(), x.contiguous()
    batch_c = batch.contiguous() if batch is not None else None
    prob = _lib.Problem(dim, n, C, B, N, m)
    with torch.cuda.device(x.device):
        ws = _workspace("adjoint", prob, 0 if real_input else 1, real_output, x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        plan = _get_plan(prob, pos_c, batch_c, stream)
        rc = _lib.load().nfft_hip_adjoint_planned(ctypes.byref(prob), _ptr(plan), _ptr(x_c), 0 if real_input else 1,
                                                  real_output, _ptr(y), _ptr(ws), ws.numel(),
                                                  ctypes.c_void_p(stream))
    _lib.check(rc)
    return y


def nfft_forward(pos, x, batch, m, real_output):
    """torch_nfft::nfft_forward(Tensor pos, Tensor x, Tensor? batch, int m, int real_output) -> Tensor
    (csrc/core.cpp:94-105; driver core_cuda.cu:340-531)."""
    if not x.is_cuda:
        raise RuntimeError("torch_nfft.nfft_forward is currently only implemented for GPU tensors")
    dim, n, B = _check_points(pos, batch)
    real_input = _is_real(x)  # check_spectral_coeffs_input, core_cuda.cu:89-115
    _assert_input(x.dim() >= dim + 1)
    _assert_input(x.size(0) == B)
    N = x.size(1)
    _assert_input(N >= 2)
    for d in range(2, dim + 1):
        _assert_input(x.size(d) == N)
    cols = tuple(x.shape[1 + dim:])
    C = int(torch.Size(cols).numel())
    real_output = 1 if real_output else 0
    m = int(m)
    y = torch.empty((n,) + cols, dtype=torch.float32 if real_output else torch.complex64, device=x.device)
    if y.numel() == 0:
        return y
    pos_c, x_c = pos.contiguous(), x.contiguous()
    batch_c = batch.contiguous() if batch is not None else None
    prob = _lib.Problem(dim, n, C, B, N, m)
    with torch.cuda.device(x.device):
        ws = _workspace("forward", prob, 0 if real_input else 1, real_output, x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        plan = _get_plan(prob, pos_c, batch_c, stream)
        rc = _lib.load().nfft_hip_forward_planned(ctypes.byref(prob), _ptr(plan), _ptr(x_c), 0 if real_input else 1,
                                                  real_output, _ptr(y), _ptr(ws), ws.numel(),
                                                  ctypes.c_void_p(stream))
    _lib.check(rc)
    return y


def nfft_fastsum(sources, targets, x, coeffs, source_batch, target_batch, m):
    """torch_nfft::nfft_fastsum(Tensor sources, Tensor targets, Tensor x, Tensor coeffs, Tensor? source_batch,
    Tensor? target_batch, int m) -> Tensor   (csrc/core.cpp:108-121; driver core_cuda.cu:535-852).

    y = Re?[ forward_targets( coeffs * adjoint_sources(x) ) ]: the reference fuses the three steps on the
    oversampled grid; on the band spectrum they are exactly adjoint -> product with coeffs -> forward.  When
    sources and targets are the same tensor the point plan is shared (core_cuda.cu:552-564)."""
    if not x.is_cuda:
        raise RuntimeError("torch_nfft.nfft_fastsum is currently only implemented for GPU tensors")
    if not coeffs.is_cuda:
        raise RuntimeError("coeffs must be CUDA tensor")
    dim = sources.size(1) if sources.dim() == 2 else -1
    _assert_input(coeffs.dim() == dim)  # core_cuda.cu:585-590
    N = coeffs.size(0)
    for d in range(1, dim):
        _assert_input(coeffs.size(d) == N)
    real_coeffs = _is_real(coeffs)
    _assert_input(targets.dim() == 2 and targets.size(1) == dim)
    real_input = not x.is_complex()
    yhat = nfft_adjoint(sources, x, source_batch, N, m, 0)
    B = yhat.size(0)
    C = yhat.numel() // (B * N ** dim) if yhat.numel() else 0
    if yhat.numel():
        coeffs_c = coeffs.contiguous()
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _lib.check(_lib.load().nfft_hip_spectral_multiply(_ptr(yhat), _ptr(coeffs_c), 0 if real_coeffs else 1, B,
                                                              N ** dim, C, ctypes.c_void_p(stream)))
    # check_point_input(targets) happens inside; the batch sizes must agree (core_cuda.cu:566-568)
    return nfft_forward(targets, yhat, target_batch, m, 1 if real_input else 0)


def _coeff_device(device=None):
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


def gaussian_analytic_coeffs(sigma, N, dim, device=None):
    """torch_nfft::gaussian_analytic_coeffs(float sigma, int N, int dim) -> Tensor  (core_cuda.cu:855-877)."""
    device = _coeff_device(device)
    _assert_input(1 <= dim <= 3 and N >= 2)
    out = torch.empty((N,) * dim, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(_lib.load().nfft_hip_gaussian_analytic_coeffs(float(sigma), N, dim, _ptr(out), ctypes.c_void_p(stream)))
    return out


def _coeffs_ws(N, dim, device):
    nbytes = _lib.load().nfft_hip_coeffs_workspace_bytes(N, dim)
    if nbytes < 0:
        _lib.check(_lib.EINVAL if _lib.last_error().startswith("Input mismatch") else _lib.EFFT)
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def gaussian_interpolated_coeffs(sigma, N, dim, p, eps, device=None):
    """torch_nfft::gaussian_interpolated_coeffs(float sigma, int N, int dim, int p, float eps) -> Tensor
    (core_cuda.cu:880-941)."""
    device = _coeff_device(device)
    _assert_input(1 <= dim <= 3 and N >= 2)
    if p > 0:
        raise RuntimeError("Gaussian interpolated coeffs are currently only implemented for p<=0")
    if eps != 0.0:
        raise RuntimeError("Gaussian interpolated coeffs are currently only implemented for eps=0")
    out = torch.empty((N,) * dim, dtype=torch.complex64, device=device)
    with torch.cuda.device(device):
        ws = _coeffs_ws(N, dim, device)
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(_lib.load().nfft_hip_gaussian_interpolated_coeffs(float(sigma), N, dim, int(p), float(eps), _ptr(out),
                                                                     _ptr(ws), ws.numel(), ctypes.c_void_p(stream)))
    return out


def interpolation_grid(N, dim, device=None):
    """torch_nfft::interpolation_grid(int N, int dim) -> Tensor [N]*dim + [dim]  (core_cuda.cu:944-966)."""
    device = _coeff_device(device)
    _assert_input(1 <= dim <= 3 and N >= 2)
    out = torch.empty((N,) * dim + (dim,), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(_lib.load().nfft_hip_interpolation_grid(N, dim, 0, _ptr(out), ctypes.c_void_p(stream)))
    return out


def radial_interpolation_grid(N, dim, device=None):
    """torch_nfft::radial_interpolation_grid(int N, int dim) -> Tensor [N]*dim  (core_cuda.cu:969-991)."""
    device = _coeff_device(device)
    _assert_input(1 <= dim <= 3 and N >= 2)
    out = torch.empty((N,) * dim, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(_lib.load().nfft_hip_interpolation_grid(N, dim, 1, _ptr(out), ctypes.c_void_p(stream)))
    return out


def interpolated_kernel_coeffs(grid_values):
    """torch_nfft::interpolated_kernel_coeffs(Tensor grid_values) -> Tensor  (core_cuda.cu:994-1064)."""
    if not grid_values.is_cuda:
        raise RuntimeError("torch_nfft.interpolated_kernel_coeffs is currently only implemented for GPU tensors")
    dim = grid_values.dim()
    _assert_input(1 <= dim <= 3)
    N = grid_values.size(0)
    for d in range(1, dim):
        _assert_input(grid_values.size(d) == N)
    real = _is_real(grid_values)
    vals = grid_values.contiguous()
    out = torch.empty((N,) * dim, dtype=torch.complex64, device=vals.device)
    with torch.cuda.device(vals.device):
        ws = _coeffs_ws(N, dim, vals.device)
        stream = torch.cuda.current_stream(vals.device).cuda_stream
        _lib.check(_lib.load().nfft_hip_interpolated_kernel_coeffs(_ptr(vals), 0 if real else 1, N, dim, _ptr(out),
                                                                   _ptr(ws), ws.numel(), ctypes.c_void_p(stream)))
    return out


_registered = False


def register():
    """Expose the operators as ``torch.ops.torch_nfft.*`` with the reference's schemas
    (csrc/core.cpp:176-184), so code written against the reference's operator names keeps working."""
    global _registered
    if _registered:
        return
    try:
        lib = torch.library.Library("torch_nfft", "DEF")
    except RuntimeError:
        # another provider of the torch_nfft namespace (e.g. the reference itself) is already loaded
        _registered = True
        return
    lib.define("nfft_adjoint(Tensor pos, Tensor x, Tensor? batch, int N, int m, int real_output) -> Tensor")
    lib.define("nfft_forward(Tensor pos, Tensor x, Tensor? batch, int m, int real_output) -> Tensor")
    lib.define("nfft_fastsum(Tensor sources, Tensor targets, Tensor x, Tensor coeffs, Tensor? source_batch, "
               "Tensor? target_batch, int m) -> Tensor")
    lib.define("gaussian_analytic_coeffs(float sigma, int N, int dim) -> Tensor")
    lib.define("gaussian_interpolated_coeffs(float sigma, int N, int dim, int p, float eps) -> Tensor")
    lib.define("interpolation_grid(int N, int dim) -> Tensor")
    lib.define("radial_interpolation_grid(int N, int dim) -> Tensor")
    lib.define("interpolated_kernel_coeffs(Tensor grid_values) -> Tensor")
    lib.impl("nfft_adjoint", nfft_adjoint, "CompositeExplicitAutograd")
    lib.impl("nfft_forward", nfft_forward, "CompositeExplicitAutograd")
    lib.impl("nfft_fastsum", nfft_fastsum, "CompositeExplicitAutograd")
    lib.impl("gaussian_analytic_coeffs", gaussian_analytic_coeffs, "CompositeExplicitAutograd")
    lib.impl("gaussian_interpolated_coeffs", gaussian_interpolated_coeffs, "CompositeExplicitAutograd")
    lib.impl("interpolation_grid", interpolation_grid, "CompositeExplicitAutograd")
    lib.impl("radial_interpolation_grid", radial_interpolation_grid, "CompositeExplicitAutograd")
    lib.impl("interpolated_kernel_coeffs", interpolated_kernel_coeffs, "CompositeExplicitAutograd")
    register._lib = lib  # keep alive
    _registered = True
