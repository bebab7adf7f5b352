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
    lib.impl("nfft_adjoint", nfft_adjoint, "CompositeExplicitAutograd")
    lib.impl("nfft_forward", nfft_forward, "CompositeExplicitAutograd")
    register._lib = lib  # keep alive
    _registered = True
