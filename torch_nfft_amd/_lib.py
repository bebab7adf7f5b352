"""Loader for the two native libraries: the C-ABI library ``libnfft_hip.so`` (declared in
``include/nfft_hip.h``; bound here with ctypes for the stage-level entry points, the timers and non-torch style
tests) and the torch operator registry ``core.so`` built on top of it (``csrc/core.cpp``), which is loaded with
``torch.ops.load_library`` like the reference's (``torch_nfft/__init__.py:11``).

The product path has no CPU fallback: if a library is missing or lacks a symbol the import fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NFFT_HIP_LIB") or os.path.join(_HERE, "libnfft_hip.so")
CORE_PATH = os.path.join(_HERE, "core.so")

ABI_VERSION = 4
POINTS_IN_QUARTER_BALL = 1

# every symbol include/nfft_hip.h declares
SYMBOLS = (
    "nfft_hip_abi_version",
    "nfft_hip_last_error",
    "nfft_hip_check_status",
    "nfft_hip_adjoint_workspace_bytes",
    "nfft_hip_forward_workspace_bytes",
    "nfft_hip_adjoint",
    "nfft_hip_forward",
    "nfft_hip_adjoint_planned",
    "nfft_hip_forward_planned",
    "nfft_hip_plan_bytes",
    "nfft_hip_plan_points",
    "nfft_hip_plan_verify",
    "nfft_hip_spread_scratch_bytes",
    "nfft_hip_spread",
    "nfft_hip_interpolate",
    "nfft_hip_spectral_multiply",
    "nfft_hip_fastsum_workspace_bytes",
    "nfft_hip_fastsum",
    "nfft_hip_fastsum_planned",
    "nfft_hip_gaussian_analytic_coeffs",
    "nfft_hip_interpolation_grid",
    "nfft_hip_coeffs_workspace_bytes",
    "nfft_hip_gaussian_interpolated_coeffs",
    "nfft_hip_interpolated_kernel_coeffs",
    "nfft_hip_plan_needed",
    "nfft_hip_profile_enable",
    "nfft_hip_profile_stages",
    "nfft_hip_profile_collect",
)
STAGES = ("plan", "gather", "zero", "spread", "fft", "rolloff", "interp")

OK, EINVAL, EWORKSPACE, EFFT, EHIP, EKERNEL = 0, 1, 2, 3, 4, 5


class Problem(ctypes.Structure):
    """``nfft_hip_problem`` of include/nfft_hip.h."""
    _fields_ = [
        ("dim", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("num_points", ctypes.c_int64),
        ("num_columns", ctypes.c_int64),
        ("batch_size", ctypes.c_int64),
        ("N", ctypes.c_int64),
        ("m", ctypes.c_int64),
    ]


    def __init__(self, dim=0, num_points=0, num_columns=0, batch_size=1, N=0, m=0, flags=0):
        super().__init__(dim, flags, num_points, num_columns, batch_size, N, m)


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it brings its own libamdhip64, and the library must bind to THAT runtime (the one that owns the
    # tensors' device context).  Loaded the other way round, libnfft_hip.so pulls in the system ROCm runtime and its
    # first HIP call fails ("hipGetDevice failed").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "torch_nfft_amd: %s is missing -- build it with `python torch_nfft_amd/build.py` "
            "(there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name in SYMBOLS:
        if not hasattr(lib, name):
            raise ImportError("torch_nfft_amd: %s does not export %s" % (LIB_PATH, name))
    P = ctypes.POINTER(Problem)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.nfft_hip_abi_version.restype = ci
    lib.nfft_hip_last_error.restype = ctypes.c_char_p
    lib.nfft_hip_check_status.argtypes = [vp, ci]
    lib.nfft_hip_check_status.restype = ci
    for f in (lib.nfft_hip_adjoint_workspace_bytes, lib.nfft_hip_forward_workspace_bytes):
        f.argtypes = [P, ci, ci]
        f.restype = i64
    for f in (lib.nfft_hip_adjoint, lib.nfft_hip_forward):
        f.argtypes = [P, vp, vp, ci, vp, ci, vp, vp, i64, vp]
        f.restype = ci
    for f in (lib.nfft_hip_adjoint_planned, lib.nfft_hip_forward_planned):
        f.argtypes = [P, vp, vp, ci, ci, vp, vp, i64, vp]
        f.restype = ci
    lib.nfft_hip_plan_needed.argtypes = [P]
    lib.nfft_hip_plan_needed.restype = ci
    lib.nfft_hip_plan_bytes.argtypes = [P]
    lib.nfft_hip_plan_bytes.restype = i64
    lib.nfft_hip_plan_points.argtypes = [P, vp, vp, vp, i64, vp]
    lib.nfft_hip_plan_points.restype = ci
    lib.nfft_hip_spread_scratch_bytes.argtypes = [P, i64]
    lib.nfft_hip_spread_scratch_bytes.restype = i64
    lib.nfft_hip_spread.argtypes = [P, vp, vp, i64, vp, vp, vp]
    lib.nfft_hip_spread.restype = ci
    lib.nfft_hip_plan_verify.argtypes = [P, vp, vp, vp, vp]
    lib.nfft_hip_plan_verify.restype = ci
    lib.nfft_hip_interpolate.argtypes = [P, vp, vp, i64, vp, vp]
    lib.nfft_hip_interpolate.restype = ci
    lib.nfft_hip_spectral_multiply.argtypes = [vp, vp, ci, i64, i64, i64, vp]
    lib.nfft_hip_spectral_multiply.restype = ci
    lib.nfft_hip_fastsum_workspace_bytes.argtypes = [P, P, ci, ci, ci]
    lib.nfft_hip_fastsum_workspace_bytes.restype = i64
    lib.nfft_hip_fastsum.argtypes = [P, vp, vp, P, vp, vp, vp, ci, vp, ci, vp, vp, i64, vp]
    lib.nfft_hip_fastsum.restype = ci
    lib.nfft_hip_fastsum_planned.argtypes = [P, vp, P, vp, vp, ci, vp, ci, vp, vp, i64, vp]
    lib.nfft_hip_fastsum_planned.restype = ci
    lib.nfft_hip_gaussian_analytic_coeffs.argtypes = [ctypes.c_double, i64, ctypes.c_int32, vp, vp]
    lib.nfft_hip_gaussian_analytic_coeffs.restype = ci
    lib.nfft_hip_interpolation_grid.argtypes = [i64, ctypes.c_int32, ci, vp, vp]
    lib.nfft_hip_interpolation_grid.restype = ci
    lib.nfft_hip_coeffs_workspace_bytes.argtypes = [i64, ctypes.c_int32]
    lib.nfft_hip_coeffs_workspace_bytes.restype = i64
    lib.nfft_hip_gaussian_interpolated_coeffs.argtypes = [ctypes.c_double, i64, ctypes.c_int32, i64, ctypes.c_double,
                                                          vp, vp, i64, vp]
    lib.nfft_hip_gaussian_interpolated_coeffs.restype = ci
    lib.nfft_hip_interpolated_kernel_coeffs.argtypes = [vp, ci, i64, ctypes.c_int32, vp, vp, i64, vp]
    lib.nfft_hip_interpolated_kernel_coeffs.restype = ci
    lib.nfft_hip_profile_enable.argtypes = [ci]
    lib.nfft_hip_profile_enable.restype = None
    lib.nfft_hip_profile_stages.argtypes = [ctypes.c_uint]
    lib.nfft_hip_profile_stages.restype = None
    lib.nfft_hip_profile_collect.argtypes = [vp, vp, ci]
    lib.nfft_hip_profile_collect.restype = ci
    if lib.nfft_hip_abi_version() != ABI_VERSION:
        raise ImportError("torch_nfft_amd: ABI version mismatch in %s" % LIB_PATH)
    _lib = lib
    return lib


_core_loaded = False


def load_core():
    """Registers ``torch.ops.torch_nfft.*`` from ``core.so`` (after libnfft_hip.so, which it links against: the
    library loaded above is the instance it binds to)."""
    global _core_loaded
    if _core_loaded:
        return
    load()
    import torch
    if not os.path.exists(CORE_PATH):
        raise ImportError(
            "torch_nfft_amd: %s is missing -- build it with `python torch_nfft_amd/build.py`" % CORE_PATH)
    try:
        torch.ops.load_library(CORE_PATH)
    except Exception as e:  # a second provider of the torch_nfft namespace, or an unresolved symbol
        raise ImportError("torch_nfft_amd: cannot load %s: %s" % (CORE_PATH, e)) from e
    _core_loaded = True


def profile_enable(on, stages=None):
    """Stage timers on / off; ``stages`` = names of the stages to time (default: all of STAGES)."""
    mask = 0xFFFFFFFF if stages is None else sum(1 << STAGES.index(s) for s in stages)
    load().nfft_hip_profile_stages(mask)
    load().nfft_hip_profile_enable(1 if on else 0)


def profile_collect():
    """{stage: (total_ms, launches)} since the last collect (waits for the recorded events)."""
    n = len(STAGES)
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_int64 * n)()
    check(load().nfft_hip_profile_collect(ms, cnt, n))
    return {STAGES[i]: (float(ms[i]), int(cnt[i])) for i in range(n)}


def check_status(stream=None, synchronize=True):
    """Raise if a kernel on the current device has reported a fault (``nfft_hip_check_status``); with ``synchronize``
    the stream (default: torch's current one) is drained first, so that work just enqueued is covered."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    check(load().nfft_hip_check_status(ctypes.c_void_p(stream), 1 if synchronize else 0))


def last_error():
    return load().nfft_hip_last_error().decode("utf-8", "replace")


def check(rc):
    """Map a C-ABI return code onto the exceptions the reference raises (RuntimeError)."""
    if rc == OK:
        return
    msg = last_error()
    if rc == EINVAL and not msg.startswith("Input mismatch"):
        msg = "Input mismatch: " + msg
    raise RuntimeError(msg)
