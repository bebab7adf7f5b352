"""torch_nfft_amd -- MI355X-native NFFT forward/adjoint behind the torch_nfft API.

Importing the package loads the HIP C-ABI library ``libnfft_hip.so`` (the reference loads its
``core.so`` the same way, ``torch_nfft/__init__.py:11``) and registers ``torch.ops.torch_nfft.*``.
There is no CPU fallback: a missing library is an ImportError.
"""
from . import _lib

_lib.load()

from . import ops  # noqa: E402
from .nfft import nfft_adjoint, nfft_forward, NfftAdjointFunction, NfftForwardFunction  # noqa: E402

ops.register()

__all__ = ["nfft_adjoint", "nfft_forward", "NfftAdjointFunction", "NfftForwardFunction"]
