"""Drop-in package name: ``import torch_nfft`` resolves to the MI355X-native implementation in ``torch_nfft_amd``
(whose import loads ``libnfft_hip.so`` and the operator registry ``core.so``).  Everything the reference's
``torch_nfft/__init__.py:14-20`` exports is available under the same names."""
from torch_nfft_amd import *  # noqa: F401,F403
from torch_nfft_amd import __all__  # noqa: F401
from torch_nfft_amd import coeffs, kernel, matrices, ndft, nfft, utils  # noqa: F401
