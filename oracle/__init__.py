"""CPU oracle for the NFFT forward/adjoint hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (``torch_nfft_amd``)
never imports this package and fails loudly when its HIP library is missing.

Contents
--------
``ndft.py``      exact NDFT (adjoint / forward / fastsum, float64, chunked over
                 points) -- restates ``torch_nfft/ndft.py:5-62`` of the reference.
``nfft_ref.py``  float64 restatement of the reference's *algorithm* (Gaussian
                 window, oversampling 2, 2m+2 taps; spreading -> FFT ->
                 roll-off) -- restates ``csrc/cuda/spatial_window_operations.cu``,
                 ``csrc/cuda/spectral_window_operations.cu`` and the drivers in
                 ``csrc/cuda/core_cuda.cu:144-531``.
``coeffs_ref.py`` float64 restatement of the kernel-coefficient recipes (csrc/cuda/kernel_coeffs.cu).
``ndft_c.c``     plain-C (OpenMP) exact NDFT used for the CPU baseline timing;
                 built into ``oracle/_build/libndft_oracle.so`` by ``oracle/Makefile``.
``make_golden.py``  imports the reference's own ``torch_nfft/ndft.py`` (CPU,
                 this container only) and freezes seeded input/output vectors
                 into ``tests/golden/*.npz``.

Pinning: the reference ships no golden vectors (its five test scripts print
error norms of unseeded random data).  The oracle is pinned against outputs of
the reference's own exact transform ``torch_nfft/ndft.py`` run in the build
container (fixtures in ``tests/golden/``, generator ``oracle/make_golden.py``).
The reference's CUDA path itself is unbuildable here (needs nvcc + cuFFT).
"""
