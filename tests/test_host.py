"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/nfft_hip.h
declares, argument validation (no compute calls: there is no GPU here), the operator layer's input
checks and error messages, and the sharding arithmetic."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nfft_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nfft_hip_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from torch_nfft_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 12
    for name in syms:
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == syms
    assert _lib.load().nfft_hip_abi_version() == _lib.ABI_VERSION


def test_problem_validation_without_gpu():
    from torch_nfft_amd import _lib
    lib = _lib.load()
    ok = _lib.Problem(3, 1000, 2, 4, 16, 4)
    assert lib.nfft_hip_plan_bytes(ctypes.byref(ok)) > 1000 * 4 * 4
    for bad in [_lib.Problem(0, 10, 1, 1, 16, 3), _lib.Problem(4, 10, 1, 1, 16, 3), _lib.Problem(2, 10, 1, 1, 15, 3),
                _lib.Problem(2, 10, 1, 1, 16, 0), _lib.Problem(2, 10, 1, 1, 16, 9), _lib.Problem(2, 10, 1, 0, 16, 3),
                _lib.Problem(2, -1, 1, 1, 16, 3), _lib.Problem(1, 10, 1, 1, 2, 2),
                _lib.Problem(3, 10, 1, 100000, 256, 4)]:  # (point set, tile) bins beyond 32-bit indices
        assert lib.nfft_hip_plan_bytes(ctypes.byref(bad)) == -1
        assert _lib.last_error().startswith("Input mismatch")
    # error mapping: the reference raises RuntimeError("Input mismatch") (csrc/cuda/cuda_utils.cu:3)
    with pytest.raises(RuntimeError, match="Input mismatch"):
        _lib.check(_lib.EINVAL)
    # compute entry points refuse a null / short workspace before touching the device
    rc = lib.nfft_hip_plan_points(ctypes.byref(ok), None, None, None, 0, None)
    assert rc == _lib.EWORKSPACE


def test_operator_layer_rejects_cpu_and_bad_inputs():
    import torch_nfft_amd as tn
    x = torch.zeros(5)
    pos = torch.zeros(5, 2)
    with pytest.raises(RuntimeError, match="torch_nfft.nfft_adjoint is currently only implemented for GPU tensors"):
        tn.nfft_adjoint(x, pos)
    with pytest.raises(RuntimeError, match="torch_nfft.nfft_forward is currently only implemented for GPU tensors"):
        tn.nfft_forward(torch.zeros(1, 8, 8), pos)
    # schemas registered under the reference's operator namespace with the reference's argument order
    s = str(torch.ops.torch_nfft.nfft_adjoint.default._schema)
    assert s.startswith("torch_nfft::nfft_adjoint(Tensor pos, Tensor x, Tensor? batch, int N, int m, int real_output)")
    s = str(torch.ops.torch_nfft.nfft_forward.default._schema)
    assert s.startswith("torch_nfft::nfft_forward(Tensor pos, Tensor x, Tensor? batch, int m, int real_output)")


def test_signatures_match_reference():
    import inspect
    import torch_nfft_amd as tn
    assert str(inspect.signature(tn.nfft_adjoint)) == "(x, pos, batch=None, bandwidth=16, cutoff=3, real_output=False)"
    assert str(inspect.signature(tn.nfft_forward)) == "(x, pos, batch=None, cutoff=3, real_output=False)"


def test_batch_sharding_arithmetic():
    from torch_nfft_amd import distributed as d
    for B in (1, 2, 5, 8, 32, 33):
        for world in (1, 2, 3, 8):
            rs = [d.batch_range(B, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == B
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(b1 - b0 for b0, b1 in rs) - min(b1 - b0 for b0, b1 in rs) <= 1
    batch = torch.tensor([0, 0, 1, 1, 1, 3, 3, 4])
    assert d.point_bounds(batch, 5, 2, 8) == [0, 5, 8]
    assert d.point_bounds(None, 1, 4, 7) == [0, 7, 7, 7, 7]
