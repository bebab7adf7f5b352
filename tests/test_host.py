"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/nfft_hip.h
declares, argument validation (no compute calls: there is no GPU here), the operator layer's input
checks and error messages, and the sharding arithmetic."""
import ctypes
import os
import re

import numpy as np
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
    # a single point set belongs to the rank batch_range gives it to -- the last one -- and so do its points (they sat on
    # rank 0 until round 4: no rank transformed anything)
    assert d.point_bounds(None, 1, 4, 7) == [0, 0, 0, 0, 7]
    assert [d.batch_range(1, r, 4) for r in range(4)] == [(0, 0), (0, 0), (0, 0), (0, 1)]


def test_operator_registry_is_native():
    """torch.ops.torch_nfft.* comes from core.so (TORCH_LIBRARY in csrc/core.cpp), all eight schemas of the
    reference's csrc/core.cpp:176-184, and no Python kernel stands behind any of them."""
    import torch_nfft_amd  # noqa: F401
    from torch_nfft_amd import _lib
    assert any(os.path.realpath(p) == os.path.realpath(_lib.CORE_PATH) for p in torch.ops.loaded_libraries)
    expect = {
        "nfft_fastsum": "(Tensor sources, Tensor targets, Tensor x, Tensor coeffs, Tensor? source_batch, "
                        "Tensor? target_batch, int m) -> Tensor",
        "gaussian_analytic_coeffs": "(float sigma, int N, int dim) -> Tensor",
        "gaussian_interpolated_coeffs": "(float sigma, int N, int dim, int p, float eps) -> Tensor",
        "interpolation_grid": "(int N, int dim) -> Tensor",
        "radial_interpolation_grid": "(int N, int dim) -> Tensor",
        "interpolated_kernel_coeffs": "(Tensor grid_values) -> Tensor",
    }
    for name, sig in expect.items():
        op = getattr(torch.ops.torch_nfft, name).default
        assert str(op._schema) == "torch_nfft::" + name + sig
        assert not op.py_kernels
    with pytest.raises(RuntimeError, match="torch_nfft.nfft_fastsum is currently only implemented for GPU tensors"):
        torch.ops.torch_nfft.nfft_fastsum(torch.zeros(3, 2), torch.zeros(3, 2), torch.zeros(3), torch.zeros(8, 8),
                                          None, None, 3)


def test_drop_in_package_name_and_exact_transforms():
    """`import torch_nfft` gives this implementation; its ndft_* / exact_* helpers (pure torch, any device) reproduce
    the golden vectors frozen from the reference's torch_nfft/ndft.py."""
    import torch_nfft
    import torch_nfft_amd as tn
    from conftest import load_golden, rel_l2
    assert torch_nfft.nfft_adjoint is tn.nfft_adjoint
    t = torch.from_numpy
    g = load_golden("g1_adjoint_2d_batched")
    y = torch_nfft.ndft_adjoint(t(g["x"]), t(g["pos"]), t(g["batch"]), N=16)
    assert y.dtype == torch.complex64 and rel_l2(y.numpy(), g["y_adjoint"]) < 5e-6
    g = load_golden("g4_3d_ragged")
    assert rel_l2(tn.ndft_adjoint(t(g["x_complex"]), t(g["pos"]), t(g["batch"]), N=16).numpy(), g["y_adjoint_complex"]) < 5e-6
    assert rel_l2(tn.ndft_forward(t(g["xhat"]), t(g["pos"]), t(g["batch"])).numpy(), g["y_forward"]) < 5e-6
    g = load_golden("g3_1d_n64")
    assert rel_l2(tn.ndft_forward(t(g["xhat"]), t(g["pos"])).numpy(), g["y_forward"]) < 5e-6
    g = load_golden("g6_fastsum_2d")
    y = tn.ndft_fastsum(t(g["x"]), t(g["coeffs"]), t(g["pos"]))
    assert y.dtype == torch.float32 and rel_l2(y.numpy(), g["y_fastsum"]) < 5e-6
    assert rel_l2(tn.exact_trigonometric_matrix(t(g["coeffs"]), t(g["pos"])).numpy(), g["exact_trig"]) < 5e-6
    sigma = float(np.sqrt(-1.0 / np.log(max(g["exact_gauss"][0, 1], 1e-30)) * np.sum((g["pos"][0] - g["pos"][1]) ** 2)))
    assert rel_l2(tn.exact_gaussian_matrix(sigma, t(g["pos"])).numpy(), g["exact_gauss"]) < 1e-4
    # batched: block-diagonal matrices
    pos = torch.rand(7, 2) * 0.5 - 0.25
    batch = torch.tensor([0, 0, 0, 1, 1, 2, 2])
    A = tn.exact_gaussian_matrix(0.3, pos, batch=batch)
    assert A.shape == (7, 7) and float(A[0, 3]) == 0.0 and float(A[5, 6]) > 0.0


def test_bench_self_launches_ranks_before_touching_the_gpu(monkeypatch):
    """`bench.py --gpus N` outside a launcher becomes the parent of a torch.distributed.run job with N ranks on
    127.0.0.1 (and exits with the job's code); under a launcher (WORLD_SIZE set) it does not spawn."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"] = cmd
        seen["env"] = env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7  # a failed worker fails the bench
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
