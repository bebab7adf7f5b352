"""GPU parity of the fused 1-D path (csrc/small1d.hip): 1-D transforms whose oversampled grid fits one workgroup's LDS
run as ONE kernel per direction on the caller's points, without a point plan.  Checked against the oracle
(oracle/nfft_ref.py = the reference's algorithm in float64; oracle/ndft.py = the exact sums) and against the general path
(point plan -> spreading -> rocFFT -> roll-off) through the planned C entry points on the same inputs."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import ndft, nfft_ref

pytestmark = pytest.mark.gpu

T1N = 2e-6  # fp64 sums of the taps in LDS, fp32 FFT of <= 4096 cells: observed ~2e-7
T2 = {1: 2e-1, 2: 2e-2, 3: 3e-3, 4: 5e-4, 5: 1e-4, 6: 5e-5, 7: 3e-5, 8: 2e-5}


@pytest.fixture(scope="module")
def tn():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import torch_nfft_amd
    return torch_nfft_amd


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _problem(rng, n, sizes, cols, complex_x):
    """points on the torus incl. both ends of [-1/2, 1/2) and a cell boundary; `sizes` = points per point set"""
    pos = (rng.random((n, 1)) - 0.5).astype(np.float32)
    pos[:4, 0] = (-0.5, np.nextafter(np.float32(0.5), np.float32(0)), 0.0, 0.25)
    batch = None if sizes is None else np.repeat(np.arange(len(sizes)), sizes).astype(np.int64)
    x = rng.standard_normal((n,) + cols).astype(np.float32)
    if complex_x:
        x = (x + 1j * rng.standard_normal(x.shape)).astype(np.complex64)
    return pos, batch, x


def test_plan_needed_says_which_problems_are_fused(tn):
    from torch_nfft_amd import _lib
    lib = _lib.load()
    need = lambda d, n, B, N, m: lib.nfft_hip_plan_needed(ctypes.byref(_lib.Problem(d, n, 1, B, N, m)))
    assert need(1, 1000, 1, 64, 2) == 0          # BASELINE config 1
    assert need(1, 1000, 1, 2048, 8) == 0        # 4096 cells: the largest fused grid
    assert need(1, 1000, 1, 4096, 4) == 1        # 8192 cells
    assert need(1, 1000, 1, 100, 4) == 1         # 200 cells: not a power of two
    assert need(1, 10 ** 6, 1, 64, 2) == 1       # too many points for one workgroup
    assert need(1, 10 ** 5, 8, 64, 2) == 0       # ... but fine over 8 point sets
    assert need(2, 1000, 1, 64, 2) == 1 and need(3, 1000, 1, 16, 2) == 1


@pytest.mark.parametrize("N,m", [(2, 1), (8, 2), (64, 2), (64, 8), (512, 4), (2048, 3)])
@pytest.mark.parametrize("complex_x", [False, True])
def test_fused_1d_vs_oracle(tn, N, m, complex_x):
    """three point sets (the middle one EMPTY), two columns, both directions, complex and real_output results"""
    rng = np.random.default_rng(9000 + N + m)
    sizes = [311, 0, 402]
    n, B, cols = sum(sizes), 3, (2,)
    pos, batch, x = _problem(rng, n, sizes, cols, complex_x)
    for real_output in (False, True):
        ya = tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=N, cutoff=m, real_output=real_output)
        assert ya.shape == (B, N) + cols and ya.dtype == (torch.float32 if real_output else torch.complex64)
        assert rel_l2(host(ya), nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m, real_output=real_output)) < T1N
        assert float(ya[1].abs().max()) == 0.0  # the empty point set
    if N >= 8:  # (the window is wider than a 4-cell grid: the NFFT itself is no approximation of the NDFT there)
        assert rel_l2(host(tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=N, cutoff=m)),
                      ndft.ndft_adjoint(x, pos, batch, N=N)) < T2[m]
    xh = rng.standard_normal((B, N) + cols).astype(np.float32)
    if complex_x:
        xh = (xh + 1j * rng.standard_normal(xh.shape)).astype(np.complex64)
    for real_output in (False, True):
        yf = tn.nfft_forward(dev(xh), dev(pos), dev(batch), cutoff=m, real_output=real_output)
        assert yf.shape == (n,) + cols and yf.dtype == (torch.float32 if real_output else torch.complex64)
        assert rel_l2(host(yf), nfft_ref.nfft_forward(xh, pos, batch, m=m, real_output=real_output)) < T1N
    if N >= 8:
        assert rel_l2(host(tn.nfft_forward(dev(xh), dev(pos), dev(batch), cutoff=m)), ndft.ndft_forward(xh, pos, batch)) < T2[m]


@pytest.mark.parametrize("N,m,B", [(64, 2, 1), (256, 4, 4), (2048, 6, 2)])
def test_fused_and_general_path_agree(tn, N, m, B):
    """nfft_hip_adjoint / nfft_hip_forward (fused) against nfft_hip_plan_points + the *_planned entry points (the general
    path, which ignores nfft_hip_plan_needed) on the same device buffers."""
    from torch_nfft_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(31 + N)
    n, C = 5000, 3
    sizes = None if B == 1 else list(rng.multinomial(n, np.ones(B) / B))
    pos, batch, x = _problem(rng, n, sizes, (C,), True)
    prob = _lib.Problem(1, n, C, B, N, m)
    P = ctypes.byref(prob)
    assert lib.nfft_hip_plan_needed(P) == 0
    pt, bt, xt = dev(pos), dev(batch), dev(x)
    plan = torch.empty(lib.nfft_hip_plan_bytes(P), dtype=torch.uint8, device="cuda")
    _lib.check(lib.nfft_hip_plan_points(P, _p(pt), _p(bt), _p(plan), plan.numel(), _stream()))
    wsa = torch.empty(lib.nfft_hip_adjoint_workspace_bytes(P, 1, 0), dtype=torch.uint8, device="cuda")
    y_fused = torch.full((B, N, C), float("nan"), dtype=torch.complex64, device="cuda")
    y_general = torch.full_like(y_fused, float("nan"))
    _lib.check(lib.nfft_hip_adjoint(P, _p(pt), _p(xt), 1, _p(bt), 0, _p(y_fused), None, 0, _stream()))  # no workspace
    _lib.check(lib.nfft_hip_adjoint_planned(P, _p(plan), _p(xt), 1, 0, _p(y_general), _p(wsa), wsa.numel(), _stream()))
    assert rel_l2(host(y_fused), host(y_general)) < T1N
    wsf = torch.empty(lib.nfft_hip_forward_workspace_bytes(P, 1, 0), dtype=torch.uint8, device="cuda")
    f_fused = torch.full((n, C), float("nan"), dtype=torch.complex64, device="cuda")
    f_general = torch.full_like(f_fused, float("nan"))
    _lib.check(lib.nfft_hip_forward(P, _p(pt), _p(y_general), 1, _p(bt), 0, _p(f_fused), None, 0, _stream()))
    _lib.check(lib.nfft_hip_forward_planned(P, _p(plan), _p(y_general), 1, 0, _p(f_general), _p(wsf), wsf.numel(), _stream()))
    assert rel_l2(host(f_fused), host(f_general)) < T1N


def test_fused_1d_empty_input_and_faults(tn):
    from torch_nfft_amd import _lib, ops
    lib = _lib.load()
    # no points: the adjoint is all zeros, the forward transform an empty tensor
    y = tn.nfft_adjoint(torch.zeros((0, 2), device="cuda"), torch.zeros((0, 1), device="cuda"), None, bandwidth=64, cutoff=2)
    assert y.shape == (1, 64, 2) and float(y.abs().max()) == 0.0
    yf = tn.nfft_forward(y, torch.zeros((0, 1), device="cuda"), None, cutoff=2)
    assert yf.shape == (0, 2)
    # a batch vector that names a point set outside [0, batch_size) is reported (C boundary: the torch operator derives
    # batch_size from the vector itself)
    rng = np.random.default_rng(5)
    n = 300
    pos, batch, x = _problem(rng, n, [100, 100, 100], (), False)
    prob = _lib.Problem(1, n, 1, 2, 64, 2)  # batch_size 2, but the vector holds 0..2
    yt = torch.zeros((2, 64), dtype=torch.complex64, device="cuda")
    ops.check_status()
    pt, xt, bt = dev(pos), dev(x), dev(batch)  # (kept alive: the call is asynchronous)
    rc = lib.nfft_hip_adjoint(ctypes.byref(prob), _p(pt), _p(xt), 0, _p(bt), 0, _p(yt), None, 0, _stream())
    assert rc == 0
    with pytest.raises(RuntimeError, match="Input mismatch: batch holds an index outside"):
        ops.check_status()
    ops.check_status()
    # the two sets that do exist are transformed from their own points
    ref = nfft_ref.nfft_adjoint(x[:200], pos[:200], batch[:200], N=64, m=2)
    assert rel_l2(host(yt), ref) < T1N


def test_fused_1d_pair_is_adjoint(tn):
    """the two fused kernels are each other's transposes (same window, same roll-off): <A x, w> == <x, A^H w>, and the
    autograd backward of one is the other (reference: nfft.py:22-28, 48-54)"""
    rng = np.random.default_rng(77)
    n, N, m = 2000, 128, 4
    pos, _, x = _problem(rng, n, None, (), True)
    w = (rng.standard_normal((1, N)) + 1j * rng.standard_normal((1, N))).astype(np.complex64)
    ya = tn.nfft_adjoint(dev(x), dev(pos), None, bandwidth=N, cutoff=m)
    fw = tn.nfft_forward(dev(w), dev(pos), None, cutoff=m)
    lhs = complex((ya * dev(w).conj()).sum())
    rhs = complex((dev(x) * fw.conj()).sum())
    assert abs(lhs - rhs) < 1e-5 * abs(lhs)
    xr = dev(x.real.copy()).requires_grad_(True)
    wr = dev(rng.standard_normal((1, N)).astype(np.float32))
    (tn.nfft_adjoint(xr, dev(pos), None, bandwidth=N, cutoff=m, real_output=True) * wr).sum().backward()
    expect = tn.nfft_forward(wr, dev(pos), None, cutoff=m, real_output=True)
    assert rel_l2(host(xr.grad), host(expect)) < 1e-5
