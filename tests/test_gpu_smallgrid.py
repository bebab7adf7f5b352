"""GPU parity of the fused small-grid path (csrc/smallgrid.hip): transforms whose oversampled grid fits one workgroup's
LDS (<= 4096 cells: 1-D N <= 2048, 2-D N <= 32, 3-D N <= 8) run as ONE kernel per direction on the caller's points,
without a point plan.  Checked against the oracle
(oracle/nfft_ref.py = the reference's algorithm in float64; oracle/ndft.py = the exact sums) and against the general path
(point plan -> spreading -> rocFFT -> roll-off) through the planned C entry points on the same inputs."""
import ctypes
import os

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


def _problem(rng, n, sizes, cols, complex_x, d=1):
    """points on the torus incl. both ends of [-1/2, 1/2) and a cell boundary; `sizes` = points per point set"""
    pos = (rng.random((n, d)) - 0.5).astype(np.float32)
    pos[:4, :] = np.array((-0.5, np.nextafter(np.float32(0.5), np.float32(0)), 0.0, 0.25), dtype=np.float32)[:, None]
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
    assert need(1, 10 ** 5, 1, 64, 2) == 1       # 6e5 window taps: too many for one workgroup
    assert need(1, 7 * 10 ** 4, 8, 64, 2) == 0   # ... but fine over 8 point sets
    assert need(2, 500, 1, 32, 4) == 0 and need(3, 200, 1, 8, 2) == 0    # 64^2 and 16^3 cells
    assert need(2, 1000, 1, 64, 2) == 1 and need(3, 1000, 1, 16, 2) == 1 # 128^2, 32^3 cells
    assert need(2, 1000, 1, 32, 4) == 1 and need(2, 500, 1, 32, 4) == 0   # 10^5 / 5e4 window taps in one workgroup


@pytest.mark.parametrize("d,N,m", [(1, 2, 1), (1, 8, 2), (1, 64, 2), (1, 64, 8), (1, 512, 4), (1, 2048, 3),
                                   (2, 2, 1), (2, 4, 3), (2, 16, 3), (2, 16, 4), (2, 32, 8), (2, 32, 1),
                                   (3, 2, 1), (3, 4, 2), (3, 8, 2), (3, 8, 7)])
@pytest.mark.parametrize("complex_x", [False, True])
def test_fused_vs_oracle(tn, d, N, m, complex_x):
    """three point sets (the middle one EMPTY), two columns, both directions, complex and real_output results"""
    from torch_nfft_amd import _lib
    rng = np.random.default_rng(9000 + 100 * d + N + m)
    # (the fused path takes at most 8e4 window taps per point set on average: fewer points for the wide windows)
    taps = (2 * m + 2) ** d
    big = max(2, min(402, int(0.55 * 180000 // taps)))
    sizes = [max(1, int(0.77 * big)), 0, big]
    n, B, cols = sum(sizes), 3, (2,)
    assert _lib.load().nfft_hip_plan_needed(ctypes.byref(_lib.Problem(d, n, 2, B, N, m))) == 0
    pos, batch, x = _problem(rng, n, sizes, cols, complex_x, d)
    for real_output in (False, True):
        ya = tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=N, cutoff=m, real_output=real_output)
        assert ya.shape == (B,) + (N,) * d + cols and ya.dtype == (torch.float32 if real_output else torch.complex64)
        assert rel_l2(host(ya), nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m, real_output=real_output)) < T1N
        assert float(ya[1].abs().max()) == 0.0  # the empty point set
    if N >= 8:  # (the window is wider than a 4-cell grid: the NFFT itself is no approximation of the NDFT there)
        assert rel_l2(host(tn.nfft_adjoint(dev(x), dev(pos), dev(batch), bandwidth=N, cutoff=m)),
                      ndft.ndft_adjoint(x, pos, batch, N=N)) < T2[m]
    xh = rng.standard_normal((B,) + (N,) * d + cols).astype(np.float32)
    if complex_x:
        xh = (xh + 1j * rng.standard_normal(xh.shape)).astype(np.complex64)
    for real_output in (False, True):
        yf = tn.nfft_forward(dev(xh), dev(pos), dev(batch), cutoff=m, real_output=real_output)
        assert yf.shape == (n,) + cols and yf.dtype == (torch.float32 if real_output else torch.complex64)
        assert rel_l2(host(yf), nfft_ref.nfft_forward(xh, pos, batch, m=m, real_output=real_output)) < T1N
    if N >= 8:
        assert rel_l2(host(tn.nfft_forward(dev(xh), dev(pos), dev(batch), cutoff=m)), ndft.ndft_forward(xh, pos, batch)) < T2[m]


@pytest.mark.parametrize("d,N,m,B", [(1, 64, 2, 1), (1, 256, 4, 4), (1, 2048, 6, 2), (2, 16, 4, 3), (2, 32, 3, 1), (3, 8, 2, 2),
                                     (3, 8, 4, 1)])
def test_fused_and_general_path_agree(tn, d, N, m, B):
    """nfft_hip_adjoint / nfft_hip_forward (fused) against nfft_hip_plan_points + the *_planned entry points (the general
    path, which ignores nfft_hip_plan_needed) on the same device buffers."""
    from torch_nfft_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(31 + N + d)
    n, C = int(0.9 * 80000 * min(B, 8) // (2 * m + 2) ** d), 3  # (8e4 window taps per set is the fused path's limit)
    sizes = None if B == 1 else list(rng.multinomial(n, np.ones(B) / B))
    pos, batch, x = _problem(rng, n, sizes, (C,), True, d)
    prob = _lib.Problem(d, n, C, B, N, m)
    P = ctypes.byref(prob)
    assert lib.nfft_hip_plan_needed(P) == 0
    pt, bt, xt = dev(pos), dev(batch), dev(x)
    plan = torch.empty(lib.nfft_hip_plan_bytes(P), dtype=torch.uint8, device="cuda")
    _lib.check(lib.nfft_hip_plan_points(P, _p(pt), _p(bt), _p(plan), plan.numel(), _stream()))
    wsa = torch.empty(lib.nfft_hip_adjoint_workspace_bytes(P, 1, 0), dtype=torch.uint8, device="cuda")
    y_fused = torch.full((B,) + (N,) * d + (C,), float("nan"), dtype=torch.complex64, device="cuda")
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


def test_fused_empty_input_and_faults(tn):
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
    # a bad entry in the MIDDLE of the vector (the ends are fine): the fused kernels look at every row's index, as the
    # general path's sort does -- out of range, and out of order
    for bad_value, message in ((7, "Input mismatch"), (0, "Input mismatch: the batch vector is not sorted")):
        b2 = batch.copy()
        b2[150] = bad_value
        bt2 = dev(b2)
        for fwd in (False, True):
            ops.plan_cache_clear()
            ops.check_status()
            with pytest.raises(RuntimeError, match=message):
                if fwd:
                    tn.nfft_forward(torch.zeros((3, 64), dtype=torch.complex64, device="cuda"), pt, bt2, cutoff=2)
                else:
                    tn.nfft_adjoint(xt, pt, bt2, bandwidth=64, cutoff=2)
                ops.check_status()
            try:
                ops.check_status()
            except RuntimeError:
                pass


def test_fused_pair_is_adjoint(tn):
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


def test_general_path_on_small_grids():
    """NFFT_HIP_SMALL_GRID=0 turns the fused path off: the same small problems through point plan -> LDS-tile spreading ->
    rocFFT / the pruned column passes (16^3 cells) -> roll-off -> gather, against the oracle (these sizes reach the general
    path only this way since the fused kernels exist)."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
import torch_nfft_amd as tn
from oracle import nfft_ref
worst = 0.0
for d, N, m in ((1, 64, 2), (1, 32, 3), (1, 512, 4), (2, 16, 3), (2, 32, 4), (2, 8, 1), (2, 16, 6), (3, 8, 2), (3, 8, 4), (3, 4, 1)):
    rng = np.random.default_rng(1000 * d + N + m)
    n, B = 700, 3
    pos = (rng.random((n, d)) - 0.5).astype(np.float32)
    batch = np.sort(rng.integers(0, B, n)).astype(np.int64); batch[0], batch[-1] = 0, B - 1
    x = (rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))).astype(np.complex64)
    xt, pt, bt = torch.from_numpy(x).cuda(), torch.from_numpy(pos).cuda(), torch.from_numpy(batch).cuda()
    y = tn.nfft_adjoint(xt, pt, bt, bandwidth=N, cutoff=m)
    ref = nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)
    worst = max(worst, np.linalg.norm(y.cpu().numpy() - ref) / np.linalg.norm(ref))
    f = tn.nfft_forward(y, pt, bt, cutoff=m, real_output=True)
    reff = nfft_ref.nfft_forward(y.cpu().numpy(), pos, batch, m=m, real_output=True)
    worst = max(worst, np.linalg.norm(f.cpu().numpy() - reff) / np.linalg.norm(reff))
print("RESULT", worst)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NFFT_HIP_SMALL_GRID="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) < 2e-5
