"""GPU tests of the drop-in boundary itself:

* the un-planned C entry points ``nfft_hip_adjoint`` / ``nfft_hip_forward`` / ``nfft_hip_fastsum`` exactly as a
  non-torch host would call them (ctypes, raw device pointers, caller-provided workspace), on the golden vectors
  frozen from the reference's ``torch_nfft/ndft.py`` and against the oracle;
* the native operator registry ``core.so`` (``torch.ops.torch_nfft.*`` served by C++, not by Python);
* the ``torch_nfft`` drop-in package name;
* the point-plan cache across streams and the rejection of a negative batch index;
* the batch-sharded wrapper (``torch_nfft_amd.distributed``) on one GPU: world_size 1, and every shard of a simulated
  world computed one after the other, against the unsharded HIP call.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import ndft, nfft_ref

pytestmark = pytest.mark.gpu

T1 = 2e-5
T2 = {3: 3e-3, 4: 5e-4}


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


def c_adjoint(x, pos, batch, N, m, real_output=False):
    """nfft_hip_adjoint through ctypes: numpy in, numpy out."""
    from torch_nfft_amd import _lib
    lib = _lib.load()
    n, d = pos.shape
    B = int(batch[-1]) + 1 if batch is not None else 1
    C = int(np.prod(x.shape[1:], dtype=np.int64))
    cplx = 1 if np.iscomplexobj(x) else 0
    prob = _lib.Problem(d, n, C, B, N, m)
    nbytes = lib.nfft_hip_adjoint_workspace_bytes(ctypes.byref(prob), cplx, 1 if real_output else 0)
    assert nbytes > 0, _lib.last_error()
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    xt, pt, bt = dev(x), dev(pos), dev(batch)
    y = torch.full((B,) + (N,) * d + x.shape[1:], float("nan"),
                   dtype=torch.float32 if real_output else torch.complex64, device="cuda")
    _lib.check(lib.nfft_hip_adjoint(ctypes.byref(prob), _p(pt), _p(xt), cplx, _p(bt), 1 if real_output else 0, _p(y),
                                    _p(ws), nbytes, _stream()))
    return host(y)


def c_forward(xhat, pos, batch, m, real_output=False):
    from torch_nfft_amd import _lib
    lib = _lib.load()
    n, d = pos.shape
    B, N = xhat.shape[0], xhat.shape[1]
    cols = xhat.shape[1 + d:]
    C = int(np.prod(cols, dtype=np.int64))
    cplx = 1 if np.iscomplexobj(xhat) else 0
    prob = _lib.Problem(d, n, C, B, N, m)
    nbytes = lib.nfft_hip_forward_workspace_bytes(ctypes.byref(prob), cplx, 1 if real_output else 0)
    assert nbytes > 0, _lib.last_error()
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    xt, pt, bt = dev(xhat), dev(pos), dev(batch)
    y = torch.full((n,) + cols, float("nan"), dtype=torch.float32 if real_output else torch.complex64, device="cuda")
    _lib.check(lib.nfft_hip_forward(ctypes.byref(prob), _p(pt), _p(xt), cplx, _p(bt), 1 if real_output else 0, _p(y),
                                    _p(ws), nbytes, _stream()))
    return host(y)


# ----------------------------------------------------------------------------- un-planned C entry points

def test_c_abi_adjoint_forward_golden_g1_2d(tn):
    g = load_golden("g1_adjoint_2d_batched")
    y = c_adjoint(g["x"], g["pos"], g["batch"], 16, 4)
    assert y.shape == (3, 16, 16, 10)
    assert rel_l2(y, g["y_adjoint"]) < T2[4]
    assert rel_l2(y, nfft_ref.nfft_adjoint(g["x"], g["pos"], g["batch"], N=16, m=4)) < T1
    f = c_forward(y, g["pos"], g["batch"], 4)
    assert rel_l2(f, nfft_ref.nfft_forward(y, g["pos"], g["batch"], m=4)) < T1
    assert rel_l2(f, ndft.ndft_forward(y, g["pos"], g["batch"])) < T2[4]


def test_c_abi_adjoint_forward_golden_g4_3d(tn):
    g = load_golden("g4_3d_ragged")
    for key in ("real", "complex"):
        y = c_adjoint(g["x_" + key], g["pos"], g["batch"], 16, 4)
        assert rel_l2(y, g["y_adjoint_" + key]) < T2[4]
        assert rel_l2(y, nfft_ref.nfft_adjoint(g["x_" + key], g["pos"], g["batch"], N=16, m=4)) < T1
    yr = c_adjoint(g["x_complex"], g["pos"], g["batch"], 16, 4, real_output=True)
    assert yr.dtype == np.float32
    assert rel_l2(yr, nfft_ref.nfft_adjoint(g["x_complex"], g["pos"], g["batch"], N=16, m=4, real_output=True)) < T1
    f = c_forward(g["xhat"], g["pos"], g["batch"], 4)
    assert rel_l2(f, g["y_forward"]) < T2[4]
    assert rel_l2(f, nfft_ref.nfft_forward(g["xhat"], g["pos"], g["batch"], m=4)) < T1
    fr = c_forward(g["xhat"], g["pos"], g["batch"], 4, real_output=True)
    assert rel_l2(fr, nfft_ref.nfft_forward(g["xhat"], g["pos"], g["batch"], m=4, real_output=True)) < T1


def test_c_abi_matrix_core_grid_unplanned(tn):
    """The un-planned entry points on a grid that takes the matrix-core kernels (the plan is built into the
    workspace by the call itself)."""
    rng = np.random.default_rng(17)
    n, N, m = 5000, 32, 4
    pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
    batch = np.sort(rng.integers(0, 2, n)).astype(np.int64)
    batch[0], batch[-1] = 0, 1
    x = rng.standard_normal((n, 2)).astype(np.float32)
    y = c_adjoint(x, pos, batch, N, m)
    assert rel_l2(y, nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)) < T1
    f = c_forward(y, pos, batch, m, real_output=True)
    assert rel_l2(f, nfft_ref.nfft_forward(y, pos, batch, m=m, real_output=True)) < T1


@pytest.mark.parametrize("d,N,shared,complex_x,complex_coeffs", [(2, 16, True, False, False), (2, 16, False, True, True),
                                                                   (3, 32, False, False, True), (3, 32, True, True, False),
                                                                   (1, 64, False, False, False),
                                                                   # 1-D, grid in LDS: two fused kernels (smallgrid.hip)
                                                                   (1, 64, True, True, True), (1, 512, False, True, False),
                                                                   (1, 100, False, True, True)])  # (200 cells: general path)
def test_c_abi_fastsum(tn, d, N, shared, complex_x, complex_coeffs):
    """nfft_hip_fastsum, one C call: spread -> FFT -> coefficient product -> FFT -> gather (core_cuda.cu:535-852)."""
    from torch_nfft_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(100 * d + N + shared)
    m, B, C = 4, 2, 3
    ns, nt = 400, 400 if shared else 650
    def pts(n):
        pos = ((rng.random((n, d)) - 0.5) * 0.5).astype(np.float32)
        batch = np.sort(rng.integers(0, B, n)).astype(np.int64)
        batch[0], batch[-1] = 0, B - 1
        return pos, batch
    src, sb = pts(ns)
    tgt, tb = (src, sb) if shared else pts(nt)
    x = rng.standard_normal((ns, C)).astype(np.float32)
    if complex_x:
        x = (x + 1j * rng.standard_normal((ns, C))).astype(np.complex64)
    coeffs = rng.standard_normal((N,) * d).astype(np.float32)
    if complex_coeffs:
        coeffs = (coeffs + 1j * rng.standard_normal((N,) * d)).astype(np.complex64)
    ps = _lib.Problem(d, ns, C, B, N, m)
    pt = _lib.Problem(d, nt, C, B, N, m)
    nbytes = lib.nfft_hip_fastsum_workspace_bytes(ctypes.byref(ps), ctypes.byref(pt), 1 if complex_x else 0,
                                                  1 if shared else 0, 0)
    assert nbytes > 0, _lib.last_error()
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    srct, sbt, xt, ct = dev(src), dev(sb), dev(x), dev(coeffs)
    tgtt, tbt = (srct, sbt) if shared else (dev(tgt), dev(tb))
    y = torch.full((nt, C), float("nan"), dtype=torch.complex64 if complex_x else torch.float32, device="cuda")
    _lib.check(lib.nfft_hip_fastsum(ctypes.byref(ps), _p(srct), _p(sbt), ctypes.byref(pt), _p(tgtt), _p(tbt), _p(xt),
                                    1 if complex_x else 0, _p(ct), 1 if complex_coeffs else 0, _p(y), _p(ws), nbytes,
                                    _stream()))
    ref = nfft_ref.nfft_fastsum(x, coeffs, src, None if shared else tgt, sb, None if shared else tb, m=m)
    if not complex_x:
        ref = ref.real
    assert rel_l2(host(y), ref) < T1
    exact = ndft.ndft_fastsum(x, coeffs, src, None if shared else tgt, sb, None if shared else tb)
    if not complex_x:
        exact = exact.real
    assert rel_l2(host(y), exact) < 2e-3
    # the operator gives the same numbers (it calls the planned variant of the same driver)
    y2 = tn.nfft_fastsum(xt, ct, srct, None if shared else tgtt, sbt, None if shared else tbt, cutoff=m)
    assert y2.dtype == y.dtype
    assert rel_l2(host(y2), host(y)) < 2e-6


# ----------------------------------------------------------------------------- native operator registry

def test_operators_are_served_by_core_so(tn):
    from torch_nfft_amd import _lib
    assert any(os.path.realpath(p) == os.path.realpath(_lib.CORE_PATH) for p in torch.ops.loaded_libraries)
    mapped = open("/proc/self/maps").read()
    assert "torch_nfft_amd/core.so" in mapped and "torch_nfft_amd/libnfft_hip.so" in mapped
    names = ["nfft_adjoint", "nfft_forward", "nfft_fastsum", "gaussian_analytic_coeffs", "gaussian_interpolated_coeffs",
             "interpolation_grid", "radial_interpolation_grid", "interpolated_kernel_coeffs"]
    for name in names:
        op = getattr(torch.ops.torch_nfft, name).default
        # registered from C++ (TORCH_LIBRARY in csrc/core.cpp): no Python kernel behind any dispatch key
        assert not op.py_kernels, name
    a = torch.ops.torch_nfft.gaussian_analytic_coeffs(0.2, 8, 2)
    assert a.is_cuda and a.shape == (8, 8)


def test_drop_in_package_name(tn):
    """`import torch_nfft` (the reference's package name) resolves to this implementation."""
    import torch_nfft
    assert torch_nfft.nfft_adjoint is tn.nfft_adjoint and torch_nfft.nfft_fastsum is tn.nfft_fastsum
    for name in ("nfft_forward", "ndft_adjoint", "ndft_forward", "ndft_fastsum", "exact_gaussian_matrix",
                 "exact_trigonometric_matrix", "gaussian_analytic_coeffs", "GramMatrix", "AdjacencyMatrix",
                 "GaussianKernel"):
        assert hasattr(torch_nfft, name), name


def test_negative_batch_index_is_rejected(tn):
    pos = torch.zeros((4, 2), device="cuda")
    x = torch.ones((4,), device="cuda")
    with pytest.raises(RuntimeError, match="Input mismatch"):
        tn.nfft_adjoint(x, pos, torch.tensor([-1, 0, 0, 1], device="cuda"), bandwidth=8, cutoff=2)


def test_batch_vector_ends_are_remembered_and_follow_in_place_updates(tn):
    """B = batch[-1] + 1 costs a blocking read-back per operator call; the ends of a batch vector are remembered by tensor
    identity + version counter (core.cpp: batch_ends).  An in-place edit through the tensor must be seen, the cache
    switches of the plan cache clear / disable it, and a vector that turns invalid is rejected."""
    from torch_nfft_amd import ops
    rng = np.random.default_rng(17)
    n, N, m = 600, 16, 3
    pos = dev((rng.random((n, 2)) - 0.5).astype(np.float32))
    x = dev(rng.standard_normal(n).astype(np.float32))
    batch = dev((np.arange(n) >= n // 2).astype(np.int64))          # two point sets
    ops.plan_cache_clear()
    assert tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m).shape[0] == 2
    assert tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m).shape[0] == 2   # (from the remembered ends)
    batch[n - 10:] = 2                                               # in place: version counter changes
    y3 = tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
    assert y3.shape[0] == 3
    ref = nfft_ref.nfft_adjoint(host(x), host(pos), host(batch), N=N, m=m)
    assert rel_l2(host(y3), ref) < T1
    ops.plan_cache_enabled(False)
    assert tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m).shape[0] == 3
    ops.plan_cache_enabled(True)
    batch[0] = -1
    with pytest.raises(RuntimeError, match="Input mismatch"):
        tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
    with pytest.raises(RuntimeError, match="Input mismatch"):       # (remembered ends are checked like fresh ones)
        tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
    ops.plan_cache_clear()


def test_plan_cache_across_streams(tn):
    """A plan built on one stream and consumed on another: the consumer waits for the build and the results agree."""
    from torch_nfft_amd import ops
    rng = np.random.default_rng(5)
    n, N, m = 20000, 32, 4
    pos = dev((rng.random((n, 3)) - 0.5).astype(np.float32))
    x = dev(rng.standard_normal(n).astype(np.float32))
    ops.plan_cache_clear()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        y1 = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
    h0 = ops.plan_cache_stats()["hits"]
    with torch.cuda.stream(s2):  # no host synchronisation in between
        y2 = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
        f2 = tn.nfft_forward(y2, pos, None, cutoff=m, real_output=True)
    assert ops.plan_cache_stats()["hits"] == h0 + 2
    torch.cuda.synchronize()
    assert rel_l2(host(y2), host(y1)) < 2e-6
    assert rel_l2(host(f2), nfft_ref.nfft_forward(host(y1), host(pos), None, m=m, real_output=True)) < T1


# ----------------------------------------------------------------------------- batch-sharded wrapper on one GPU

def test_sharded_wrapper_world1_and_simulated_shards(tn):
    """torch_nfft_amd.distributed with the HIP operators: world_size 1 (no process group) equals the plain call, and
    the shards a world of R ranks would compute -- run here one after the other on the one GPU through the wrapper's
    own partitioning -- concatenate to the unsharded result."""
    from torch_nfft_amd import distributed as D
    rng = np.random.default_rng(9)
    B, N, m, C = 5, 32, 4, 3
    counts = np.array([700, 0, 1300, 400, 900])
    batch = np.repeat(np.arange(B), counts).astype(np.int64)
    n = batch.shape[0]
    pos = (rng.random((n, 3)) - 0.5).astype(np.float32)
    x = rng.standard_normal((n, C)).astype(np.float32)
    xt, pt, bt = dev(x), dev(pos), dev(batch)
    full = tn.nfft_adjoint(xt, pt, bt, bandwidth=N, cutoff=m)
    assert rel_l2(host(full), nfft_ref.nfft_adjoint(x, pos, batch, N=N, m=m)) < T1
    y1 = D.nfft_adjoint(xt, pt, bt, bandwidth=N, cutoff=m)  # world_size 1
    assert torch.equal(y1, full) or rel_l2(host(y1), host(full)) < 2e-6
    fwd_full = tn.nfft_forward(full, pt, bt, cutoff=m, real_output=True)
    f1 = D.nfft_forward(full, pt, bt, cutoff=m, real_output=True)
    assert rel_l2(host(f1), host(fwd_full)) < 2e-6
    for world in (2, 3, 8):
        slabs, rows = [], []
        for rank in range(world):
            slabs.append(D.shard_adjoint(xt, pt, bt, B, rank, world, bandwidth=N, cutoff=m))
            rows.append(D.shard_forward(full, pt, bt, B, rank, world, cutoff=m, real_output=True))
            # sharded-input forward: the rank's slab only (never the replicated spectrum), with the layout's last set
            _, bounds, lasts = D._layout(bt, world, n)
            if bounds[rank + 1] > bounds[rank]:
                own = D.shard_forward(slabs[-1], pt, bt, B, rank, world, cutoff=m, real_output=True, bounds=bounds,
                                      x_is_local=True, last_set=lasts[rank])
                assert rel_l2(host(own), host(rows[-1])) < 2e-6
        assert [s.shape[0] for s in slabs] == [D.batch_range(B, r, world)[1] - D.batch_range(B, r, world)[0]
                                               for r in range(world)]
        assert rel_l2(host(torch.cat(slabs, 0)), host(full)) < 2e-6
        assert rel_l2(host(torch.cat(rows, 0)), host(fwd_full)) < 2e-6


# ----------------------------------------------------------------------------- device-side fault reports

def test_streamed_gather_stall_is_reported_to_the_host():
    """A bounded wait of the streamed interpolation kernel that runs out must not hand back unfinished rows as a
    result (include/nfft_hip.h nfft_hip_check_status).  The variant library tests/variants/libnfft_hip_spin0.so is the
    product library with interp_stream.hip compiled for a spin limit of 0, so the first wait of every work item gives
    up: the forward transform raises the device's fault flag, `ops.check_status()` raises, so does the next operator
    call (which does not run), and the call after that works again (the flag is cleared when it is reported).
    Reference behaviour being replaced: print + exit() on a device error, csrc/cuda/cuda_utils.cu:5-16."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variant = os.path.join(root, "tests", "variants", "libnfft_hip_spin0.so")
    assert os.path.exists(variant), "variant library missing: run python torch_nfft_amd/build.py"
    code = r'''
import sys, torch
sys.path.insert(0, %r)
import torch_nfft_amd as tn
from torch_nfft_amd import ops
g = torch.Generator(device="cuda").manual_seed(3)
n, N, m = 60000, 64, 4
pos = torch.rand((n, 3), generator=g, device="cuda") - 0.5
xh = torch.randn((1, N, N, N), generator=g, device="cuda", dtype=torch.float32).to(torch.complex64)
ops.check_status()                      # clean so far
y = tn.nfft_forward(xh, pos, None, cutoff=m, real_output=True)   # streamed gather: every wait gives up at once
try:
    ops.check_status()
    print("NO_FAULT_REPORTED")
except RuntimeError as e:
    print("FIRST", "streamed interpolation" in str(e))
ops.check_status()                      # reported once, then clear
y = tn.nfft_forward(xh, pos, None, cutoff=m, real_output=True)
torch.cuda.synchronize()
try:
    tn.nfft_adjoint(torch.ones(n, device="cuda"), pos, None, bandwidth=N, cutoff=m)   # the NEXT entry point refuses
    print("NEXT_CALL_RAN")
except RuntimeError as e:
    print("SECOND", "device fault" in str(e))
z = tn.nfft_adjoint(torch.ones(n, device="cuda"), pos, None, bandwidth=N, cutoff=m)
ops.check_status()
print("THIRD", bool(torch.isfinite(z.real).all()))
''' % root
    env = dict(os.environ, NFFT_HIP_LIB=variant, NFFT_HIP_STREAM_MIN="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    assert "FIRST True" in out.stdout and "SECOND True" in out.stdout and "THIRD True" in out.stdout, out.stdout


def test_status_is_clean_after_the_product_streamed_gather(tn):
    """The product library's streamed gather (forced onto a small problem by its own subprocess elsewhere; here the C3-like
    default path on 3e6 points) leaves no fault behind: check_status() after a forward transform returns quietly."""
    from torch_nfft_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    n, N, m = 5_000_000, 128, 4
    pos = torch.rand((n, 3), generator=g, device="cuda") - 0.5
    xh = torch.randn((1, N, N, N), generator=g, device="cuda").to(torch.complex64)
    y = tn.nfft_forward(xh, pos, None, cutoff=m, real_output=True)
    ops.check_status()
    assert bool(torch.isfinite(y).all())


def test_batch_index_out_of_range_in_the_middle_is_reported(tn):
    """Only batch[0] and batch[-1] are read back by the host (as in the reference, core_cuda.cu:60); an index outside
    [0, B) in between is binned clamped (every access stays in range) and reported by the sort kernel: the next
    operator call raises "Input mismatch".  Both sort paths: the two-level sort and the fallback for many bins."""
    from torch_nfft_amd import ops
    rng = np.random.default_rng(8)
    for d, N, n in ((3, 32, 4000), (1, 1 << 19, 3000)):  # (2^19: 12 288 first-level bins > 8 192: fallback sort)
        pos = dev((rng.random((n, d)) - 0.5).astype(np.float32))
        batch = np.sort(rng.integers(0, 3, n)).astype(np.int64)
        batch[0], batch[-1] = 0, 2
        for wrong in (7, -1):
            bad = batch.copy()
            bad[n // 2] = wrong
            ops.plan_cache_clear()
            ops.check_status()
            # the sort kernel raises the flag; whichever entry point looks first reports it -- a later stage of this very
            # call if the kernel has finished by then, else the explicit check (which drains the stream first)
            with pytest.raises(RuntimeError, match="Input mismatch: batch holds an index outside"):
                tn.nfft_adjoint(dev(rng.standard_normal(n).astype(np.float32)), pos, dev(bad), bandwidth=N, cutoff=3)
                ops.check_status()
            try:
                ops.check_status()  # (a later sort pass of the same call -- the owned plan -- may have raised it again)
            except RuntimeError:
                pass
            ops.check_status()  # clear now
        ops.plan_cache_clear()
        y = tn.nfft_adjoint(dev(np.ones(n, dtype=np.float32)), pos, dev(batch), bandwidth=N, cutoff=3)
        ops.check_status()
        assert y.shape[0] == 3


def test_operators_under_inference_mode(tn):
    """Tensors created under torch.inference_mode() carry no version counter: the point-plan cache must not ask for one
    (the reference's operators work in inference mode).  Adjoint, forward and fastsum on inference tensors against the
    same calls on ordinary tensors."""
    from torch_nfft_amd import ops
    rng = np.random.default_rng(12)
    n, N, m = 3000, 32, 3
    pos_np = ((rng.random((n, 3)) - 0.5) * 0.5).astype(np.float32)
    x_np = rng.standard_normal((n, 2)).astype(np.float32)
    pos, x = dev(pos_np), dev(x_np)
    co = tn.gaussian_analytic_coeffs(0.2, dim=3, N=N)
    ya = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
    yf = tn.nfft_forward(ya, pos, None, cutoff=m)
    ys = tn.nfft_fastsum(x, co, pos, cutoff=m)
    before = ops.plan_cache_stats()
    with torch.inference_mode():
        pos_i, x_i = dev(pos_np), dev(x_np)
        assert pos_i.is_inference()
        ya_i = tn.nfft_adjoint(x_i, pos_i, None, bandwidth=N, cutoff=m)
        yf_i = tn.nfft_forward(ya_i, pos_i, None, cutoff=m)
        ys_i = tn.nfft_fastsum(x_i, co.clone(), pos_i, cutoff=m)
        bt = torch.zeros(n, dtype=torch.int64, device="cuda")
        yb_i = tn.nfft_adjoint(x_i, pos_i, bt, bandwidth=N, cutoff=m)
    assert rel_l2(host(ya_i), host(ya)) < 1e-6 and rel_l2(host(yf_i), host(yf)) < 1e-6
    assert rel_l2(host(ys_i), host(ys)) < 1e-6 and rel_l2(host(yb_i), host(ya)) < 1e-6
    after = ops.plan_cache_stats()
    assert after["hits"] == before["hits"]  # inference points never hit (or enter) the cache


# ----------------------------------------------------------------------------- plan reuse

def test_operator_cache_miss_and_hit_routes_agree(tn):
    """torch.ops.torch_nfft.nfft_adjoint on a plan-cache miss and on a hit (the same plan, other coefficient arrays read
    through the index in the plan records): same numbers, real and complex coefficients, against the oracle."""
    from torch_nfft_amd import ops
    rng = np.random.default_rng(92)
    n, N, m = 200000, 64, 4
    pos = dev((rng.random((n, 3)) - 0.5).astype(np.float32))
    for cplx in (False, True):
        x = rng.standard_normal(n).astype(np.float32)
        if cplx:
            x = (x + 1j * rng.standard_normal(n)).astype(np.complex64)
        xt = dev(x)
        ops.plan_cache_clear()
        s0 = ops.plan_cache_stats()
        y_miss = tn.nfft_adjoint(xt, pos, None, bandwidth=N, cutoff=m)
        y_hit = tn.nfft_adjoint(xt, pos, None, bandwidth=N, cutoff=m)
        s1 = ops.plan_cache_stats()
        assert s1["misses"] == s0["misses"] + 1 and s1["hits"] == s0["hits"] + 1
        assert rel_l2(host(y_hit), host(y_miss)) < 1e-6
        assert rel_l2(host(y_miss), nfft_ref.nfft_adjoint(x, host(pos), None, N=N, m=m)) < 2e-6


def test_unsorted_batch_vector_is_reported(tn):
    """The operand scales of the matrix-core spreading kernel are the largest |x| of every point set, taken over the set's
    ROW RANGE (the batch vector is sorted, docs/source/theory/dataformat.rst:35-37).  A point whose set index is out of
    order can carry a coefficient far above the maximum that was found for its set: the kernel clamps it (no operand
    leaves the f16 range) and reports it -- the call fails with "Input mismatch" instead of returning a wrong spectrum."""
    from torch_nfft_amd import ops
    rng = np.random.default_rng(17)
    n, N, m = 6000, 64, 3  # (N = 32 with a narrow window runs the LDS kernels since round 4: no operand scales, nothing to report)
    pos = dev((rng.random((n, 3)) - 0.5).astype(np.float32))
    batch = np.repeat(np.arange(2), n // 2).astype(np.int64)
    x = np.ones(n, dtype=np.float32)
    batch[100], x[100] = 1, 1000.0  # belongs to set 1 but sits in the rows of set 0
    ops.plan_cache_clear()
    ops.check_status()
    with pytest.raises(RuntimeError, match="Input mismatch: the batch vector is not sorted"):
        tn.nfft_adjoint(dev(x), pos, dev(batch), bandwidth=N, cutoff=m)
        ops.check_status()
    try:
        ops.check_status()
    except RuntimeError:
        pass
    ops.plan_cache_clear()
    batch[100] = 0
    y = tn.nfft_adjoint(dev(x), pos, dev(batch), bandwidth=N, cutoff=m)
    ops.check_status()
    assert rel_l2(host(y), nfft_ref.nfft_adjoint(x, host(pos), batch, N=N, m=m)) < 2e-6


def test_stale_cached_plan_is_reported(tn):
    """A write to ``pos`` behind the version counter (``pos.data``: the cache key cannot see it) after a plan has entered the
    cache: the next call on those points hits the cache, the seal of the plan does not match the points any more, and
    the fault surfaces at the next operator / ``check_status()`` -- instead of a transform of points that are no longer
    there (the reference recomputes shifts and psi in every call: csrc/cuda/core_cuda.cu:188-211)."""
    from torch_nfft_amd import ops
    rng = np.random.default_rng(23)
    n, N, m = 20000, 32, 3
    pos_a = (rng.random((n, 3)) - 0.5).astype(np.float32)
    pos_b = (rng.random((n, 3)) - 0.5).astype(np.float32)
    batch = np.sort(rng.integers(0, 3, n)).astype(np.int64)
    batch[0], batch[-1] = 0, 2
    x = rng.standard_normal(n).astype(np.float32)
    pos, xt, bt = dev(pos_a), dev(x), dev(batch)
    ops.plan_cache_clear()
    ops.check_status()
    s0 = ops.plan_cache_stats()
    y = tn.nfft_adjoint(xt, pos, bt, bandwidth=N, cutoff=m)
    assert rel_l2(host(y), nfft_ref.nfft_adjoint(x, pos_a, batch, N=N, m=m)) < 2e-6
    y = tn.nfft_adjoint(xt, pos, bt, bandwidth=N, cutoff=m)   # a genuine hit: verified, clean
    ops.check_status()
    assert ops.plan_cache_stats()["hits"] == s0["hits"] + 1
    version = pos._version
    pos.data.copy_(dev(pos_b))                                  # the version counter does not move
    assert pos._version == version
    tn.nfft_adjoint(xt, pos, bt, bandwidth=N, cutoff=m)          # hits the stale plan
    with pytest.raises(RuntimeError, match="stale point plan"):
        ops.check_status()
    ops.check_status()                                           # (the flag is cleared by the report)
    # the same for the batch vector
    ops.plan_cache_clear()
    tn.nfft_adjoint(xt, pos, bt, bandwidth=N, cutoff=m)
    b2 = batch.copy()
    b2[n // 2:] = 2
    b2[:n // 2] = np.minimum(b2[:n // 2], 1)
    bt.data.copy_(dev(np.sort(b2)))
    with pytest.raises(RuntimeError, match="stale point plan"):
        tn.nfft_adjoint(xt, pos, bt, bandwidth=N, cutoff=m)
        ops.check_status()
    try:
        ops.check_status()
    except RuntimeError:
        pass
    # after clearing the cache the new points are transformed
    ops.plan_cache_clear()
    bt2 = dev(np.sort(b2))
    y = tn.nfft_adjoint(xt, pos, bt2, bandwidth=N, cutoff=m)
    ops.check_status()
    assert rel_l2(host(y), nfft_ref.nfft_adjoint(x, pos_b, np.sort(b2), N=N, m=m)) < 2e-6
    # verification can be switched off by callers who never write behind the counter
    ops.plan_cache_verify(False)
    try:
        pos.data.copy_(dev(pos_a))
        tn.nfft_adjoint(xt, pos, bt2, bandwidth=N, cutoff=m)
        ops.check_status()                                       # (stale, and nobody looks)
    finally:
        ops.plan_cache_verify(True)
        ops.plan_cache_clear()


def test_rejected_parameters_the_reference_accepts(tn):
    """The reference puts no upper bound on the cutoff and does not require an even bandwidth (csrc/cuda/core_cuda.cu:
    118-137); this library serves m <= 8 (beyond that fp32 gains nothing) and even N.  The deviation is a loud one: the
    operators raise "Input mismatch" with the reason."""
    pos = torch.zeros((4, 2), device="cuda")
    x = torch.zeros((4,), device="cuda")
    with pytest.raises(RuntimeError, match="Input mismatch: cutoff m must be in 1..8"):
        tn.nfft_adjoint(x, pos, bandwidth=32, cutoff=9)
    with pytest.raises(RuntimeError, match="Input mismatch: cutoff m must be in 1..8"):
        tn.nfft_forward(torch.zeros((1, 32, 32), device="cuda"), pos, cutoff=9)
    with pytest.raises(RuntimeError, match="Input mismatch.*even"):
        tn.nfft_adjoint(x, pos, bandwidth=15, cutoff=3)
    with pytest.raises(RuntimeError, match="Input mismatch.*even"):
        tn.nfft_forward(torch.zeros((1, 15, 15), device="cuda"), pos, cutoff=3)
    y = tn.nfft_adjoint(x, pos, bandwidth=32, cutoff=8)  # the largest cutoff served
    assert y.shape == (1, 32, 32)


def test_shard_local_inputs_world1(tn):
    """`inputs_are_local` of the sharded wrapper with the HIP operators and no process group (world 1): the rank's own point
    sets in, its slab / rows out -- the same numbers as the plain operators (the multi-rank exchange of the counts is
    covered under gloo in tests/test_distributed_cpu.py)."""
    from torch_nfft_amd import distributed as tnd
    rng = np.random.default_rng(41)
    n, N, m = 30000, 32, 3
    pos = dev((rng.random((n, 3)) - 0.5).astype(np.float32))
    batch = dev(np.sort(rng.integers(0, 3, n)).astype(np.int64))
    x = dev(rng.standard_normal((n, 2)).astype(np.float32))
    ya = tnd.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m, inputs_are_local=True, local_batch_size=4)
    ref = tn.nfft_adjoint(x, pos, batch, bandwidth=N, cutoff=m)
    assert ya.shape[0] == 4 and float(ya[3].abs().max()) == 0.0          # a trailing empty point set
    assert rel_l2(host(ya[:3]), host(ref)) < 1e-6                        # (float atomics: not bit for bit)
    yf = tnd.nfft_forward(ya, pos, batch, cutoff=m, real_output=True, inputs_are_local=True)
    assert rel_l2(host(yf), host(tn.nfft_forward(ya[:3], pos, batch, cutoff=m, real_output=True))) < 1e-6
