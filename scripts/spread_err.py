"""Developer tool: error of the spreading stage alone (nfft_hip_spread) against the float64 oracle gridding."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from torch_nfft_amd import _lib
from oracle import nfft_ref

lib = _lib.load()
rng = np.random.default_rng(3)
d, N, m, n = 3, 32, int(os.environ.get("M_CUT", 4)), 40000
pos = (rng.random((n, d)) - 0.5).astype(np.float32)
batch = np.zeros(n, dtype=np.int64)
p = lambda t: ctypes.c_void_p(t.data_ptr())
prob = _lib.Problem(d, n, 1, 1, N, m)
plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
post = torch.from_numpy(pos).cuda()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(post), None, p(plan), plan.numel(), s))
M = 2 * N
for name, x in (("rand01", rng.random((n, 1))), ("randn", rng.standard_normal((n, 1))), ("rand*3", 3 * rng.random((n, 1)))):
    x = x.astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    grid = torch.zeros((1,) + (M,) * d, device="cuda")
    scratch = torch.empty(n + 256, device="cuda")
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(xt), 1, p(grid), p(scratch), s))
    ref = nfft_ref.spread(x, pos, batch, N, m).real.reshape((1,) + (M,) * d)
    got = grid.cpu().numpy().astype(np.float64)
    print(name, "rel_l2", np.linalg.norm(got - ref) / np.linalg.norm(ref), "max", np.abs(got - ref).max() / np.abs(ref).max())
    err = np.abs(got - ref)[0]
    idx = np.argsort(err.ravel())[::-1][:8]
    for i in idx:
        c = np.unravel_index(i, err.shape)
        print("   cell", c, "got", got[0][c], "ref", ref[0][c], "err/max", err[c] / np.abs(ref).max())
    c = np.unravel_index(idx[0], err.shape)
    np.set_printoptions(linewidth=200, precision=2)
    d3 = (got - ref)[0] / np.abs(ref).max()
    print("   row profile (cols c-6..c+6):", d3[c[0], c[1], max(0, c[2] - 6):c[2] + 7])
    print("   col profile (rows r-6..r+6):", d3[c[0], max(0, c[1] - 6):c[1] + 7, c[2]])
    print("   plane profile (z-6..z+6):", d3[max(0, c[0] - 6):c[0] + 7, c[1], c[2]])
    print("   fraction of cells with err > 1e-6 max:", (err > 1e-6 * np.abs(ref).max()).mean())
# single point: per-tap relative errors
for trial in range(3):
    pos1 = (rng.random((1, d)) - 0.5).astype(np.float32)
    prob1 = _lib.Problem(d, 1, 1, 1, N, m)
    post1 = torch.from_numpy(pos1).cuda()
    _lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob1), p(post1), None, p(plan), plan.numel(), s))
    x1 = np.array([[0.7]], dtype=np.float32)
    grid = torch.zeros((1,) + (M,) * d, device="cuda")
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob1), p(plan), p(torch.from_numpy(x1).cuda()), 1, p(grid), p(scratch), s))
    ref = nfft_ref.spread(x1, pos1, np.zeros(1, dtype=np.int64), N, m).real.reshape((1,) + (M,) * d)
    got = grid.cpu().numpy().astype(np.float64)
    nz = ref != 0
    rel = np.abs(got - ref)[nz] / np.abs(ref[nz])
    print("single point: taps", nz.sum(), "nonzero got", (got != 0).sum(), "max abs err/max", np.abs(got - ref).max() / ref.max(),
          "rel err of taps > 1e-3 max: ", rel[np.abs(ref[nz]) > 1e-3 * ref.max()].max())
