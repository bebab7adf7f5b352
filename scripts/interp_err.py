"""Developer tool: error of the interpolation stage alone (nfft_hip_interpolate) against a float64 gather."""
import ctypes, sys, os, itertools
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from torch_nfft_amd import _lib
from oracle import nfft_ref

lib = _lib.load()
rng = np.random.default_rng(3)
d, N, m, n = 3, int(os.environ.get("NBAND", 32)), int(os.environ.get("M_CUT", 4)), int(os.environ.get("NPTS", 200000))
M = 2 * N
pos = (rng.random((n, d)) - 0.5).astype(np.float32)
p = lambda t: ctypes.c_void_p(t.data_ptr())
prob = _lib.Problem(d, n, 1, 1, N, m)
plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
post = torch.from_numpy(pos).cuda()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(post), None, p(plan), plan.numel(), s))
grid = rng.standard_normal((1,) + (M,) * d).astype(np.float32) * np.exp(rng.standard_normal((1, M, 1, 1)) * 3).astype(np.float32)
gt = torch.from_numpy(grid).cuda()
yr = torch.empty((n, 1), device="cuda")
_lib.check(lib.nfft_hip_interpolate(ctypes.byref(prob), p(plan), p(gt), 1, p(yr), s))
shift, psi = nfft_ref.window_taps(pos, N, m)
exp = np.zeros(n)
g64 = grid[0].astype(np.float64)
for ls in itertools.product(range(2 * m + 2), repeat=d):
    w = np.ones(n)
    idx = []
    for a, l in enumerate(ls):
        w = w * psi[:, a, l]
        idx.append((shift[:, a] + l + M) % M)
    exp += w * g64[tuple(idx)]
got = yr.cpu().numpy()[:, 0].astype(np.float64)
err = np.abs(got - exp)
print("rel_l2", np.linalg.norm(got - exp) / np.linalg.norm(exp), "max err / max", err.max() / np.abs(exp).max())
bad = np.argsort(err)[::-1][:10]
for i in bad:
    print("  point", i, "cell", (shift[i] + m) % M, "got", got[i], "exp", exp[i])
print("points with err > 1e-4 max:", int((err > 1e-4 * np.abs(exp).max()).sum()), "of", n)
