"""Developer tool: spread a handful of points with the owner-computes kernel and list the cells that differ from the oracle."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch_nfft_amd import _lib
from oracle import nfft_ref
lib = _lib.load()
d, N, m, B, Cr = 3, 64, int(os.environ.get("M_CUT", 4)), 1, 1
M = 2 * N
cells = np.array(eval(os.environ.get("CELLS", "[[5,5,5]]")))
pos = ((cells + 0.3) / M - 0.5).astype(np.float32)
n = pos.shape[0]
x = np.ones((n, Cr), np.float32)
prob = _lib.Problem(d, n, Cr, B, N, m)
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
post, xt = torch.from_numpy(pos).cuda(), torch.from_numpy(x).cuda()
_lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(post), None, p(plan), plan.numel(), s))
scratch = torch.empty(lib.nfft_hip_spread_scratch_bytes(ctypes.byref(prob), Cr) // 4, device="cuda")
grid = torch.full((Cr, M, M, M), float("nan"), device="cuda")
_lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(xt), Cr, p(grid), p(scratch), s))
torch.cuda.synchronize()
got = grid.cpu().numpy()
ref = nfft_ref.spread(x, pos, None, N, m).real.reshape(Cr, M, M, M)
print("nan cells:", int(np.isnan(got).sum()), "rel err:", np.linalg.norm(np.nan_to_num(got) - ref) / np.linalg.norm(ref))
bad = np.argwhere(np.abs(np.nan_to_num(got, nan=1e9) - ref) > 1e-5 * np.abs(ref).max())
print("bad cells:", len(bad))
for b in bad[:40]:
    print(tuple(b), got[tuple(b)], ref[tuple(b)])
