import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from torch_nfft_amd import _lib
from oracle import nfft_ref
lib = _lib.load()
rng = np.random.default_rng(5)
d, N, m = 3, 32, 4
M = 2 * N
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for trial in range(4):
    pos1 = (rng.random((1, d)) - 0.5).astype(np.float32)
    prob1 = _lib.Problem(d, 1, 1, 1, N, m)
    plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob1)), dtype=torch.uint8, device="cuda")
    post1 = torch.from_numpy(pos1).cuda()
    _lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob1), p(post1), None, p(plan), plan.numel(), s))
    x1 = np.array([[0.7]], dtype=np.float32)
    grid = torch.zeros((1,) + (M,) * d, device="cuda")
    scratch = torch.empty(512, device="cuda")
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob1), p(plan), p(torch.from_numpy(x1).cuda()), 1, p(grid), p(scratch), s))
    ref = nfft_ref.spread(x1, pos1, np.zeros(1, dtype=np.int64), N, m).real.reshape((M,) * d)
    got = grid.cpu().numpy()[0].astype(np.float64)
    shift, psi = nfft_ref.window_taps(pos1, N, m)
    cell = shift[0] % M
    nzr = np.argwhere(ref != 0)
    miss = np.argwhere((ref != 0) & (got == 0) & (np.abs(ref) > 1e-7 * ref.max()))
    rel = (miss - cell + M // 2) % M - M // 2
    print("first tap cell", cell, "missing idx axis2", sorted(set(miss[:, 2])), "axis1", sorted(set(miss[:, 1]))[:4], "axis0", sorted(set(miss[:, 0]))[:4])
    print("cell", cell, "missing taps:", len(miss), "offsets axis0", sorted(set(rel[:, 0])), "axis1", sorted(set(rel[:, 1])), "axis2", sorted(set(rel[:, 2])))
