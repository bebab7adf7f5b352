"""Developer tool: where a pipeline step of spread_mfma_kernel goes, per wave role, from a trace build
(hipcc ... -DNFFT_HIP_TRACE on spread_mfma.hip; NFFT_HIP_LIB=<that library>): shader-clock stamps of the first 16 workgroups at the
loop top, after the accumulation, after the operand build and after the barrier of every step."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch_nfft_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
d, N, m, n = 3, 256, 4, 10_000_000
prob = _lib.Problem(d, n, 1, 1, N, m)
gen = torch.Generator(device="cuda").manual_seed(1)
pos = torch.rand((n, d), generator=gen, device="cuda") - 0.5
x = torch.rand((n,), generator=gen, device="cuda")
plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(pos), None, p(plan), plan.numel(), s))
grid = torch.empty((2 * N,) * 3, device="cuda")
scratch = torch.empty(n + 256, device="cuda")
WG, WAVES, STEPS = 16, 16, 64
trace = torch.zeros((WG, WAVES, STEPS, 4), dtype=torch.int64, device="cuda")
assert lib.nfft_dbg_set_step_trace(p(trace)) == 0
for it in range(3):
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
torch.cuda.synchronize()
trace.zero_()
_lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
torch.cuda.synchronize()
raw = trace.cpu().numpy()
ntask = (raw[..., 2] >> 56).astype(np.float64)  # build tasks the wave took in the step (packed into the stamp's top byte)
raw[..., 2] &= (1 << 56) - 1
t = raw.astype(np.float64)
ok = (t[..., 0] > 0) & (t[..., 3] > 0)
ok[:, :, :3] = False  # (pipeline fill)
NOWN = int(os.environ.get("NOWN", 12))  # plane-owner waves of the build under test
roles = {"owners (waves 0-%d)" % (NOWN - 1): range(0, NOWN), "stagers (waves %d, %d)" % (NOWN, NOWN + 1): (NOWN, NOWN + 1),
         "builders (waves %d-15)" % (NOWN + 2): range(NOWN + 2, 16)}
print("s_memtime ticks (100 MHz on gfx950 -> 10 ns each); steps 3..63 of the first 16 workgroups")
step = (t[..., 3] - t[..., 0])
print("step (top -> after barrier), all waves: mean %.0f ticks" % step[ok].mean())
for name, waves in roles.items():
    sel = np.zeros_like(ok); sel[:, list(waves)] = True; sel &= ok
    a = (t[..., 1] - t[..., 0])[sel].mean(); b = (t[..., 2] - t[..., 1])[sel].mean(); c = (t[..., 3] - t[..., 2])[sel].mean()
    print("%-26s staging/accumulate %.0f   build %.0f (%.2f tasks per step and wave)   wait at the barrier %.0f"
          % (name, a, b, ntask[sel].mean(), c))
# the wave that arrives last at the barrier, per step
arrive = t[..., 2]
last = np.argmax(np.where(ok, arrive, 0), axis=1)  # [wg, step]
cnt = np.bincount(last[ok.any(axis=1)].ravel(), minlength=16)
print("last wave at the barrier (count per wave):", cnt.tolist())
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "spread_steps.npy"), t)
