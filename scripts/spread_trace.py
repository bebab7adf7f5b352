"""Developer tool: per-workgroup timeline of spread_mfma_kernel at the C3 size from a trace build
(scripts/exp_build.sh spread_mfma.hip trace:torch_nfft_amd/csrc/spread_mfma.hip:"-DNFFT_HIP_TRACE";
NFFT_HIP_LIB=scripts/ubench/libnfft_trace.so).  Prints how busy the CUs are over the launch, the share of the
per-item prologue and the tail, and saves the raw stamps under gpurun_out/."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch_nfft_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
d, N, m, n = 3, int(os.environ.get("NBAND", 256)), int(os.environ.get("M_CUT", 4)), int(os.environ.get("NPTS", 10_000_000))
prob = _lib.Problem(d, n, 1, 1, N, m)
gen = torch.Generator(device="cuda").manual_seed(1)
if os.environ.get("CLUSTERS") == "1":
    centres = torch.rand((8, d), generator=gen, device="cuda") - 0.5
    which = torch.randint(0, 8, (n,), generator=gen, device="cuda")
    pos = centres[which] + 0.05 * torch.randn((n, d), generator=gen, device="cuda")
    pos = pos - torch.floor(pos + 0.5)
else:
    pos = torch.rand((n, d), generator=gen, device="cuda") - 0.5
x = torch.rand((n,), generator=gen, device="cuda")
plan = torch.empty(lib.nfft_hip_plan_bytes(ctypes.byref(prob)), dtype=torch.uint8, device="cuda")
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.nfft_hip_plan_points(ctypes.byref(prob), p(pos), None, p(plan), plan.numel(), s))
grid = torch.empty((2 * N,) * 3, device="cuda")
scratch = torch.empty(n + 256, device="cuda")
nwg = 1 << 16
trace = torch.zeros((nwg, 8), dtype=torch.int64, device="cuda")
assert lib.nfft_dbg_set_spread_trace(p(trace)) == 0
for it in range(3):
    _lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
torch.cuda.synchronize()
trace.zero_()
torch.cuda.synchronize()
t0 = time.perf_counter()
_lib.check(lib.nfft_hip_spread(ctypes.byref(prob), p(plan), p(x), 1, p(grid), p(scratch), s))
torch.cuda.synchronize()
print("call: %.3f ms (zero-fill + spreading + overflow launch)" % ((time.perf_counter() - t0) * 1e3))
t = trace.cpu().numpy().astype(np.int64)
ran = t[:, 0] != 0
t = t[ran]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(out, exist_ok=True)
np.save(os.path.join(out, "spread_trace_%s.npy" % os.environ.get("TRACE_TAG", "c3")), t)
tick = 10.0  # ns per stamp (100 MHz)
start, xmax, loop, end, hw, cnt, preflush = (t[:, k] for k in (0, 1, 2, 3, 4, 5, 6))
worked = end != 0
T0, T1 = start.min(), max(end.max(), start.max())
print("workgroups launched %d, with work %d; kernel span %.1f us" % (len(t), worked.sum(), (T1 - T0) * tick / 1e3))
cu = (hw & 0xffffffff)
cu_key = ((hw >> 32) & 0xf) * 1000 + ((cu >> 13) & 7) * 100 + ((cu >> 12) & 1) * 16 + ((cu >> 8) & 0xf)  # xcc, se, sh, cu
keys = np.unique(cu_key)
print("distinct CUs seen: %d" % len(keys))
busy = np.zeros(len(keys)); last = np.zeros(len(keys)); first = np.zeros(len(keys)); items = np.zeros(len(keys), dtype=int)
for i, k in enumerate(keys):
    sel = (cu_key == k) & worked
    busy[i] = (end[sel] - start[sel]).sum() * tick / 1e3
    last[i] = (end[sel].max() - T0) * tick / 1e3 if sel.any() else 0
    first[i] = (start[cu_key == k].min() - T0) * tick / 1e3
    items[i] = sel.sum()
span = (T1 - T0) * tick / 1e3
print("per CU: items min/mean/max %d / %.2f / %d; busy us min/mean/max %.0f / %.0f / %.0f; last end us min/mean/max %.0f / %.0f / %.0f; first start max %.1f"
      % (items.min(), items.mean(), items.max(), busy.min(), busy.mean(), busy.max(), last.min(), last.mean(), last.max(), first.max()))
print("CU utilisation over the kernel span: %.1f %%  (sum of busy / (CUs x span))" % (100 * busy.sum() / (len(keys) * span)))
w = worked
dur = (end[w] - start[w]) * tick / 1e3
pro1 = (xmax[w] - start[w]) * tick / 1e3
pro2 = (loop[w] - xmax[w]) * tick / 1e3
main = (preflush[w] - loop[w]) * tick / 1e3
tail = (end[w] - preflush[w]) * tick / 1e3
kb = (cnt[w] & 0xffffffff).astype(np.float64)
pts = (cnt[w] >> 32).astype(np.float64)
print("items: duration us mean %.1f (min %.1f max %.1f); max-|x| + permutation pass %.2f; schedule + first staging %.2f; main loop %.1f; final flush %.2f"
      % (dur.mean(), dur.min(), dur.max(), pro1.mean(), pro2.mean(), main.mean(), tail.mean()))
print("shares of the summed item time: prologue %.1f %%, main loop %.1f %%, final flush %.1f %%"
      % (100 * (pro1 + pro2).sum() / dur.sum(), 100 * main.sum() / dur.sum(), 100 * tail.sum() / dur.sum()))
print("K-blocks per item mean %.0f, points per item mean %.0f, fill %.3f; main-loop ns per K-block %.1f (over items with >= 100 K-blocks)"
      % (kb.mean(), pts.mean(), pts.sum() / (16 * kb.sum()), (main[kb >= 100] * 1e3 / kb[kb >= 100]).mean()))
# how many CUs are busy over time (20 bins)
edges = np.linspace(T0, T1, 21)
occ = []
for a, b_ in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(end[w], b_) - np.maximum(start[w], a), 0, None).sum() / (b_ - a)
    occ.append(ov)
print("workgroups resident over time (20 bins): " + " ".join("%.0f" % o for o in occ))
