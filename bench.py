#!/usr/bin/env python3
"""Benchmark of the NFFT hot path on MI355X (contract: see the task's bench.py section).

Headline workload (BASELINE.json metric "Mpoints/s (adjoint+forward, 3-D N=256 m=4)"): config C3 -- d=3, N=256,
m=4, n=10^7 uniform points, one point set, real fp32 coefficients.  One "step" = nfft_adjoint(x) followed by
nfft_forward(of that spectrum, real_output=True), i.e. one pass of the hot path in each direction over one batch
of synthetic input, inputs resident in HBM, the point plan rebuilt every step.  value = n_gpus * n / t_step / 1e6.

The JSON line also carries, under "configs", one leg per other BASELINE.json configuration that fits one GPU
(C1, C2, one GPU's share of C4, C5) and the clustered variant of C3, each with the median of
>= 10 individually timed steps and its stage split -- parity cases first, timing legs second, never the headline.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a torch.distributed.run child
process, spawned before this process touches the GPU); under a launcher (WORLD_SIZE set) it is one rank of the job.
N > 1: every rank runs the C3 workload on its own point set (the batch axis is the sharding axis of this path; a
single point set cannot be split without a distributed FFT), no data-path collective, "scaling": "weak"; plus the
C4 leg sharded for real: `torch_nfft_amd.distributed` over B = 4 N point sets x 64 columns (B = 32 at N = 8, the
configuration C4 names) with the RCCL all-gather of the per-rank spectra.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_FMA_PER_S = 78.65e12      # 157.3 TFLOP/s fp32 vector = 78.65e12 FMA/s
MFMA_F16_FLOPS = 2.5e15        # dense f16 matrix peak (MI355X_MICROARCH.md)
ARITHMETIC = ("fp32 in / fp32 out; window sums on v_mfma_f32_32x32x16_f16 with two-way f16-split operands "
              "(hi*hi + hi*lo + lo*hi, ~22 significant bits) and fp32 accumulation; FFT and roll-off in fp32")
KERNEL_SOURCES = ("spread_mfma.hip", "common.h", "mfma_split.h", "window.h")
TRAFFIC_RECORD = "r04_spread_traffic.json"
# gather kernel at C3: FETCH_SIZE x 2 + WRITE_SIZE from the committed PMC passes (builder-measured)
# (8.532e5 KiB x 2 fetched + 3.205e5 KiB written per launch of interp_stream_kernel<10, false, 3>)
INTERP_TRAFFIC = {"bytes": int((8.532e5 * 2 + 3.205e5) * 1024),
                  "source": "profiles/r04_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE passes of this command at "
                            "C3, builder-measured in round 4, replayed here; not measured by this run)"}
TRAFFIC_SOURCE = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE (x2, the gfx950 correction) and WRITE_SIZE passes of this command, "
                  "taken by the builder (scripts/profile_round.sh) and replayed here while the kernel's name, the workload and "
                  "the hash of the kernel's source files still match; null otherwise -- not measured by this run" % TRAFFIC_RECORD)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--bandwidth", type=int, default=256)
    ap.add_argument("--cutoff", type=int, default=4)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--dist", choices=["uniform", "clusters"], default="uniform",
                    help="point distribution of the headline workload: uniform on the torus (the metric's workload) "
                         "or 8 Gaussian clusters (sigma 0.05, SURVEY.md 8(d)'s robustness case)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the per-config legs (headline line only)")
    ap.add_argument("--legs", default="c1,c2,c3clustered,c4share,c5,r1,r2,r3", help="comma-separated legs to run at N=1")
    ap.add_argument("--leg-reps", type=int, default=11, help="individually timed steps per leg (median reported)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 outside a launcher: become the parent of a torch.distributed.run job.  Nothing in this process
    has touched the GPU yet (torch is not even imported); the child's exit code is ours."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def kernel_source_hash():
    """Hash of the dominant kernel's source files without their // comments and blank lines (an edited comment does not
    make a measured traffic figure stale, an edited statement does)."""
    h = hashlib.sha1()
    for name in KERNEL_SOURCES:
        path = os.path.join(ROOT, "torch_nfft_amd", "csrc", name)
        if os.path.exists(path):
            with open(path, "r", encoding="utf-8", errors="replace") as f:
                for line in f:
                    code = line.split("//", 1)[0].strip()
                    if code:
                        h.update(code.encode("utf-8"))
                        h.update(b"\n")
    return h.hexdigest()[:16]


def measured_traffic(kernel, workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE with the gfx950
    x2 correction + WRITE_SIZE; written by scripts/pmc_traffic.py into profiles/).  Used only when the record was
    taken for this kernel, this workload and THIS kernel source (hash of the kernel's source files): after any edit
    of the kernel the figure is stale and the line says null until the PMC passes are re-run."""
    try:
        with open(os.path.join(ROOT, "profiles", TRAFFIC_RECORD)) as f:
            t = json.load(f)
        if t.get("kernel") == kernel and t.get("workload") == workload and t.get("source_hash") == kernel_source_hash():
            return t["fetch_bytes_corrected"] + t["write_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(dim, N, target_s):
    """Oracle (this repo's restatement of torch_nfft/ndft.py, plain C + OpenMP) timed on the host cores on a
    bounded sample of the same workload: all N^dim frequencies, n' points (cost is linear in n')."""
    import numpy as np
    from oracle import ndft_cpu
    cores = ndft_cpu.max_threads()
    rng = np.random.default_rng(20240)

    def run(npts):
        pos = (rng.random((npts, dim)) - 0.5).astype(np.float32)
        x = rng.random((npts, 1))
        t0 = time.perf_counter()
        y = ndft_cpu.ndft_adjoint(x, pos, None, N=N)
        t1 = time.perf_counter()
        ndft_cpu.ndft_forward(y, pos, None)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    ta, tf = run(64)  # calibration
    per_point = (ta + tf) / 64
    npts = int(max(64, min(1_000_000, target_s / max(per_point, 1e-9))))
    ta, tf = run(npts)
    return {
        "value": npts / (ta + tf) / 1e6,
        "unit": "Mpoints/s",
        "cores": cores,
        "kind": "port",
        "sample": "exact NDFT adjoint+forward (oracle/ndft_c.c, float64, OpenMP), all %d^%d frequencies, %d points: "
                  "adjoint %.2f s + forward %.2f s" % (N, dim, npts, ta, tf),
    }


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (NFFT_BENCH_FORCE_DIST=1 under a 1-rank launcher runs the multi-rank code path on one GPU: a rehearsal)
    distributed = world > 1 or os.environ.get("NFFT_BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    json_out = sys.stdout
    if distributed:
        # RCCL prints a version banner on the C-level stdout when its first communicator comes up: send everything
        # that is not the result line to stderr, so that rank 0's stdout carries exactly ONE line, the JSON
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    n_gpus = world

    import torch_nfft_amd as tn
    from torch_nfft_amd import _lib, ops
    from torch_nfft_amd import distributed as tnd

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if not distributed:
            return v
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def make_points(n, d, kind, gen):
        if kind == "uniform":
            return torch.rand((n, d), generator=gen, device=dev) - 0.5
        centres = torch.rand((8, d), generator=gen, device=dev) - 0.5
        which = torch.randint(0, 8, (n,), generator=gen, device=dev)
        pos = centres[which] + 0.05 * torch.randn((n, d), generator=gen, device=dev)
        return pos - torch.floor(pos + 0.5)  # back onto the torus

    def timed_series(step, reps, warm=2):
        """Median of `reps` individually synchronised steps (stage timers OFF: their event records cost ~70 us of
        stream time per step, half of a small leg), then the per-stage GPU times (HIP events on the launch stream)
        averaged per step over `reps` further steps."""
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        _lib.profile_enable(True)
        _lib.profile_collect()
        for _ in range(reps):
            step()
        stages = _lib.profile_collect()
        _lib.profile_enable(False)
        ts.sort()
        return ts[len(ts) // 2], ts[0], {k: v[0] / reps for k, v in stages.items() if v[1]}

    # ------------------------------------------------------------------ headline: C3
    d, N, m, n = args.dim, args.bandwidth, args.cutoff, args.points
    M = 2 * N
    gen = torch.Generator(device=dev).manual_seed(20240 + rank)
    pos = make_points(n, d, args.dist, gen)
    x = torch.rand((n,), generator=gen, device=dev)

    def step(fresh_plan=True):
        # One pass of the hot path in each direction over the same point set.  The point plan (tile binning) is
        # built once per step and shared by the adjoint and the forward transform of that step; it is dropped
        # at the start of every step, so no step reuses work of an earlier one.
        if fresh_plan:
            ops.plan_cache_clear()
        y = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
        return tn.nfft_forward(y, pos, None, cutoff=m, real_output=True)

    for _ in range(args.warmup):
        step()
    barrier()
    # Timed region: only the dominant kernel's stage carries its two HIP events (roofline.achieved is measured live,
    # on the launch stream); timers on all seven stages cost 40-70 us per step.  The full stage table comes from a
    # second pass over the same K steps.
    _lib.profile_enable(True, stages=("spread",))
    _lib.profile_collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    spread_live = _lib.profile_collect()["spread"]
    _lib.profile_enable(True)
    for _ in range(args.steps):
        step()
    stages = _lib.profile_collect()
    _lib.profile_enable(False)
    elapsed = max_over_ranks(elapsed)
    # the same step timed one at a time (median of >= 10), and with the plan kept across steps
    med_ms, min_ms, _ = timed_series(step, max(10, args.leg_reps), warm=0)
    cached_ms, _, _ = timed_series(lambda: step(fresh_plan=False), max(10, args.leg_reps), warm=1)

    legs = {}

    def guarded(name, fn):
        """A leg that fails is recorded as {"error": ...}; the headline line is printed whatever the legs do."""
        try:
            return fn()
        except Exception as e:  # noqa: BLE001 (a bench leg must not take the result line down with it)
            legs[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
            try:
                _lib.profile_enable(False)
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
            return None

    def leg_adjoint_forward(name, workload, d_, N_, m_, pos_, x_, batch_, units, unit_name, real_output=True):
        def st():
            ops.plan_cache_clear()
            y = tn.nfft_adjoint(x_, pos_, batch_, bandwidth=N_, cutoff=m_)
            return tn.nfft_forward(y, pos_, batch_, cutoff=m_, real_output=real_output)
        med, mn, per_stage = timed_series(st, args.leg_reps)
        legs[name] = {"workload": workload, "ms_per_step_median": med, "ms_per_step_min": mn,
                      "value": units / (med * 1e-3) / 1e6, "unit": unit_name, "stage_ms_per_step": per_stage}
        return per_stage

    want = set() if args.no_legs or distributed else set(s.strip() for s in args.legs.split(",") if s.strip())
    g2 = torch.Generator(device=dev).manual_seed(777)

    def leg_c1():
        p1 = make_points(1000, 1, "uniform", g2)
        leg_adjoint_forward("C1", "1-D adjoint+forward, N=64, m=2, 1 000 uniform points, one point set (launch-latency "
                            "bound: 1 KiB grid)", 1, 64, 2, p1, torch.rand((1000,), generator=g2, device=dev), None,
                            1000, "Mpoints/s")

    def leg_c2():
        p2 = make_points(100_000, 2, "uniform", g2)
        leg_adjoint_forward("C2", "2-D adjoint+forward, N=128, m=4, 100 000 uniform points, batch_size=1", 2, 128, 4,
                            p2, torch.rand((100_000,), generator=g2, device=dev), None, 100_000, "Mpoints/s")

    def leg_c3clustered():
        pc = make_points(n, 3, "clusters", g2)
        leg_adjoint_forward("C3-clustered", "C3 with %d points in 8 Gaussian clusters (sigma 0.05) instead of uniform" % n,
                            3, N, m, pc, x, None, n, "Mpoints/s")

    def leg_c4share():
        B4, C4, n4 = 4, 64, 100_000
        p4 = make_points(B4 * n4, 3, "uniform", g2)
        b4 = torch.arange(B4 * n4, device=dev) // n4
        x4 = torch.randn((B4 * n4, C4), generator=g2, device=dev)
        st4 = leg_adjoint_forward("C4-share", "one GPU's share of C4 (B=32 over 8 GPUs): 3-D adjoint+forward, N=128, m=4, "
                                  "4 point sets x 100 000 uniform points (assumed; BASELINE.json gives no n), 64 real "
                                  "coefficient columns, forward with real_output", 3, 128, 4, p4, x4, b4,
                                  B4 * n4 * C4, "M point-columns/s")
        # every (set, column) grid of 256^3 floats is written once -- but the stage's time follows its K-block count, not these
        # bytes (the same 17.2 GB take 4.9 ... 25.7 ms as the points per set go from 12 500 to 400 000: profiles/r04_experiments.md)
        alg4 = B4 * n4 * (4 * 3 + 4 * C4) + 8 * B4 * n4 + B4 * C4 * (256 ** 3) * 4
        sp4 = st4.get("spread", 0.0) + st4.get("zero", 0.0)
        if sp4 > 0:
            legs["C4-share"]["roofline"] = {
                "stage": "zero-fill + spreading (all kernels between the sorted points and the finished grids)",
                "bound": "mfma", "bound_note": "the K-block pipeline of the matrix-core kernel (instruction issue + LDS round trips); "
                "achieved / peak / frac are HBM figures by bytes written", "algorithmic_bytes_per_step": alg4, "ms_per_step": sp4,
                "achieved": alg4 / (sp4 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg4 / (sp4 * 1e-3) / 1e9 / HBM_PEAK_GBS}

    def leg_c5():
        n5 = 1_000_000
        src = (torch.rand((n5, 3), generator=g2, device=dev) - 0.5) * 0.5  # radius-1/4 box (test_fastsum.py:17-18)
        tgt = (torch.rand((n5, 3), generator=g2, device=dev) - 0.5) * 0.5
        x5 = torch.rand((n5,), generator=g2, device=dev)
        co = tn.gaussian_analytic_coeffs(0.1, dim=3, N=256)

        def st5():
            ops.plan_cache_clear()
            return tn.nfft_fastsum(x5, co, src, tgt, cutoff=4)
        med, mn, per_stage = timed_series(st5, args.leg_reps)
        legs["C5"] = {"workload": "fastsum, Gaussian kernel sigma=0.1 (analytic coefficients), 10^6 sources x 10^6 "
                                  "targets, 3-D N=256, m=4 (one native call, plans rebuilt every step)",
                      "ms_per_step_median": med, "ms_per_step_min": mn, "value": (2 * n5) / (med * 1e-3) / 1e6,
                      "unit": "Mpoints/s (sources + targets)", "stage_ms_per_step": per_stage}

    # The reference's own stated regime (torch_nfft/nfft.py:150-156: d <= 3, N in {16, 32, 64}, many points, m <= 8; its tests run
    # d=2, N=16, 3 x 1000 points, 10 columns: test/test_adjoint.py:21-32) -- complex spectra back into the forward transform
    def leg_r1():
        B, npts = 8, 100_000
        p = make_points(B * npts, 3, "uniform", g2)
        b = torch.arange(B * npts, device=dev) // npts
        leg_adjoint_forward("R1", "reference regime: 3-D adjoint+forward, N=32, m=3 (the API default), 8 point sets x 100 000 "
                            "uniform points, one real column", 3, 32, 3, p, torch.rand((B * npts,), generator=g2, device=dev),
                            b, B * npts, "Mpoints/s", real_output=False)

    def leg_r2():
        p = make_points(1_000_000, 3, "uniform", g2)
        leg_adjoint_forward("R2", "reference regime: 3-D adjoint+forward, N=64, m=4, 10^6 uniform points, one point set",
                            3, 64, 4, p, torch.rand((1_000_000,), generator=g2, device=dev), None, 1_000_000, "Mpoints/s",
                            real_output=False)

    def leg_r3():
        B, npts, C = 3, 1000, 10
        p = (torch.rand((B * npts, 2), generator=g2, device=dev) - 0.5) * 0.5  # radius 1/4 (test_adjoint.py:24)
        b = torch.arange(B * npts, device=dev) // npts
        leg_adjoint_forward("R3", "the reference's test shape (test/test_adjoint.py:21-32): 2-D adjoint+forward, N=16, m=3, "
                            "3 point sets x 1 000 points, 10 real columns", 2, 16, 3, p,
                            torch.rand((B * npts, C), generator=g2, device=dev), b, B * npts * C, "M point-columns/s",
                            real_output=False)

    for key, name, fn in (("c1", "C1", leg_c1), ("c2", "C2", leg_c2), ("c3clustered", "C3-clustered", leg_c3clustered),
                          ("c4share", "C4-share", leg_c4share), ("c5", "C5", leg_c5), ("r1", "R1", leg_r1),
                          ("r2", "R2", leg_r2), ("r3", "R3", leg_r3)):
        if key in want and (key != "c3clustered" or (d, N, m) == (3, 256, 4)):
            guarded(name, fn)
            torch.cuda.empty_cache()

    # ------------------------------------------------------------------ N > 1: C4 sharded for real (RCCL all-gather)
    leg_hung = False
    if distributed:
        def leg_c4_sharded():
            Br, C4, n4 = 4, 64, 100_000           # per rank: 4 point sets (B = 4 N in all, 32 at N = 8: the configuration C4 names)
            B4 = Br * world
            g4 = torch.Generator(device=dev).manual_seed(4242 + rank)  # every rank makes ITS point sets: nothing is replicated
            p4 = torch.rand((Br * n4, 3), generator=g4, device=dev) - 0.5
            b4 = torch.arange(Br * n4, device=dev) // n4
            x4 = torch.randn((Br * n4, C4), generator=g4, device=dev)
            # The timed pipeline keeps inputs and spectra sharded: adjoint(inputs_are_local, gather=False) -> this rank's
            # [B_r, N^3, C] slab -> forward(inputs_are_local) -> all-gather of the [n, C] rows (102 MB per rank).  Nothing
            # replicates the points or the 34.4 GB of spectra.  The all-gather of the spectra is timed apart, as the optional
            # step it is (callers that want the full spectrum on every rank).
            t_adj, t_fwd, t_total, t_gather = [], [], [], []
            for it in range(2 + 5):
                ops.plan_cache_clear()
                barrier()
                t0 = time.perf_counter()
                ya = tnd.nfft_adjoint(x4, p4, b4, bandwidth=128, cutoff=4, gather=False, inputs_are_local=True,
                                      local_batch_size=Br)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                yf = tnd.nfft_forward(ya, p4, b4, cutoff=4, real_output=True, gather=True, inputs_are_local=True)
                barrier()
                t2 = time.perf_counter()
                if it >= 2:
                    t_adj.append(max_over_ranks(t1 - t0)); t_fwd.append(max_over_ranks(t2 - t1))
                    t_total.append(max_over_ranks(t2 - t0))
                del yf
                if it >= 4:  # optional: the full spectrum on every rank
                    barrier()
                    t3 = time.perf_counter()
                    yfull = tnd._all_gather_rows(ya, [Br] * world, None)
                    barrier()
                    t_gather.append(max_over_ranks(time.perf_counter() - t3))
                    del yfull
                del ya
            med = lambda v: sorted(v)[len(v) // 2] * 1e3
            gathered_bytes = B4 * (128 ** 3) * C4 * 8
            legs["C4-sharded"] = {
                "workload": "C4 sharded over %d GPUs: B=%d point sets x 100 000 points, 64 real columns, N=128, m=4; every "
                            "rank holds only its 4 point sets (inputs_are_local); torch_nfft_amd.distributed adjoint (spectra "
                            "stay sharded) -> forward on the rank's own slab + all-gather of the rows (RCCL)" % (world, B4),
                "n_gpus": world, "ms_per_step_median": med(t_total), "ms_adjoint_local": med(t_adj),
                "ms_forward_and_row_gather": med(t_fwd),
                "optional_all_gather_of_spectra": {
                    "ms": med(t_gather), "bytes_gathered_per_rank": gathered_bytes,
                    "GBps_received_per_rank": gathered_bytes * (world - 1) / world / (med(t_gather) * 1e-3) / 1e9},
                "value": B4 * n4 * C4 / (med(t_total) * 1e-3) / 1e6, "unit": "M point-columns/s"}

        # The leg runs in a helper thread that the main thread waits for with a limit: an exception is recorded like that
        # of any other leg, and a collective that never returns (the one failure a try/except cannot see) leaves the
        # headline line intact -- the process then leaves through os._exit once the line is out.
        import threading

        def sharded_leg_thread():
            torch.cuda.set_device(local_rank)  # (the current device is per host thread: a new thread starts on device 0)
            guarded("C4-sharded", leg_c4_sharded)
        worker = threading.Thread(target=sharded_leg_thread, daemon=True)
        worker.start()
        worker.join(float(os.environ.get("NFFT_BENCH_LEG_TIMEOUT", "240")))
        if worker.is_alive():
            leg_hung = True
            legs["C4-sharded"] = {"error": "timeout: the sharded leg did not return (a collective is stuck?)"}

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = n_gpus * n / (elapsed / args.steps) / 1e6
        # dominant kernel: the spreading kernel (stage "spread" = one launch of it per step, plus its empty work-list
        # launch): the matrix-core kernel for 3-D grids of 64^3 and up with m <= 7, else spread_kernel
        W = 2 * m + 2
        mfma = d == 3 and M >= 64 and W <= 16 and os.environ.get("NFFT_HIP_SPREAD", "m")[:1] not in ("l", "r")
        kname = "spread_mfma_kernel<%d, false, false, false>" % W if mfma else "spread_kernel<%d,%d>" % (d, W)
        # matrix flops the kernel issues per tap row: 3 MFMA terms x 2 x 32 x 64 x 16 per (plane, 16 points) -- half of it for
        # a K-block whose windows lie in one 32-column half of the tile (the plan orders slabs by column group when the
        # work items are big; K-blocks that straddle a group boundary do both halves, so this is a slight underestimate)
        T2 = 65 - W
        grouped = mfma and n / (5.4 * 256) >= 3000 and os.environ.get("NFFT_HIP_COLGROUPS", "1") != "0"
        half_tiles = (0.5 * (33 - W) + 0.5 * (T2 - 32) + (W - 1)) / T2 if grouped else 1.0
        mfma_flops = n * W * 3 * 2 * 32 * 64 * half_tiles if mfma else 0
        sp_ms, sp_cnt = spread_live
        sp_avg = sp_ms / max(sp_cnt, 1)
        alg_bytes = n * (4 * d + 4) + (M ** d) * 4  # every point read once, real grid written once
        achieved = alg_bytes / (sp_avg * 1e-3) / 1e9 if sp_avg > 0 else 0.0
        taps = n * (2 * m + 2) ** d
        per_stage = {k: (v[0] / max(v[1], 1)) for k, v in stages.items() if v[1]}
        pipe_ms = sp_avg + per_stage.get("gather", 0.0) + per_stage.get("zero", 0.0)
        plan_ms = per_stage.get("plan", 0.0)
        workload = "C3: %d-D adjoint+forward, N=%d, m=%d, %d %s points per GPU, batch_size=1 per GPU, real fp32 x, " \
                   "forward with real_output" % (d, N, m, n, "uniform" if args.dist == "uniform"
                                                 else "clustered (8 Gaussian clusters)")
        out = {
            "metric": "Mpoints/s (adjoint+forward, 3-D N=256 m=4)",
            "value": value,
            "unit": "Mpoints/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "ms_per_step_median": med_ms,
            "ms_per_step_min": min_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "arithmetic": ARITHMETIC,
            "data": "synthetic",
            "config": {
                "workload": workload,
                "points_per_gpu": n, "bandwidth": N, "cutoff": m, "dim": d,
                "parallelism": "batch-sharded x%d (one point set per GPU, no collective)" % n_gpus,
            },
            "roofline": {
                "kernel": kname,
                # The figures below are the north star's: algorithmic HBM bytes over the launch time against the HBM peak.
                # What BINDS the kernel is not HBM: its window sums are matrix work (0.86 PFLOP issued per launch) fed by
                # vector instructions, and by the SQ counters the SIMDs spend ~80 % of the launch issuing the two
                # (profiles/r03_pmc_sq_counters.txt, profiles/r04_experiments.md); `issue_roof` carries that side.
                "bound": "mfma" if mfma else "hbm",
                "bound_note": "matrix + vector instruction issue (they do not overlap on a SIMD); achieved / peak / frac are "
                              "the HBM figures BASELINE.json's metric asks for, issue_roof the binding side",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(kname, workload),
                "traffic_source": TRAFFIC_SOURCE,
                "issue_roof": {
                    "mfma_f16_issued_tflops": (mfma_flops / (sp_avg * 1e-3)) / 1e12 if sp_avg > 0 else 0.0,
                    "mfma_f16_peak_tflops": MFMA_F16_FLOPS / 1e12,
                    "frac_of_mfma_f16_peak": (mfma_flops / (sp_avg * 1e-3)) / MFMA_F16_FLOPS if sp_avg > 0 else 0.0,
                    "valu_plus_mfma_busy_frac_of_simd_cycles": 0.80,
                    "busy_frac_source": "profiles/r04_pmc_sq_counters.txt ((SQ_ACTIVE_INST_VALU x 4 + SQ_VALU_MFMA_BUSY_CYCLES) / (1 024 "
                                        "SIMDs x launch duration x 2.0 GHz): (1.41e9 + 0.84e9) / 2.83e9; builder-measured, not "
                                        "re-measured by this run)",
                },
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": sp_avg,
                "launches": sp_cnt,
                "taps_per_s": taps / (sp_avg * 1e-3) if sp_avg > 0 else 0.0,
                "frac_of_valu_fma_peak": (taps / (sp_avg * 1e-3)) / VALU_FMA_PER_S if sp_avg > 0 else 0.0,
                "frac_of_mfma_f16_peak": (mfma_flops / (sp_avg * 1e-3)) / MFMA_F16_FLOPS if sp_avg > 0 else 0.0,
                # SURVEY.md 8(d) "metric 2": the same bytes over every kernel between (pos, x) and the finished grid
                # (coefficient gather + zero fill + spreading), and with the point plan on top
                "achieved_incl_gather_zero": alg_bytes / (pipe_ms * 1e-3) / 1e9 if pipe_ms > 0 else 0.0,
                "achieved_incl_gather_zero_plan": alg_bytes / ((pipe_ms + plan_ms) * 1e-3) / 1e9 if pipe_ms > 0 else 0.0,
            },
            # the forward gather (interp_stream_kernel at this size): every point read once, its result written once, the
            # complex grid -- here a real one, the C2R output -- read once: SURVEY.md 8(d) counts n (4 d + 4 C) + M^d 8
            "roofline_interp": {
                "kernel": "interp_stream_kernel<%d, false, 3>" % W if mfma else "interp_kernel<%d,%d>" % (d, W),
                "bound": "mfma" if mfma else "hbm",
                "bound_note": "as the spreading kernel: matrix + vector instruction issue of its consumer waves",
                "algorithmic_bytes_per_launch": n * (4 * d + 4) + (M ** d) * 8,
                "avg_launch_ms": per_stage.get("interp", 0.0),
                "achieved": (n * (4 * d + 4) + (M ** d) * 8) / (per_stage["interp"] * 1e-3) / 1e9 if per_stage.get("interp") else 0.0,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (n * (4 * d + 4) + (M ** d) * 8) / (per_stage["interp"] * 1e-3) / 1e9 / HBM_PEAK_GBS if per_stage.get("interp") else 0.0,
                "traffic": INTERP_TRAFFIC.get("bytes") if (d, N, m, n) == (3, 256, 4, 10_000_000) else None,
                "traffic_source": INTERP_TRAFFIC.get("source"),
            },
            "stage_ms_per_launch": per_stage,
            "stage_timers": "timed region: HIP events around the spreading stage only (roofline.avg_launch_ms); the stage "
                            "table comes from a second pass of the same steps with all seven stage timers, which cost "
                            "40-70 us per step; the legs' medians are taken with the timers off",
            "value_with_plan_kept_across_steps": n_gpus * n / (cached_ms * 1e-3) / 1e6,
            "configs": legs,
        }
        if not args.no_cpu_baseline and not distributed:
            out["cpu_baseline"] = cpu_baseline(d, N, args.cpu_seconds)
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if distributed:
        if leg_hung:
            sys.stderr.flush()
            os._exit(0)  # (a stuck collective: no barrier, no tear-down -- the result line is out)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
