#!/usr/bin/env python3
"""Benchmark of the NFFT hot path on MI355X (contract: see the task's bench.py section).

Workload (BASELINE.json metric "Mpoints/s (adjoint+forward, 3-D N=256 m=4)"): config C3 -- d=3, N=256,
m=4, n=10^7 uniform points, one point set, real fp32 coefficients.  One "step" = nfft_adjoint(x) followed
by nfft_forward(of that spectrum, real_output=True), i.e. one pass of the hot path in each direction over
one batch of synthetic input, inputs resident in HBM.  value = n_gpus * n / t_step / 1e6.

N > 1 GPUs (launched by torch.distributed.run): every rank runs the same-sized workload on its own point
set (the batch axis is the sharding axis of this path; a single point set cannot be split without a
distributed FFT), no data-path collective, "scaling": "weak".
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_FMA_PER_S = 78.65e12      # 157.3 TFLOP/s fp32 vector = 78.65e12 FMA/s
MFMA_F16_FLOPS = 2.5e15        # dense f16 matrix peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--bandwidth", type=int, default=256)
    ap.add_argument("--cutoff", type=int, default=4)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--dist", choices=["uniform", "clusters"], default="uniform",
                    help="point distribution: uniform on the torus (the metric's workload) or 8 Gaussian clusters "
                         "(sigma 0.05, SURVEY.md 8(d)'s robustness case)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def measured_traffic(d, N, m, n):
    """HBM bytes per launch of the spreading kernel from the committed rocprofv3 PMC passes (FETCH_SIZE with the
    gfx950 x2 correction + WRITE_SIZE; profiles/r01_v6_spread_traffic.json), if they were taken on this workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_v6_spread_traffic.json")) as f:
            t = json.load(f)
        w = t["workload"]
        if (w["dim"], w["bandwidth"], w["cutoff"], w["points"]) == (d, N, m, n):
            return t["fetch_bytes_corrected"] + t["write_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(dim, N, target_s):
    """Oracle (this repo's restatement of torch_nfft/ndft.py, plain C + OpenMP) timed on the host cores on a
    bounded sample of the same workload: all N^dim frequencies, n' points (cost is linear in n')."""
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
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    n_gpus = world

    import torch_nfft_amd as tn
    from torch_nfft_amd import _lib

    d, N, m, n = args.dim, args.bandwidth, args.cutoff, args.points
    M = 2 * N
    gen = torch.Generator(device=dev).manual_seed(20240 + rank)
    if args.dist == "uniform":
        pos = torch.rand((n, d), generator=gen, device=dev) - 0.5
    else:
        centres = torch.rand((8, d), generator=gen, device=dev) - 0.5
        which = torch.randint(0, 8, (n,), generator=gen, device=dev)
        pos = centres[which] + 0.05 * torch.randn((n, d), generator=gen, device=dev)
        pos = pos - torch.floor(pos + 0.5)  # back onto the torus
    x = torch.rand((n,), generator=gen, device=dev)

    from torch_nfft_amd import ops

    def step(fresh_plan=True):
        # One pass of the hot path in each direction over the same point set.  The point plan (tile binning) is
        # built once per step and shared by the adjoint and the forward transform of that step; it is dropped
        # at the start of every step, so no step reuses work of an earlier one.
        if fresh_plan:
            ops.plan_cache_clear()
        y = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
        return tn.nfft_forward(y, pos, None, cutoff=m, real_output=True)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    _lib.profile_enable(True)
    _lib.profile_collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    stages = _lib.profile_collect()
    _lib.profile_enable(False)
    # secondary figure: the same steps with the point plan kept across steps (iterative use on fixed points)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step(fresh_plan=False)
    barrier()
    elapsed_cached = time.perf_counter() - t1

    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = n_gpus * n / (elapsed / args.steps) / 1e6
        # dominant kernel: the spreading kernel (stage "spread" = exactly one launch of it per step): the matrix-core
        # kernel for 3-D grids of 64^3 and up with m <= 7 (unless NFFT_HIP_SPREAD selects another), else spread_kernel
        W = 2 * m + 2
        mfma = d == 3 and M >= 64 and W <= 16 and os.environ.get("NFFT_HIP_SPREAD", "m")[:1] not in ("l", "r")
        # (the matrix-core kernel has a second, "overflow" instantiation <W, true> that is launched right after it and
        # is empty unless the points are clustered; the stage timer covers both launches)
        kname = "spread_mfma_kernel<%d, false>" % W if mfma else "spread_kernel<%d,%d>" % (d, W)
        # matrix flops the kernel issues per tap row: 3 MFMA terms x 2 x 32 x 64 x 16 per (plane, 16 points)
        mfma_flops = n * W * 3 * 2 * 32 * 64 if mfma else 0
        sp_ms, sp_cnt = stages["spread"]
        sp_avg = sp_ms / max(sp_cnt, 1)
        alg_bytes = n * (4 * d + 4) + (M ** d) * 4  # every point read once, real grid written once
        achieved = alg_bytes / (sp_avg * 1e-3) / 1e9 if sp_avg > 0 else 0.0
        taps = n * (2 * m + 2) ** d
        per_stage = {k: (v[0] / max(v[1], 1)) for k, v in stages.items() if v[1]}
        pipe_ms = sp_avg + per_stage.get("gather", 0.0) + per_stage.get("zero", 0.0)
        plan_ms = per_stage.get("plan", 0.0)
        # both transforms run the point plan and an FFT: split by launch count for the report
        out = {
            "metric": "Mpoints/s (adjoint+forward, 3-D N=256 m=4)",
            "value": value,
            "unit": "Mpoints/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "C3: %d-D adjoint+forward, N=%d, m=%d, %d %s points per GPU, batch_size=1 per GPU, "
                            "real fp32 x, forward with real_output" % (d, N, m, n, "uniform" if args.dist == "uniform"
                                                                         else "clustered (8 Gaussian clusters)"),
                "points_per_gpu": n, "bandwidth": N, "cutoff": m, "dim": d,
                "parallelism": "batch-sharded x%d (one point set per GPU, no collective)" % n_gpus,
            },
            "roofline": {
                "kernel": kname,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(d, N, m, n),
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
            "stage_ms_per_launch": per_stage,
            "value_with_plan_kept_across_steps": n_gpus * n / (elapsed_cached / args.steps) / 1e6,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d, N, args.cpu_seconds)
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
