#!/bin/bash
# round 3, final measurement call: default bench line, one-rank rehearsal of the multi-rank path, profile passes
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python3 bench.py > $O/r3_final_bench.json 2> $O/r3_final_bench.err
NFFT_BENCH_FORCE_DIST=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/r3_final_dist.json 2> $O/r3_final_dist.err
bash scripts/profile_round.sh > $O/r3_final_profile.log 2>&1
