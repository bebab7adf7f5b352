#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
ROUNDS=3 timeout -k 10 900 python3 scripts/ab_stage.py scripts/ubench/libnfft_cur9.so scripts/ubench/libnfft_cur10.so > $O/r3_ab7.log 2>&1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3_t16.log 2>&1
timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/r3_bench9.json 2> $O/r3_bench9.err
