#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3_t9.log 2>&1
ROUNDS=2 timeout -k 10 600 python3 scripts/ab_stage.py scripts/ubench/libnfft_r2.so scripts/ubench/libnfft_cur5.so > $O/r3_ab4.log 2>&1
CLUSTERS=1 ROUNDS=2 timeout -k 10 600 python3 scripts/ab_stage.py scripts/ubench/libnfft_r2.so scripts/ubench/libnfft_cur5.so > $O/r3_ab4_clu.log 2>&1
NFFT_HIP_LIB=scripts/ubench/libnfft_trace.so TRACE_TAG=c3final timeout -k 10 300 python3 scripts/spread_trace.py > $O/r3_trace4_c3.log 2>&1
CLUSTERS=1 NFFT_HIP_LIB=scripts/ubench/libnfft_trace.so TRACE_TAG=clufinal timeout -k 10 300 python3 scripts/spread_trace.py > $O/r3_trace4_clu.log 2>&1
timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 > $O/r3_bench4.json 2> $O/r3_bench4.err
