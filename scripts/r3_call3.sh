#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
NFFT_HIP_LIB=scripts/ubench/libnfft_trace.so TRACE_TAG=c3idx timeout -k 10 300 python3 scripts/spread_trace.py > $O/r3_trace3_c3.log 2>&1
timeout -k 10 600 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --legs c3clustered,c5 > $O/r3_bench3.json 2> $O/r3_bench3.err
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3_t4.log 2>&1
