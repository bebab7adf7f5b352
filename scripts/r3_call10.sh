#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "many_columns or c4 or wave_per_column" > $O/r3_t10a.log 2>&1
timeout -k 10 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs c4share > $O/r3_bench5.json 2> $O/r3_bench5.err
NFFT_HIP_COL_PLANAR=1 timeout -k 10 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs c4share > $O/r3_bench5_planar.json 2> $O/r3_bench5_planar.err
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3_t10.log 2>&1
