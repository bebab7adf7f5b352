#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3_t13.log 2>&1
timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --legs c3clustered,c5 > $O/r3_bench8.json 2> $O/r3_bench8.err
NFFT_HIP_NO_OVERLAP=1 timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-legs > $O/r3_bench8_noov.json 2> $O/r3_bench8_noov.err
timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-legs > $O/r3_bench8_b.json 2> $O/r3_bench8_b.err
