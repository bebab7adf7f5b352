#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "many_columns or c4 or wave_per_column" > $O/r3_t11a.log 2>&1
timeout -k 10 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs c4share > $O/r3_bench6.json 2> $O/r3_bench6.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_c4stats -- python3 bench.py --no-cpu-baseline --legs c4share --steps 2 --warmup 1 --points 1000000 > $O/r3_c4_line.json 2> $O/r3_c4.err
python3 scripts/stats_top.py $O/r3_c4stats 30 > $O/r3_c4stats_top.txt
