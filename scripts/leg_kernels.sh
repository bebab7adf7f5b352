#!/bin/bash
# Kernel list per small leg of bench.py (run on the GPU box through gpurun): rocprofv3 --kernel-trace --stats of the leg alone,
# top rows by call count / time -> gpurun_out/leg_kernels.txt (copied to profiles/<round>_leg_kernels.txt).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
: > $O/leg_kernels.txt
for leg in c2 r1 r3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/legk_$leg -- python3 bench.py --no-cpu-baseline --legs $leg --steps 1 --warmup 1 --points 100000 --leg-reps 10 > $O/legk_$leg.json 2> $O/legk_$leg.err
  {
    echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --legs $leg --steps 1 --warmup 1 --points 100000 --leg-reps 10"
    echo "# (22 leg steps: 2 warm-up + 10 timed + 10 with stage timers; the headline loop of 10^5 points contributes the 44-call rows)"
    python3 scripts/stats_top.py $O/legk_$leg 40
    echo
  } >> $O/leg_kernels.txt
done
tail -5 $O/leg_kernels.txt
