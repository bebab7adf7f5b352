#!/bin/bash
# Round profile on the GPU box (run through gpurun): kernel stats of the default bench run and of the C4 leg, then the
# two PMC passes for the dominant kernel's HBM traffic.  Outputs under gpurun_out/r2p_*.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2p_stats -- python3 bench.py --no-legs --steps 10 --warmup 2 > $O/r2p_bench_line.json 2> $O/r2p_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2p_c4stats -- python3 bench.py --no-cpu-baseline --legs c4share --steps 2 --warmup 1 --points 1000000 > $O/r2p_c4_line.json 2> $O/r2p_c4.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r2p_pmc_fetch -- python3 bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1 > $O/r2p_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r2p_pmc_write -- python3 bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1 > $O/r2p_pmc_write.log 2>&1
python3 scripts/pmc_traffic.py $O/r2p_pmc_fetch $O/r2p_pmc_write $O/r2p_bench_line.json $O/r2p_spread_traffic.json > $O/r2p_traffic.log 2>&1
python3 scripts/pmc_summary.py $O/r2p_pmc_fetch > $O/r2p_pmc_fetch.txt
python3 scripts/pmc_summary.py $O/r2p_pmc_write > $O/r2p_pmc_write.txt
python3 scripts/stats_top.py $O/r2p_stats 30 > $O/r2p_stats_top.txt
python3 scripts/stats_top.py $O/r2p_c4stats 30 > $O/r2p_c4stats_top.txt
tail -5 $O/r2p_traffic.log
