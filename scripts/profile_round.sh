#!/bin/bash
# Round profile on the GPU box (run through gpurun): kernel stats of the default bench run and of the C4 leg, then the
# two PMC passes for the dominant kernel's HBM traffic.  Outputs under gpurun_out/r4p_*.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4p_stats -- python3 bench.py --no-legs --steps 10 --warmup 2 > $O/r4p_bench_line.json 2> $O/r4p_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4p_c4stats -- python3 bench.py --no-cpu-baseline --legs c4share --steps 2 --warmup 1 --points 1000000 > $O/r4p_c4_line.json 2> $O/r4p_c4.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r4p_pmc_fetch -- python3 bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1 > $O/r4p_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r4p_pmc_write -- python3 bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1 > $O/r4p_pmc_write.log 2>&1
python3 scripts/pmc_traffic.py $O/r4p_pmc_fetch $O/r4p_pmc_write $O/r4p_bench_line.json $O/r4p_spread_traffic.json > $O/r4p_traffic.log 2>&1
python3 scripts/pmc_summary.py $O/r4p_pmc_fetch > $O/r4p_pmc_fetch.txt
python3 scripts/pmc_summary.py $O/r4p_pmc_write > $O/r4p_pmc_write.txt
python3 scripts/stats_top.py $O/r4p_stats 30 > $O/r4p_stats_top.txt
python3 scripts/stats_top.py $O/r4p_c4stats 30 > $O/r4p_c4stats_top.txt
# SQ counters of the two matrix-core kernels (two passes)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/r4p_pmc_sq1 -- python3 bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1 > $O/r4p_pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/r4p_pmc_sq2 -- python3 bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1 > $O/r4p_pmc_sq2.log 2>&1
python3 scripts/pmc_summary.py $O/r4p_pmc_sq1 _kernel > $O/r4p_pmc_sq1.txt
python3 scripts/pmc_summary.py $O/r4p_pmc_sq2 _kernel > $O/r4p_pmc_sq2.txt
tail -5 $O/r4p_traffic.log
