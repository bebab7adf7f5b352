#!/bin/bash
# Copies what scripts/round_verify.sh left under gpurun_out/ into profiles/ under this round's names (run in the container
# after the gpurun call has merged its outputs back).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out
R=${1:-r04}
tail -1 $O/verify_bench.json > profiles/${R}_bench_line.json
cp "$(ls -t $O/r4p_stats/*/*kernel_stats.csv | head -1)" profiles/${R}_bench_kernel_stats.csv
cp "$(ls -t $O/r4p_c4stats/*/*kernel_stats.csv | head -1)" profiles/${R}_c4share_kernel_stats.csv
cp $O/r4p_spread_traffic.json profiles/${R}_spread_traffic.json
{
  echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1   (KiB per dispatch; x2 for bytes: profiles/r03_fetch_calibration.txt)"
  cat $O/r4p_pmc_fetch.txt
  echo "# the same with --pmc WRITE_SIZE (KiB per dispatch, exact)"
  cat $O/r4p_pmc_write.txt
} > profiles/${R}_pmc_hbm_traffic.txt
{
  echo "# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY (second pass: SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES) -- python3 bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1"
  echo "# SQ_*_CYCLES / ACTIVE / WAIT counters: units of 4 cycles summed over waves or SIMDs; SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the 1024 SIMDs"
  cat $O/r4p_pmc_sq1.txt
  cat $O/r4p_pmc_sq2.txt
} > profiles/${R}_pmc_sq_counters.txt
git status --short profiles | head
