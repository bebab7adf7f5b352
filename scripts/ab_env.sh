#!/bin/bash
# Developer tool: ab_stage.py of ONE library under two environments (e.g. NFFT_HIP_WORK_LIST=0 / 1), interleaved by hand.
# usage: bash scripts/ab_env.sh lib.so VAR=a VAR=b
lib=$1; a=$2; b=$3
for r in 1 2 3; do
  env $a ROUNDS=1 python scripts/ab_stage.py $lib | sed "s/^/[$a] /"
  env $b ROUNDS=1 python scripts/ab_stage.py $lib | sed "s/^/[$b] /"
done
