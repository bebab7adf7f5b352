#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r3_calib2_fetch -- scripts/ubench/fetch_calib > $O/r3_calib2_fetch.log 2>&1
python3 scripts/pmc_summary.py $O/r3_calib2_fetch > $O/r3_calib2_fetch.txt
ROUNDS=3 timeout -k 10 600 python3 scripts/ab_stage.py scripts/ubench/libnfft_r2.so scripts/ubench/libnfft_cur.so > $O/r3_ab1.log 2>&1
CLUSTERS=1 ROUNDS=2 timeout -k 10 600 python3 scripts/ab_stage.py scripts/ubench/libnfft_r2.so scripts/ubench/libnfft_cur.so > $O/r3_ab1_clu.log 2>&1
