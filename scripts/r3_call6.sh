#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3_t6.log 2>&1
ROUNDS=3 timeout -k 10 900 python3 scripts/ab_stage.py scripts/ubench/libnfft_r2.so scripts/ubench/libnfft_idord.so scripts/ubench/libnfft_cur2.so scripts/ubench/libnfft_cur3.so > $O/r3_ab2.log 2>&1
CLUSTERS=1 ROUNDS=2 timeout -k 10 600 python3 scripts/ab_stage.py scripts/ubench/libnfft_cur2.so scripts/ubench/libnfft_cur3.so > $O/r3_ab2_clu.log 2>&1
