#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
NFFT_HIP_LIB=scripts/ubench/libnfft_cur4.so timeout -k 10 120 python3 scripts/dbg_small_wide.py > $O/r3_dbg8.log 2>&1
ROUNDS=3 timeout -k 10 900 python3 scripts/ab_stage.py scripts/ubench/libnfft_r2.so scripts/ubench/libnfft_idord.so scripts/ubench/libnfft_cur2.so scripts/ubench/libnfft_cur3.so scripts/ubench/libnfft_cur4.so > $O/r3_ab3.log 2>&1
CLUSTERS=1 ROUNDS=2 timeout -k 10 600 python3 scripts/ab_stage.py scripts/ubench/libnfft_cur2.so scripts/ubench/libnfft_cur3.so scripts/ubench/libnfft_cur4.so > $O/r3_ab3_clu.log 2>&1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3_t8.log 2>&1
