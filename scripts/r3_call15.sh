#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
ROUNDS=3 timeout -k 10 900 python3 scripts/ab_stage.py scripts/ubench/libnfft_cur9.so scripts/ubench/libnfft_taskord.so > $O/r3_ab6.log 2>&1
CLUSTERS=1 ROUNDS=2 timeout -k 10 600 python3 scripts/ab_stage.py scripts/ubench/libnfft_cur9.so scripts/ubench/libnfft_taskord.so > $O/r3_ab6_clu.log 2>&1
