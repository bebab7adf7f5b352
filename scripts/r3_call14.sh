#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
ROUNDS=3 timeout -k 10 900 python3 scripts/ab_stage.py scripts/ubench/libnfft_cur5.so scripts/ubench/libnfft_cur8.so > $O/r3_ab5.log 2>&1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3_t14.log 2>&1
