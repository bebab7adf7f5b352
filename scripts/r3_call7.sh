#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for lib in cur2 cur3 cur4; do for ow in 0 1; do
echo "== $lib OWNED=$ow" >> $O/r3_dbg7.log
NFFT_HIP_OWNED=$ow NFFT_HIP_LIB=scripts/ubench/libnfft_$lib.so timeout -k 10 120 python3 scripts/dbg_small_wide.py >> $O/r3_dbg7.log 2>&1
done; done
