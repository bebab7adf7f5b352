#!/bin/bash
# round 3, first GPU call: spreading timeline (trace build), FETCH/WRITE calibration, baseline bench, GPU tests
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
NFFT_HIP_LIB=scripts/ubench/libnfft_trace.so timeout -k 10 300 python3 scripts/spread_trace.py > $O/r3_trace_c3.log 2>&1
CLUSTERS=1 TRACE_TAG=clu NFFT_HIP_LIB=scripts/ubench/libnfft_trace.so timeout -k 10 300 python3 scripts/spread_trace.py > $O/r3_trace_clu.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r3_calib_fetch -- scripts/ubench/fetch_calib > $O/r3_calib_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r3_calib_write -- scripts/ubench/fetch_calib > $O/r3_calib_write.log 2>&1
python3 scripts/pmc_summary.py $O/r3_calib_fetch > $O/r3_calib_fetch.txt
python3 scripts/pmc_summary.py $O/r3_calib_write > $O/r3_calib_write.txt
timeout -k 10 600 python3 bench.py --steps 10 --warmup 2 > $O/r3_bench0.json 2> $O/r3_bench0.err
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/r3_t1.log 2>&1
