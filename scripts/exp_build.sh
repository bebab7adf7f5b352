#!/bin/bash
# Developer tool: build variant libraries scripts/ubench/libnfft_<name>.so from experimental copies of one kernel file.
# usage: scripts/exp_build.sh <file in torch_nfft_amd/csrc it replaces> name1:<experimental.hip>:"-DX=1" ...
# (the other objects come from torch_nfft_amd/_obj, i.e. run torch_nfft_amd/build.py first)
set -e
BASE=$1; shift
OBJS=""
for f in api.hip binning.hip spread.hip spread_reg.hip spread_mfma.hip interp.hip interp_mfma.hip interp_cols.hip interp_stream.hip smallgrid.hip spectral.hip colfft.hip coeffs.hip selftest.hip fft.cpp; do
  if [ "$f" != "$BASE" ]; then OBJS="$OBJS torch_nfft_amd/_obj/$f.o"; fi
done
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; src=${rest%%:*}; flags=${rest#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -ffp-contract=on -fno-slp-vectorize $flags -Itorch_nfft_amd/csrc -I/opt/rocm/include -x hip -c $src -o /tmp/exp_$name.o && \
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-soname,libnfft_hip.so -o scripts/ubench/libnfft_$name.so $OBJS /tmp/exp_$name.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib ) &
done
wait
ls scripts/ubench/*.so
