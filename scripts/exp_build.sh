#!/bin/bash
# Developer tool: build variant libraries scripts/ubench/libnfft_<name>.so from an experimental copy of one kernel file.
# usage: scripts/exp_build.sh <experimental.hip> <file in torch_nfft_amd/csrc it replaces> name1:"-DX=1" ...
set -e
SRC=$1; shift; REPL=$1; shift
BASE=$REPL
OBJS=""
for f in api.hip binning.hip spread.hip interp.hip spectral.hip fft.cpp; do
  if [ "$f" != "$BASE" ]; then OBJS="$OBJS torch_nfft_amd/_obj/$f.o"; fi
done
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -ffp-contract=fast $flags -Itorch_nfft_amd/csrc -I/opt/rocm/include -x hip -c $SRC -o /tmp/exp_$name.o && \
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/ubench/libnfft_$name.so $OBJS /tmp/exp_$name.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib ) &
done
wait
ls scripts/ubench/*.so
