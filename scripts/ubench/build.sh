#!/bin/bash
# Builds the microbenchmarks next to their sources (they travel to the GPU box with gpurun; results: profiles/).
set -e
cd "$(dirname "$0")"
for f in *.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=on -fno-slp-vectorize -o "${f%.hip}" "$f"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -o fft_prune fft_prune.cpp -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
