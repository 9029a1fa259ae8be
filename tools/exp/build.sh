#!/bin/bash
# builds the four experiment variants of the rx3 micro-benchmark (0 full, 1 no MFMA, 2 no data loads, 3 no stores)
set -e
cd "$(dirname "$0")"
mkdir -p _bin
for e in 0 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRX_EXP=$e -o _bin/rx3_exp$e rx3_exp.hip rx3_main.hip &
done
wait
for e in 0 1 2 3 4; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGX_EXP=$e -o _bin/gemm_exp$e gemm_exp.hip gemm_main.hip &
done
wait
