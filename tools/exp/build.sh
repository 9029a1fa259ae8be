#!/bin/bash
# builds the four experiment variants of the rx3 micro-benchmark (0 full, 1 no MFMA, 2 no data loads, 3 no stores)
set -e
cd "$(dirname "$0")"
mkdir -p _bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o _bin/rx3 ../../surfh_amd/csrc/dft_rx3.hip rx3_main.hip &
for e in 0 1 2 3 4; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGX_EXP=$e -o _bin/gemm_exp$e gemm_exp.hip gemm_main.hip &
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o _bin/gemm_rx3 gemm_rx3.hip gemm_rx3_main.hip &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o _bin/gemm_pc ../../surfh_amd/csrc/gemm_pc3.hip ../../surfh_amd/csrc/gemm_pc16.hip ../../surfh_amd/csrc/gemm_cc16.hip ../../surfh_amd/csrc/gemm_bf16x3.hip gemm_pc_main.hip &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o _bin/gemm_bias ../../surfh_amd/csrc/gemm_pc3.hip ../../surfh_amd/csrc/gemm_bf16x3.hip ../../surfh_amd/csrc/gemm_f32.hip gemm_bias_main.hip &
hipcc --offload-arch=gfx950 -O3 -o _bin/copy_bw copy_bw.hip &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWPS=1 -DDEPTH=3 -o _bin/rx3v2 rx3v2.hip &
wait
