#!/bin/bash
# builds the micro-benchmarks of tools/exp into _bin/ (git-ignored, travels to the GPU box)
set -e
cd "$(dirname "$0")"
mkdir -p _bin
C="hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value"
$C -o _bin/rx3 dft_rx3.hip rx3_main.hip &
$C -o _bin/h2 ../../surfh_amd/csrc/dft_h2.hip dft_rx3.hip h2_main.hip &      # all four passes + the fused adjoint tail
$C -o _bin/cc_bench ../../surfh_amd/csrc/gemm_cc16.hip cc_main.hip &
for e in 1 2 3 4 6; do      # the GEMM with one cost removed at a time (cc_main.hip)
  $C -DCC_EXP=$e -o _bin/cc_exp$e ../../surfh_amd/csrc/gemm_cc16.hip cc_main.hip &
done
$C -o _bin/copy_bw copy_bw.hip &
$C -o _bin/stride_bw stride_bw.hip &
$C -o _bin/mall_copy mall_copy.hip &
$C -DWPS=1 -DDEPTH=3 -o _bin/rx3v2 rx3v2.hip &
$C -o _bin/ct ../../surfh_amd/csrc/dft_ct.hip dft_dif.hip ct_main.hip &   # the Cooley-Tukey pass (and its DIF variant): check against float64 + timing
for e in 2 4 6 32 64 256 512 1024; do   # ... with one cost removed at a time (CT_EXP in dft_ct.hip)
  $C -DCT_EXP=$e -o _bin/ct_e$e ../../surfh_amd/csrc/dft_ct.hip dft_dif.hip ct_main.hip &
done
wait
