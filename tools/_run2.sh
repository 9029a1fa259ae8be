cd tools/exp/_bin
(echo "== 32x32x16"; ./cc_bench; echo "== 16x16x32"; SURFH_GEMM_MFMA16=1 ./cc_bench; echo "== 32x32x16 again"; ./cc_bench | grep "forward\|adjoint"; echo "== 16x16x32 zeros"; CC_ZEROS=1 SURFH_GEMM_MFMA16=1 ./cc_bench | grep "forward\|adjoint") > ../../../gpurun_out/cc_bench6.log 2>&1
