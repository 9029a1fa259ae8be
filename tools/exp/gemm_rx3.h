// experiment (not part of the product): register-direct split-bf16 GEMM, see gemm_rx3.hip and README.md here
#pragma once
#include <hip/hip_runtime.h>

// Ct[n][m] = sum_k A[m][k] * B[n][k]: A handed over as three bf16 planes (exact split x = h + m + l, planes
// planeA elements apart, each [M][lda]); B fp32 [N][ldb]; the result is stored transposed, slab s of a split K at
// Ct + s*sCsplit.  M multiple of 128, N of 128, K of 32*splitK (gemm_rx3.hip).
struct GemmRx3Args {
    const unsigned short *A3 = nullptr;
    long planeA = 0;
    int lda = 0;
    const float *B = nullptr;
    long ldb = 0;
    float *Ct = nullptr;
    long ldct = 0;
    int M = 0, N = 0, K = 0;
    int splitK = 1;
    long sCsplit = 0;
};
int launch_gemm_rx3(hipStream_t stream, const GemmRx3Args &g);
