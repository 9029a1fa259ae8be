// micro-benchmark of the split-bf16 NT GEMM at the spectral-blur shape of config 3
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../surfh_amd/csrc/gemm_f32.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const int M = 1408, K = 1408, N = 11776;
    float *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMemset(A, 0, (size_t)M * K * 4)); CK(hipMemset(B, 0, (size_t)N * K * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    GemmArgs g;
    g.A0 = A; g.lda = K; g.B0 = B; g.ldb = K; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K;
    for (int i = 0; i < 3; ++i) { int rc = launch_gemm_nt_bf16x3(st, g); if (rc) { printf("rc %d\n", rc); return 1; } }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch_gemm_nt_bf16x3(st, g);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("gemm exp %d  %.4f ms  %.1f TF/s fp32-equivalent\n", GX_EXP, ms, 2.0 * M * N * K / ms * 1e-9);
    return 0;
}
