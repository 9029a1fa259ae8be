// signed error of the split-bf16 GEMM on non-negative operands as a function of the accumulation chain length (split K)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../../surfh_amd/csrc/gemm_f32.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const int M = 256, N = 256, K = 12288;
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    for (auto &v : A) v = ud(rng);
    for (auto &v : B) v = ud(rng);
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)32 * M * N * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    std::vector<double> ref((size_t)64 * 64);
    for (int m = 0; m < 64; ++m) for (int n = 0; n < 64; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[(size_t)(m * 4) * K + k] * B[(size_t)(n * 4) * K + k]; ref[m * 64 + n] = s; }
    for (int which = 0; which < 3; ++which)
        for (int sk : {1, 2, 4, 8, 12, 24}) {
            GemmArgs g; g.A0 = dA; g.lda = K; g.B0 = dB; g.ldb = K; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K; g.splitK = sk; g.sCsplit = (long)M * N;
            int rc = which == 0 ? launch_gemm_nt_bf16x3(0, g) : which == 1 ? launch_gemm_nt_bf16x3_pc(0, g) : 0;
            if (which == 2) { std::vector<float> bt((size_t)K * N); for (int k = 0; k < K; ++k) for (int n = 0; n < N; ++n) bt[(size_t)k * N + n] = B[(size_t)n * K + k];
                float *dBt; CK(hipMalloc(&dBt, bt.size() * 4)); CK(hipMemcpy(dBt, bt.data(), bt.size() * 4, hipMemcpyHostToDevice)); g.B0 = dBt; g.ldb = N; rc = launch_gemm_f32(0, g); CK(hipDeviceSynchronize()); hipFree(dBt); }
            if (rc) { printf("rc %d\n", rc); continue; }
            CK(hipDeviceSynchronize());
            std::vector<float> C((size_t)sk * M * N);
            CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
            double bias = 0, rms = 0;
            for (int m = 0; m < 64; ++m) for (int n = 0; n < 64; ++n) {
                float got = 0; for (int s = 0; s < sk; ++s) got += C[(size_t)s * M * N + (size_t)(m * 4) * N + n * 4];
                const double e = (got - ref[m * 64 + n]) / ref[m * 64 + n]; bias += e; rms += e * e;
            }
            printf("%-10s K=%d splitK=%2d (chain %5d k)  mean signed rel err %+.3e   rms %.3e\n", which == 0 ? "4-wave" : which == 1 ? "prod/cons" : "fp32 mfma", K, sk, K / sk, bias / 4096, std::sqrt(rms / 4096));
        }
    return 0;
}
