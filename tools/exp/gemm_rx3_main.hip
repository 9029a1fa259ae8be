// micro-benchmark + spot check of the register-direct split-bf16 GEMM at the two spectral-blur shapes of config 3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "gemm_rx3.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static void split_planes(const std::vector<float> &M, std::vector<unsigned short> &o) {
    o.resize(3 * M.size());
    for (size_t j = 0; j < M.size(); ++j) {
        float x = M[j], hh, mm; uint32_t u;
        memcpy(&u, &x, 4); u &= 0xFFFF0000u; memcpy(&hh, &u, 4); o[j] = (unsigned short)(u >> 16);
        float r = x - hh;
        memcpy(&u, &r, 4); u &= 0xFFFF0000u; memcpy(&mm, &u, 4); o[M.size() + j] = (unsigned short)(u >> 16);
        r -= mm; memcpy(&u, &r, 4); u += 0x7FFFu + ((u >> 16) & 1u); o[2 * M.size() + j] = (unsigned short)(u >> 16);
    }
}

int run(int M, int N, int K, int sk, const char *name) {
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    for (auto &v : A) v = nd(rng) * 0.05f;
    for (auto &v : B) v = nd(rng) + 0.5f;
    std::vector<unsigned short> A3;
    split_planes(A, A3);
    unsigned short *dA; float *dB, *dC;
    CK(hipMalloc(&dA, A3.size() * 2)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)sk * M * N * 4));
    CK(hipMemcpy(dA, A3.data(), A3.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xFF, (size_t)sk * M * N * 4));
    GemmRx3Args g;
    g.A3 = dA; g.planeA = (long)M * K; g.lda = K; g.B = dB; g.ldb = K; g.Ct = dC; g.ldct = M; g.M = M; g.N = N; g.K = K;
    g.splitK = sk; g.sCsplit = (long)M * N;
    hipStream_t st; CK(hipStreamCreate(&st));
    int rc = launch_gemm_rx3(st, g);
    if (rc) { printf("launch rc %d\n", rc); return 1; }
    CK(hipStreamSynchronize(st));
    std::vector<float> C((size_t)sk * M * N);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, num = 0, den = 0;
    for (int t = 0; t < 4000; ++t) {
        const int m = (int)(((long)t * 7919 + 13) % M), n = (int)(((long)t * 104729 + 7) % N);
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
        double got = 0;
        for (int s = 0; s < sk; ++s) got += C[(size_t)s * M * N + (size_t)n * M + m];
        num += (got - ref) * (got - ref); den += ref * ref;
        worst = std::max(worst, std::fabs(got - ref));
    }
    // edges: last rows / cols
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_gemm_rx3(st, g);
    CK(hipEventRecord(e0, st));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch_gemm_rx3(st, g);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-8s M=%d N=%d K=%d sk=%d  rel L2 err %.3g  max abs %.3g   %.4f ms  %.1f TF/s fp32-equivalent (random data)\n", name, M, N, K, sk,
           std::sqrt(num / den), worst, ms, 2.0 * M * N * K / ms * 1e-9);
    hipFree(dA); hipFree(dB); hipFree(dC);
    return 0;
}

int main() {
#ifdef GX_EXP
    printf("experiment %d\n", GX_EXP);
#else
    if (run(128, 128, 64, 1, "tiny")) return 1;
    if (run(256, 384, 256, 2, "ragged")) return 1;
#endif
    if (run(1408, 2432, 8192, 8, "forward")) return 1;
    if (run(1408, 2432, 8192, 4, "forward")) return 1;
    if (run(8192, 2432, 1408, 1, "adjoint")) return 1;
    return 0;
}
