// micro-benchmark + spot check of the all-consumer two-piece fp16 GEMM (surfh_amd/csrc/gemm_cc16.hip) at the shapes of
// config 3, band 2C.  Build with -DCC_EXP=mask to remove one cost of the kernel at a time (1 no DMA inside the loop,
// 2 no fragment reads, 4 no MFMAs; results are then wrong by construction); CC_ZEROS=1 | 2 runs it on all-zero / constant
// operands (same instruction stream, no bit toggling: the clock the chip holds is the difference).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <vector>
#include "../../surfh_amd/csrc/gemm_f32.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static int run(int M, int N, int K, int sk, const char *name) {
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    const char *zm = getenv("CC_ZEROS");                 // 1: all-zero operands, 2: constant operands (no bit toggling)
    const int zmode = zm ? atoi(zm) : 0;
    for (auto &v : A) v = zmode == 1 ? 0.f : zmode == 2 ? 1.0f : nd(rng) + 0.5f;
    for (auto &v : B) v = zmode == 1 ? 0.f : zmode == 2 ? 0.0625f : nd(rng) * 0.05f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)sk * M * N * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned short *dB16, *dA16; unsigned *dmax;
    CK(hipMalloc(&dB16, B.size() * 4)); CK(hipMalloc(&dA16, A.size() * 4)); CK(hipMalloc(&dmax, (size_t)M * 4));
    float amaxB = 0.f;
    for (auto v : B) amaxB = std::max(amaxB, std::fabs(v));
    const float sB = gemm_f16x2_scale(amaxB);
    if (launch_split2h(st, dB, dB16, (long)B.size(), (long)B.size(), sB)) { printf("split2h failed\n"); return 1; }
    std::vector<unsigned> rows((size_t)M, 0u);
    for (int m = 0; m < M; ++m) {
        float am = 0.f;
        for (int k = 0; k < K; ++k) am = std::max(am, std::fabs(A[(size_t)m * K + k]));
        memcpy(&rows[m], &am, 4);
    }
    CK(hipMemcpy(dmax, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    if (launch_split_rows2h(st, dA, dmax, dA16, M, K, (long)A.size())) { printf("split_rows2h failed\n"); return 1; }
    CK(hipStreamSynchronize(st));
    GemmArgs g;
    g.lda = K; g.ldb = K; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K; g.splitK = sk; g.sCsplit = (long)M * N;
    g.B16 = dB16; g.pB16 = (long)B.size(); g.sB16 = sB; g.amax = dmax; g.A3 = dA16; g.pA3 = (long)A.size();
    // CC_FARFRAC=f: K-step lists with the last fraction f of every tile's steps in the far class (one product): timing of
    // mixed passes; the error then shows what single fp16 products cost on operands without structure
    int *dkl = nullptr;
    if (const char *ff = getenv("CC_FARFRAC")) {
        const int nb = K / 32, tilesN = (N + 255) / 256, stride = 2 + nb, nf = (int)(atof(ff) * nb);
        std::vector<int> kl((size_t)tilesN * stride);
        for (int t = 0; t < tilesN; ++t) {
            kl[(size_t)t * stride] = nb - nf; kl[(size_t)t * stride + 1] = nf;
            for (int b = 0; b < nb; ++b) kl[(size_t)t * stride + 2 + b] = b;
        }
        CK(hipMalloc(&dkl, kl.size() * 4));
        CK(hipMemcpy(dkl, kl.data(), kl.size() * 4, hipMemcpyHostToDevice));
        g.klist = dkl; g.klistStride = stride;
    }
    CK(hipMemset(dC, 0xFF, (size_t)sk * M * N * 4));
    int rc = launch_gemm_nt_f16x2_cc(st, g);
    if (rc) { printf("launch rc %d\n", rc); return 1; }
    CK(hipStreamSynchronize(st));
    std::vector<float> C((size_t)sk * M * N);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0, worst = 0;
    for (int t = 0; t < 4000; ++t) {
        const int m = (t < 300) ? M - 1 - t % std::min(M, 300) : (int)(((long)t * 7919 + 13) % M);
        const int n = (t & 1) ? N - 1 - (t / 2) % std::min(N, 300) : (int)(((long)t * 104729 + 7) % N);
        double ref = 0, got = 0;
        for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
        for (int s = 0; s < sk; ++s) got += C[(size_t)s * M * N + (size_t)m * N + n];
        num += (got - ref) * (got - ref); den += ref * ref;
        worst = std::max(worst, std::fabs(got - ref));
    }
    for (int i = 0; i < 3; ++i) launch_gemm_nt_f16x2_cc(st, g);
    CK(hipEventRecord(e0, st));
    const int reps = 30;
    for (int i = 0; i < reps; ++i) launch_gemm_nt_f16x2_cc(st, g);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-8s M=%d N=%d K=%d sk=%d  rel L2 err %.3g (worst abs %.3g)  %.4f ms  %.1f TF/s fp32-equivalent, %.0f TF/s fp16 issued\n", name, M, N, K, sk,
           std::sqrt(num / den), worst, ms, 2.0 * M * N * K / ms * 1e-9, 6.0 * M * N * K / ms * 1e-9);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dB16); hipFree(dmax); hipFree(dA16); hipFree(dkl);
    return 0;
}

int main() {
#ifdef CC_EXP
    printf("experiment mask %d\n", CC_EXP);
#endif
    if (run(256, 384, 512, 2, "ragged")) return 1;
    if (run(320, 256, 96, 1, "short")) return 1;
    if (run(1664, 1408, 16896, 6, "forward")) return 1;
    if (run(1664, 1408, 16896, 8, "forward")) return 1;
    if (run(1664, 16896, 1408, 1, "adjoint")) return 1;
    if (run(4096, 4096, 8192, 1, "square")) return 1;
    return 0;
}
