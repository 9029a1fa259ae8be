// micro-benchmark + spot check: producer/consumer split-bf16 GEMM (gemm_pc3.hip) against the shipped 4-wave kernel
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

int run(int M, int N, int K, int sk, const char *name, bool check) {
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    const bool zeros = getenv("GEMM_ZEROS") != nullptr;     // zero operands: same instruction stream, far fewer bit toggles
    for (auto &v : A) v = zeros ? 0.f : nd(rng) + 0.5f;
    for (auto &v : B) v = zeros ? 0.f : nd(rng) * 0.05f;
    float *dA, *dB, *dC;
    const int nslab = sk;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)nslab * M * N * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    GemmArgs g;
    g.A0 = dA; g.lda = K; g.B0 = dB; g.ldb = K; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K; g.splitK = sk; g.sCsplit = (long)M * N;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned short *dA3, *dB3;
    CK(hipMalloc(&dA3, A.size() * 6)); CK(hipMalloc(&dB3, B.size() * 6));
    if (launch_split3(st, dA, dA3, (long)A.size(), (long)A.size()) || launch_split3(st, dB, dB3, (long)B.size(), (long)B.size())) { printf("split failed\n"); return 1; }
    CK(hipStreamSynchronize(st));
    // two-piece fp16 operands: B split once with its power-of-two scale, A's maximum in the 64 slots its producer would fill
    unsigned short *dB16; unsigned *dmax;
    CK(hipMalloc(&dB16, B.size() * 4)); CK(hipMalloc(&dmax, (size_t)M * 4));
    float amaxB = 0.f;
    for (auto v : B) amaxB = std::max(amaxB, std::fabs(v));
    const float sB = gemm_f16x2_scale(amaxB);
    if (launch_split2h(st, dB, dB16, (long)B.size(), (long)B.size(), sB)) { printf("split2h failed\n"); return 1; }
    {
        std::vector<unsigned> rows((size_t)M, 0u);            // max |A[m][:]| per row, as bit patterns
        for (int m = 0; m < M; ++m) {
            float am = 0.f;
            for (int k = 0; k < K; ++k) am = std::max(am, std::fabs(A[(size_t)m * K + k]));
            memcpy(&rows[m], &am, 4);
        }
        CK(hipMemcpy(dmax, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipStreamSynchronize(st));
    // all-consumer experiment kernel: A pre-split too, one scale for the whole operand
    unsigned short *dA16;
    CK(hipMalloc(&dA16, A.size() * 4));
    if (launch_split_rows2h(st, dA, dmax, dA16, M, K, (long)A.size())) { printf("split_rows2h failed\n"); return 1; }
    CK(hipStreamSynchronize(st));
    const char *names[6] = {"shipped", "prod/cons", "pc B-pre", "pc AB-pre", "pc f16x2", "cc f16x2"};
    for (int which = 0; which < 6; ++which) {
        if (which == 5 && M % 64) continue;
        if (which == 0 && N % 128) continue;
        GemmArgs gg = g;
        if (which == 2 || which == 3) { gg.B3 = dB3; gg.pB3 = (long)B.size(); }
        if (which == 3) { gg.A3 = dA3; gg.pA3 = (long)A.size(); }
        if (which >= 4) { gg.B16 = dB16; gg.pB16 = (long)B.size(); gg.sB16 = sB; gg.amax = dmax; }
        if (which == 5) { gg.A3 = dA16; gg.pA3 = (long)A.size(); }
        auto launch = [&]() { return which == 5 ? launch_gemm_nt_f16x2_cc(st, gg) : which == 4 ? launch_gemm_nt_f16x2_pc(st, gg) : which ? launch_gemm_nt_bf16x3_pc(st, gg) : launch_gemm_nt_bf16x3(st, gg); };
        CK(hipMemset(dC, 0xFF, (size_t)nslab * M * N * 4));
        int rc = launch();
        if (rc) { printf("launch rc %d\n", rc); return 1; }
        CK(hipStreamSynchronize(st));
        double num = 0, den = 0;
        if (check) {
            std::vector<float> C((size_t)nslab * M * N);
            CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
            for (int t = 0; t < 3000; ++t) {
                const int m = (int)(((long)t * 7919 + 13) % M), n = (t < 64) ? N - 1 - t % std::min(N, 64) : (int)(((long)t * 104729 + 7) % N);
                double ref = 0, got = 0;
                for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
                for (int s = 0; s < (which ? nslab : sk); ++s) got += C[(size_t)s * M * N + (size_t)m * N + n];
                num += (got - ref) * (got - ref); den += ref * ref;
            }
        }
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, st));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-8s %-9s M=%d N=%d K=%d sk=%d  rel L2 err %.3g   %.4f ms  %.1f TF/s fp32-equivalent\n", name, names[which], M, N, K, sk,
               check ? std::sqrt(num / den) : -1.0, ms, 2.0 * M * N * K / ms * 1e-9);
    }
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dA3); hipFree(dB3); hipFree(dB16); hipFree(dmax); hipFree(dA16);
    return 0;
}

int main() {
#ifdef PC_EXP
    printf("experiment %d\n", PC_EXP);
    if (run(1664, 1408, 16896, 3, "forward", false)) return 1;
    return 0;
#endif
    if (run(128, 256, 64, 1, "tiny", true)) return 1;
    if (run(128, 128, 96, 1, "half", true)) return 1;
    if (run(256, 384, 512, 2, "ragged", true)) return 1;
    // config 3, band 2C: forward  yT[NP][LdetP] = Xs[NP][K] W[LdetP][K]^T ; adjoint  XsT[NP][K] = ymat[NP][LdetP] Wt[K][LdetP]^T
    if (run(1664, 1408, 16896, 6, "forward", true)) return 1;
    if (run(1664, 1408, 16896, 3, "forward", false)) return 1;
    if (run(1664, 1408, 16896, 8, "forward", false)) return 1;
    if (run(1664, 1408, 16896, 16, "forward", false)) return 1;
    if (run(1664, 16896, 1408, 1, "adjoint", true)) return 1;
    return 0;
}
