// micro-benchmark of the register-direct split-bf16 DFT pass: shapes of config 3's four passes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dft_rx3.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const int Na = 251, Nb = 251, ha = 126, hb = 126, NAP = 256, NBP = 256, KBP = 128;
    const long LP = 4096;
    const int MP = 128, KP = 128;
    float *cube, *ycol, *spec;
    unsigned short *A;
    CK(hipMalloc(&cube, (size_t)NBP * NAP * LP * 4));
    CK(hipMalloc(&ycol, (size_t)2 * NAP * KBP * LP * 4));
    CK(hipMalloc(&spec, (size_t)2 * NAP * KBP * LP * 4));
    CK(hipMalloc(&A, (size_t)6 * MP * KP * 2));
    CK(hipMemset(cube, 0, (size_t)NBP * NAP * LP * 4));
    CK(hipMemset(ycol, 0, (size_t)2 * NAP * KBP * LP * 4));
    CK(hipMemset(spec, 0, (size_t)2 * NAP * KBP * LP * 4));
    CK(hipMemset(A, 0, (size_t)6 * MP * KP * 2));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pass = 0; pass < 4; ++pass) {
        DftRx3Args g;
        g.A[0] = A; g.A[1] = A + 3 * MP * KP; g.planeA = (long)MP * KP; g.lda = KP; g.MP = MP; g.KP = KP;
        const char *name = "";
        if (pass == 0) {          // r2c along beta
            name = "rows_fwd"; g.src[0] = cube; g.src[1] = cube; g.ldb = NAP * LP; g.fold[0] = 1; g.fold[1] = -1; g.Kn = Nb;
            g.dst[0] = ycol; g.dst[1] = ycol + (long)KBP * NAP * LP; g.ldc = NAP * LP; g.mode = 1; g.e11 = -1; g.rvalid = hb; g.N = (int)(Na * LP);
        } else if (pass == 1) {   // c2c along alpha batched over kb
            name = "cols_fwd(2v)"; g.src[0] = ycol; g.src[1] = ycol + (long)KBP * NAP * LP; g.ldb = LP; g.sB = NAP * LP; g.fold[0] = 1; g.fold[1] = -1; g.Kn = Na;
            g.dst[0] = spec; g.ldc = KBP * LP; g.sC = LP; g.mode = 0; g.Rn = Na; g.rvalid = ha; g.e01 = 1; g.e10 = 1; g.e11 = -1; g.N = (int)LP; g.batch = hb;
            g.nvar = 2; g.A_alt[0] = g.A[1]; g.A_alt[1] = g.A[0]; g.fold_alt[0] = -1; g.fold_alt[1] = 1; g.dst_alt = spec + (long)NAP * KBP * LP;
            g.packed = getenv("RX3_PACKED") ? 1 : 0;
        } else if (pass == 2) {   // c2c along alpha, unbatched wide N
            name = "cols_inv(2v)"; g.src[0] = spec; g.src[1] = spec + (long)NAP * KBP * LP; g.ldb = KBP * LP; g.fold[0] = 1; g.fold[1] = -1; g.Kn = Na;
            g.dst[0] = ycol; g.ldc = KBP * LP; g.mode = 0; g.Rn = Na; g.rvalid = ha; g.e01 = -1; g.e10 = 1; g.e11 = 1; g.N = (int)(hb * LP);
            g.nvar = 2; g.A_alt[0] = g.A[1]; g.A_alt[1] = g.A[0]; g.fold_alt[0] = -1; g.fold_alt[1] = 1; g.dst_alt = ycol + (long)NAP * KBP * LP;
            g.packed = getenv("RX3_PACKED") ? 1 : 0;
        } else {                  // c2r along beta batched over alpha
            name = "rows_inv"; g.src[0] = ycol; g.src[1] = ycol + (long)NAP * KBP * LP; g.ldb = LP; g.sB = KBP * LP;
            g.dst[0] = cube; g.ldc = NAP * LP; g.sC = LP; g.mode = 0; g.e01 = -1; g.e10 = 1; g.e11 = 1; g.Rn = Nb; g.rvalid = hb; g.N = (int)LP; g.batch = Na;
        }
        for (int i = 0; i < 3; ++i) { int rc = launch_dft_rx3(st, g); if (rc) { printf("launch rc %d\n", rc); return 1; } }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch_dft_rx3(st, g);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-12s %.4f ms\n", name, ms / reps);
    }
    // ---- the same two transforms cut into wavelength chunks: the intermediate (half-transformed) array of a chunk is
    // a small buffer that is rewritten chunk after chunk and should stay in the 256 MB Infinity Cache
    for (int Lch : {128, 256, 512, 1024}) {
        float *ys;
        CK(hipMalloc(&ys, (size_t)2 * KBP * NAP * Lch * 4));
        CK(hipMemset(ys, 0, (size_t)2 * KBP * NAP * Lch * 4));
        auto fwd = [&](long l0) {
            DftRx3Args g;   // r2c along beta, batched over alpha
            g.A[0] = A; g.A[1] = A + 3 * MP * KP; g.planeA = (long)MP * KP; g.lda = KP; g.MP = MP; g.KP = KP;
            g.src[0] = cube + l0; g.src[1] = cube + l0; g.ldb = NAP * LP; g.sB = LP; g.batch = Na; g.fold[0] = 1; g.fold[1] = -1; g.Kn = Nb;
            g.dst[0] = ys; g.dst[1] = ys + (long)KBP * NAP * Lch; g.ldc = (long)NAP * Lch; g.sC = Lch; g.mode = 1; g.e11 = -1; g.rvalid = hb; g.N = Lch;
            launch_dft_rx3(st, g);
            DftRx3Args h;   // c2c along alpha, batched over kb
            h.A[0] = A; h.A[1] = A + 3 * MP * KP; h.planeA = (long)MP * KP; h.lda = KP; h.MP = MP; h.KP = KP;
            h.src[0] = ys; h.src[1] = ys + (long)KBP * NAP * Lch; h.ldb = Lch; h.sB = (long)NAP * Lch; h.fold[0] = 1; h.fold[1] = -1; h.Kn = Na;
            h.dst[0] = spec + l0; h.ldc = KBP * LP; h.sC = LP; h.mode = 0; h.Rn = Na; h.rvalid = ha; h.e01 = 1; h.e10 = 1; h.e11 = -1; h.N = Lch; h.batch = hb;
            h.nvar = 2; h.A_alt[0] = h.A[1]; h.A_alt[1] = h.A[0]; h.fold_alt[0] = -1; h.fold_alt[1] = 1; h.dst_alt = spec + (long)NAP * KBP * LP + l0;
            launch_dft_rx3(st, h);
        };
        auto inv = [&](long l0) {
            DftRx3Args g;   // c2c along alpha: spectrum chunk -> small intermediate [2][NAP][KBP][Lch]
            g.A[0] = A; g.A[1] = A + 3 * MP * KP; g.planeA = (long)MP * KP; g.lda = KP; g.MP = MP; g.KP = KP;
            g.src[0] = spec + l0; g.src[1] = spec + (long)NAP * KBP * LP + l0; g.ldb = KBP * LP; g.sB = LP; g.batch = hb; g.fold[0] = 1; g.fold[1] = -1; g.Kn = Na;
            g.dst[0] = ys; g.ldc = (long)KBP * Lch; g.sC = Lch; g.mode = 0; g.Rn = Na; g.rvalid = ha; g.e01 = -1; g.e10 = 1; g.e11 = 1; g.N = Lch;
            g.nvar = 2; g.A_alt[0] = g.A[1]; g.A_alt[1] = g.A[0]; g.fold_alt[0] = -1; g.fold_alt[1] = 1; g.dst_alt = ys + (long)NAP * KBP * Lch;
            launch_dft_rx3(st, g);
            DftRx3Args h;   // c2r along beta batched over alpha
            h.A[0] = A; h.A[1] = A + 3 * MP * KP; h.planeA = (long)MP * KP; h.lda = KP; h.MP = MP; h.KP = KP;
            h.src[0] = ys; h.src[1] = ys + (long)NAP * KBP * Lch; h.ldb = Lch; h.sB = (long)KBP * Lch;
            h.dst[0] = cube + l0; h.ldc = NAP * LP; h.sC = LP; h.mode = 0; h.e01 = -1; h.e10 = 1; h.e11 = 1; h.Rn = Nb; h.rvalid = hb; h.N = Lch; h.batch = Na;
            launch_dft_rx3(st, h);
        };
        for (int dir = 0; dir < 2; ++dir) {
            for (long l0 = 0; l0 < LP; l0 += Lch) { if (dir == 0) fwd(l0); else inv(l0); }
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            const int reps = 10;
            for (int i = 0; i < reps; ++i)
                for (long l0 = 0; l0 < LP; l0 += Lch) { if (dir == 0) fwd(l0); else inv(l0); }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("chunked %s  Lch=%4d  %.4f ms per 2-D transform (intermediate %.0f MB)\n", dir == 0 ? "rfft2 " : "irfft2", Lch, ms / reps,
                   2.0 * KBP * NAP * Lch * 4 / 1e6);
        }
        CK(hipFree(ys));
    }
    return 0;
}
