// micro-benchmark of the wave-specialised DFT pass (surfh_amd/csrc/dft_ws.hip) against the one-role kernel (dft_rx3.hip):
// the four passes of config 3 (251 x 251 x 4096 planes) on random data.  Build variants with -DWS_EXP=mask (see dft_ws.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dft_ws.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void fill_k(float *p, long n, unsigned seed) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(x & 0xFFFF) / 65536.f - 0.5f;
    }
}
int main(int argc, char **argv) {
    const int Na = 251, Nb = 251, ha = 126, hb = 126, NAP = 256, NBP = 256, KBP = 128;
    const long LP = argc > 1 ? atol(argv[1]) : 4096;
    const int MP = 128, KP = 128;
    float *cube, *ycol, *spec, *mhat, *tpl, *mixtab, *tplT;
    unsigned short *A;
    const size_t ncube = (size_t)NBP * NAP * LP, nsp = (size_t)2 * NAP * KBP * LP;
    CK(hipMalloc(&cube, ncube * 4)); CK(hipMalloc(&ycol, nsp * 4)); CK(hipMalloc(&spec, nsp * 4));
    CK(hipMalloc(&A, (size_t)6 * MP * KP * 2)); CK(hipMalloc(&mhat, (size_t)4 * 2 * NAP * KBP * 4)); CK(hipMalloc(&tpl, (size_t)4 * LP * 4));
    const int mix_rows = dft_ws_mix_rows(Na, KP);
    CK(hipMalloc(&mixtab, (size_t)hb * 2 * mix_rows * 16)); CK(hipMalloc(&tplT, (size_t)LP * 16));
    hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, cube, (long)ncube, 1u);
    hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, ycol, (long)nsp, 2u);
    hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, spec, (long)nsp, 3u);
    hipLaunchKernelGGL(fill_k, dim3(64), dim3(256), 0, 0, (float *)A, (long)3 * MP * KP, 4u);
    hipLaunchKernelGGL(fill_k, dim3(64), dim3(256), 0, 0, mhat, (long)4 * 2 * NAP * KBP, 5u);
    hipLaunchKernelGGL(fill_k, dim3(64), dim3(256), 0, 0, tpl, (long)4 * LP, 6u);
    hipLaunchKernelGGL(fill_k, dim3(64), dim3(256), 0, 0, tplT, (long)4 * LP, 7u);
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool only_ws = getenv("ONLY_WS") != nullptr;
    for (int pass = 0; pass < 5; ++pass) {
        DftWsArgs g;
        g.A[0] = A; g.A[1] = A + 3 * MP * KP; g.planeA = (long)MP * KP; g.lda = KP; g.MP = MP; g.KP = KP;
        const char *name = "";
        if (pass == 0) {          // r2c along beta
            name = "rows_fwd (r2c)"; g.src[0] = cube; g.src[1] = cube; g.ldb = NAP * LP; g.fold[0] = 1; g.fold[1] = -1; g.Kn = Nb;
            g.dst[0] = ycol; g.dst[1] = ycol + (long)KBP * NAP * LP; g.ldc = NAP * LP; g.mode = 1; g.e11 = -1; g.rvalid = hb; g.N = (int)(Na * LP);
        } else if (pass == 1) {   // c2c along alpha batched over kb
            name = "cols_fwd (c2c)"; g.src[0] = ycol; g.src[1] = ycol + (long)KBP * NAP * LP; g.ldb = LP; g.sB = NAP * LP; g.fold[0] = 1; g.fold[1] = -1; g.Kn = Na;
            g.dst[0] = spec; g.ldc = KBP * LP; g.sC = LP; g.mode = 0; g.Rn = Na; g.rvalid = ha; g.e01 = 1; g.e10 = 1; g.e11 = -1; g.N = (int)LP; g.batch = hb;
            g.nvar = 2; g.A_alt[0] = g.A[1]; g.A_alt[1] = g.A[0]; g.fold_alt[0] = -1; g.fold_alt[1] = 1; g.dst_alt = spec + (long)NAP * KBP * LP;
            g.packed = 1;
        } else if (pass == 2 || pass == 4) {   // c2c along alpha, unbatched wide N (4: with the fused spectral mix)
            name = pass == 2 ? "cols_inv (c2c)" : "cols_inv_mix"; g.src[0] = spec; g.src[1] = spec + (long)NAP * KBP * LP; g.ldb = KBP * LP; g.fold[0] = 1; g.fold[1] = -1; g.Kn = Na;
            g.dst[0] = ycol; g.ldc = KBP * LP; g.mode = 0; g.Rn = Na; g.rvalid = ha; g.e01 = -1; g.e10 = 1; g.e11 = 1; g.N = (int)(hb * LP);
            g.nvar = 2; g.A_alt[0] = g.A[1]; g.A_alt[1] = g.A[0]; g.fold_alt[0] = -1; g.fold_alt[1] = 1; g.dst_alt = ycol + (long)NAP * KBP * LP;
            g.packed = 1;
            if (pass == 4) {
                g.mhat = mhat; g.tpl = tpl; g.T = 4; g.LP = (int)LP; g.PL = (long)NAP * KBP; g.KBP = KBP;
                g.mixtab = (const float4 *)mixtab; g.tplT = (const float4 *)tplT; g.mix_rows = mix_rows;
                launch_dft_ws_mix_table(st, mhat, mixtab, 4, Na, hb, g.PL, KBP, mix_rows);
            }
        } else {                  // c2r along beta batched over alpha
            name = "rows_inv (c2r)"; g.src[0] = ycol; g.src[1] = ycol + (long)NAP * KBP * LP; g.ldb = LP; g.sB = KBP * LP;
            g.dst[0] = cube; g.ldc = NAP * LP; g.sC = LP; g.mode = 0; g.e01 = -1; g.e10 = 1; g.e11 = 1; g.Rn = Nb; g.rvalid = hb; g.N = (int)LP; g.batch = Na;
        }
        for (int which = only_ws ? 1 : 0; which < 2; ++which) {
            auto launch = [&]() { return which ? launch_dft_ws(st, g) : launch_dft_rx3(st, g); };
            for (int i = 0; i < 3; ++i) { int rc = launch(); if (rc) { printf("launch rc %d\n", rc); return 1; } }
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            const int reps = 20;
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-16s %-4s %.4f ms\n", name, which ? "ws" : "rx3", ms / reps);
        }
    }
    return 0;
}
