// micro-benchmark + check of the two-piece fp16 DFT pass (surfh_amd/csrc/dft_h2.hip) against the split-bf16 kernel
// (dft_rx3.hip): the four passes of config 3 (251 x 251 x LP planes) with the real cos / sin matrices.
//   h2_main [LP] [mode]    mode 0: uniform random data; 1: random exponents (2^0 .. 2^-40 per element);
//                          2: magnitude growing along the transformed axis (forces accumulator rescales)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "dft_rx3.h"
#include "../../surfh_amd/csrc/dft_h2.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void fill_k(float *p, long n, unsigned seed, int mode, long pitch) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        float v = (float)(x & 0xFFFF) / 65536.f - 0.5f;
        if (mode == 1) v = ldexpf(v, -(int)((x >> 16) % 41));
        if (mode == 2) { const long row = (i / pitch) % 256; const long r = row < 126 ? row : (251 - row > 0 ? 251 - row : 0); v = ldexpf(v, (int)(r / 4)); }
        p[i] = v;
    }
}
__global__ void ilv_k(const float *re, const float *im, float *out, long n) {   // out[2 i + c] = plane c [i]
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { out[2 * i] = re[i]; out[2 * i + 1] = im[i]; }
}
__global__ void dilv_k(const float *in, float *re, float *im, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { re[i] = in[2 * i]; im[i] = in[2 * i + 1]; }
}
// reference of the fused adjoint tail: madj[t][c][ka][kb] = sum_l tpl[t][l] (conj(H) Y)[c], float64 accumulation
__global__ void adjmix_ref_k(const float *Y, const float *H, const float *tpl, float *madj, int T, long LP, long KBP, long PL) {
    const long k = blockIdx.x;     // ka * KBP + kb
    double ar[4] = {0, 0, 0, 0}, ai[4] = {0, 0, 0, 0};
    for (long l = threadIdx.x; l < LP; l += 256) {
        const double yr = Y[(k * LP + l) * 2], yi = Y[(k * LP + l) * 2 + 1], hr = H[(k * LP + l) * 2], hi = H[(k * LP + l) * 2 + 1];
        const double zr = hr * yr + hi * yi, zi = hr * yi - hi * yr;
        for (int t = 0; t < T; ++t) { ar[t] += tpl[t * LP + l] * zr; ai[t] += tpl[t * LP + l] * zi; }
    }
    __shared__ double red[256];
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < 2; ++c) {
            red[threadIdx.x] = c ? ai[t] : ar[t];
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
            if (threadIdx.x == 0) madj[((long)t * 2 + c) * PL + k] = (float)red[0];
            __syncthreads();
        }
}
__global__ void diff_k(const float *a, const float *b, long n, double *acc) {   // acc: sum (a-b)^2, sum b^2, max |a-b|, max |b|
    double s0 = 0, s1 = 0, m0 = 0, m1 = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double d = (double)a[i] - (double)b[i];
        s0 += d * d; s1 += (double)b[i] * b[i];
        m0 = fmax(m0, fabs(d)); m1 = fmax(m1, fabs((double)b[i]));
    }
    atomicAdd(&acc[0], s0); atomicAdd(&acc[1], s1);
    atomicMax((unsigned long long *)&acc[2], (unsigned long long)__double_as_longlong(m0));
    atomicMax((unsigned long long *)&acc[3], (unsigned long long)__double_as_longlong(m1));
}
int main(int argc, char **argv) {
    const int Na = 251, Nb = 251, ha = 126, hb = 126, NAP = 256, NBP = 256, KBP = 128;
    const long LP = argc > 1 ? atol(argv[1]) : 4096;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const int MP = 128, KP = 128;
    float *cube, *ycol, *spec, *mhat, *tpl, *outA, *outB;
    unsigned short *A3, *img;
    double *acc;
    const size_t ncube = (size_t)NBP * NAP * LP, nsp = (size_t)2 * NAP * KBP * LP;
    const size_t nout = ncube > nsp ? ncube : nsp;
    CK(hipMalloc(&cube, ncube * 4)); CK(hipMalloc(&ycol, nsp * 4)); CK(hipMalloc(&spec, nsp * 4));
    CK(hipMalloc(&outA, nout * 4)); CK(hipMalloc(&outB, nout * 4)); CK(hipMalloc(&acc, 32));
    CK(hipMalloc(&mhat, (size_t)4 * 2 * NAP * KBP * 4)); CK(hipMalloc(&tpl, (size_t)4 * LP * 4));
    // matrices: cos / sin of the 251-point transform, ortho-normalised, [MP][KP]
    std::vector<float> C((size_t)MP * KP, 0.f), S(C.size(), 0.f);
    for (int r = 0; r < ha; ++r)
        for (int k = 0; k < ha; ++k) {
            const double th = 2.0 * M_PI * (double)(((long)r * k) % Na) / (double)Na;
            C[(size_t)r * KP + k] = (float)(std::cos(th) / std::sqrt((double)Na));
            S[(size_t)r * KP + k] = (float)(std::sin(th) / std::sqrt((double)Na));
        }
    std::vector<unsigned short> a3((size_t)6 * MP * KP);
    for (int m = 0; m < 2; ++m) {
        const std::vector<float> &M = m ? S : C;
        unsigned short *o = a3.data() + (size_t)m * 3 * MP * KP;
        for (size_t j = 0; j < M.size(); ++j) {
            float x = M[j], hh, mm; unsigned u;
            memcpy(&u, &x, 4); u &= 0xFFFF0000u; memcpy(&hh, &u, 4); o[j] = (unsigned short)(u >> 16);
            float r = x - hh; memcpy(&u, &r, 4); u &= 0xFFFF0000u; memcpy(&mm, &u, 4); o[M.size() + j] = (unsigned short)(u >> 16);
            r -= mm; memcpy(&u, &r, 4); u += 0x7FFFu + ((u >> 16) & 1u); o[2 * M.size() + j] = (unsigned short)(u >> 16);
        }
    }
    std::vector<unsigned short> im(DFT_H2_IMAGE_HALFS);
    const int kA = dft_h2_build_image(C.data(), S.data(), MP, KP, KP, im.data());
    printf("kA = %d\n", kA);
    CK(hipMalloc(&A3, a3.size() * 2)); CK(hipMemcpy(A3, a3.data(), a3.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&img, im.size() * 2)); CK(hipMemcpy(img, im.data(), im.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, cube, (long)ncube, 1u, mode, (long)NAP * LP);
    hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, ycol, (long)nsp, 2u, mode, LP);
    hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, spec, (long)nsp, 3u, mode, (long)KBP * LP);
    hipLaunchKernelGGL(fill_k, dim3(64), dim3(256), 0, 0, mhat, (long)4 * 2 * NAP * KBP, 5u, 0, 1L);
    hipLaunchKernelGGL(fill_k, dim3(64), dim3(256), 0, 0, tpl, (long)4 * LP, 6u, 0, 1L);
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool only_h2 = getenv("ONLY_H2") != nullptr;
    float *ilv, *outI;     // interleaved copies of the complex source / h2's complex output
    CK(hipMalloc(&ilv, nsp * 4)); CK(hipMalloc(&outI, nsp * 4));
    const long plane = (long)NAP * KBP * LP;     // floats of one component plane (both intermediate layouts)
    for (int pass = 0; pass < 5; ++pass) {
        DftRx3Args g;
        DftH2Args q;
        g.A[0] = A3; g.A[1] = A3 + 3 * MP * KP; g.planeA = (long)MP * KP; g.lda = KP; g.MP = MP; g.KP = KP; q.KP = KP;
        const char *name = "";
        size_t nd = 0;
        if (pass == 0) {          // r2c along beta
            name = "rows_fwd (r2c)"; g.src[0] = cube; g.src[1] = cube; g.ldb = NAP * LP; g.fold[0] = 1; g.fold[1] = -1; g.Kn = Nb;
            g.ldc = NAP * LP; g.mode = 1; g.e11 = -1; g.rvalid = hb; g.N = (int)(Na * LP); nd = nsp;
            q.kind = 1; q.src = cube; q.ldb = NAP * LP; q.Kn = Nb; q.dst = outI; q.ldc = 2 * NAP * LP; q.e[0] = 1; q.e[3] = -1; q.rvalid = hb; q.N = (int)(Na * LP);
        } else if (pass == 1) {   // c2c along alpha batched over kb
            name = "cols_fwd (c2c)"; g.src[0] = ycol; g.src[1] = ycol + plane; g.ldb = LP; g.sB = NAP * LP; g.fold[0] = 1; g.fold[1] = -1; g.Kn = Na;
            g.ldc = KBP * LP; g.sC = LP; g.mode = 0; g.Rn = Na; g.rvalid = ha; g.e01 = 1; g.e10 = 1; g.e11 = -1; g.N = (int)LP; g.batch = hb;
            g.nvar = 2; g.A_alt[0] = g.A[1]; g.A_alt[1] = g.A[0]; g.fold_alt[0] = -1; g.fold_alt[1] = 1;
            g.e_alt[0] = -1; g.e_alt[1] = 1; g.e_alt[2] = 1; g.e_alt[3] = 1;
            g.packed = 1; nd = nsp;
            hipLaunchKernelGGL(ilv_k, dim3(2048), dim3(256), 0, st, ycol, ycol + plane, ilv, plane);
            q.kind = 0; q.src = ilv; q.ldb = 2 * LP; q.sB = 2 * NAP * LP; q.Kn = Na; q.dst = outI; q.ldc = 2 * KBP * LP; q.sC = 2 * LP;
            q.Rn = Na; q.rvalid = ha; q.N = (int)LP; q.batch = hb;
            q.e[0] = 1; q.e[1] = 1; q.e[2] = 1; q.e[3] = -1; q.e_alt[0] = 1; q.e_alt[1] = -1; q.e_alt[2] = 1; q.e_alt[3] = 1;
        } else if (pass == 2 || pass == 4) {   // c2c along alpha, unbatched wide N (4: with the fused spectral mix)
            name = pass == 2 ? "cols_inv (c2c)" : "cols_inv_mix"; g.src[0] = spec; g.src[1] = spec + plane; g.ldb = KBP * LP; g.fold[0] = 1; g.fold[1] = -1; g.Kn = Na;
            g.ldc = KBP * LP; g.mode = 0; g.Rn = Na; g.rvalid = ha; g.e01 = -1; g.e10 = 1; g.e11 = 1; g.N = (int)(hb * LP);
            g.nvar = 2; g.A_alt[0] = g.A[1]; g.A_alt[1] = g.A[0]; g.fold_alt[0] = -1; g.fold_alt[1] = 1;
            g.e_alt[0] = 1; g.e_alt[1] = 1; g.e_alt[2] = -1; g.e_alt[3] = 1;
            g.packed = 1; nd = nsp;
            hipLaunchKernelGGL(ilv_k, dim3(2048), dim3(256), 0, st, spec, spec + plane, ilv, plane);
            q.kind = 0; q.src = ilv; q.ldb = 2 * KBP * LP; q.Kn = Na; q.dst = outI; q.ldc = 2 * KBP * LP; q.Rn = Na; q.rvalid = ha; q.N = (int)(hb * LP);
            q.e[0] = 1; q.e[1] = -1; q.e[2] = 1; q.e[3] = 1; q.e_alt[0] = 1; q.e_alt[1] = 1; q.e_alt[2] = 1; q.e_alt[3] = -1;
            if (pass == 4) {
                g.mhat = mhat; g.tpl = tpl; g.T = 4; g.LP = (int)LP; g.PL = (long)NAP * KBP; g.KBP = KBP;
                q.mhat = mhat; q.tpl = tpl; q.T = 4; q.LP = (int)LP; q.PL = (long)NAP * KBP; q.KBP = KBP;
            }
        } else {                  // c2r along beta batched over alpha
            name = "rows_inv (c2r)"; g.src[0] = ycol; g.src[1] = ycol + plane; g.ldb = LP; g.sB = KBP * LP;
            g.ldc = NAP * LP; g.sC = LP; g.mode = 0; g.e01 = -1; g.e10 = 1; g.e11 = 1; g.Rn = Nb; g.rvalid = hb; g.N = (int)LP; g.batch = Na; nd = ncube;
            hipLaunchKernelGGL(ilv_k, dim3(2048), dim3(256), 0, st, ycol, ycol + plane, ilv, plane);
            q.kind = 2; q.src = ilv; q.ldb = 2 * LP; q.sB = 2 * KBP * LP; q.dst = outB; q.ldc = NAP * LP; q.sC = LP;
            q.e[0] = 1; q.e[1] = -1; q.e[2] = 1; q.e[3] = 1; q.Rn = Nb; q.rvalid = hb; q.N = (int)LP; q.batch = Na;
        }
        for (int which = only_h2 ? 1 : 0; which < 2; ++which) {
            float *out = which ? outB : outA;
            CK(hipMemsetAsync(out, 0, nout * 4, st));
            if (which) CK(hipMemsetAsync(outI, 0, nsp * 4, st));
            if (pass == 0) { g.dst[0] = out; g.dst[1] = out + plane; }
            else if (pass == 3) { g.dst[0] = out; g.dst[1] = nullptr; }
            else { g.dst[0] = out; g.dst[1] = nullptr; g.dst_alt = out + plane; }
            auto launch = [&]() { return which ? launch_dft_h2(st, q, img, kA) : launch_dft_rx3(st, g); };
            for (int i = 0; i < 3; ++i) { int rc = launch(); if (rc) { printf("launch rc %d\n", rc); return 1; } }
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            const int reps = 20;
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-16s %-4s %.4f ms\n", name, which ? "h2" : "rx3", ms / reps);
            if (which && pass != 3) hipLaunchKernelGGL(dilv_k, dim3(2048), dim3(256), 0, st, outI, outB, outB + plane, plane);
        }
        if (!only_h2) {
            CK(hipMemsetAsync(acc, 0, 32, st));
            hipLaunchKernelGGL(diff_k, dim3(1024), dim3(256), 0, st, outB, outA, (long)nd, acc);
            double hacc[4];
            CK(hipMemcpyAsync(hacc, acc, 32, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            printf("%-16s h2 vs rx3: rel L2 %.3e   max|d| %.3e (max|ref| %.3e)\n", name, std::sqrt(hacc[0] / hacc[1]), hacc[2], hacc[3]);
        }
    }
    // ---- the forward transform (mix + c2c along alpha, then c2r along beta) whole-cube against wavelength chunks whose
    // half-transformed intermediate is a small reused buffer (Infinity-Cache resident), chunks alternating on two streams
    {
        hipStream_t st2[2]; CK(hipStreamCreate(&st2[0])); CK(hipStreamCreate(&st2[1]));
        hipLaunchKernelGGL(ilv_k, dim3(2048), dim3(256), 0, st, spec, spec + plane, ilv, plane);
        CK(hipStreamSynchronize(st));
        auto whole = [&](float *cube_out) {
            DftH2Args q; q.KP = KP;
            q.kind = 0; q.src = ilv; q.ldb = 2 * KBP * LP; q.Kn = Na; q.dst = outI; q.ldc = 2 * KBP * LP; q.Rn = Na; q.rvalid = ha; q.N = (int)(hb * LP);
            q.e[0] = 1; q.e[1] = -1; q.e[2] = 1; q.e[3] = 1; q.e_alt[0] = 1; q.e_alt[1] = 1; q.e_alt[2] = 1; q.e_alt[3] = -1;
            q.mhat = mhat; q.tpl = tpl; q.T = 4; q.LP = (int)LP; q.PL = (long)NAP * KBP; q.KBP = KBP;
            launch_dft_h2(st, q, img, kA);
            DftH2Args r; r.KP = KP;
            r.kind = 2; r.src = outI; r.ldb = 2 * LP; r.sB = 2 * KBP * LP; r.dst = cube_out; r.ldc = NAP * LP; r.sC = LP;
            r.e[0] = 1; r.e[1] = -1; r.e[2] = 1; r.e[3] = 1; r.Rn = Nb; r.rvalid = hb; r.N = (int)LP; r.batch = Na;
            launch_dft_h2(st, r, img, kA);
        };
        for (int i = 0; i < 2; ++i) whole(outA);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) whole(outA);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("forward transform whole cube          %.4f ms\n", ms / 10);
        for (long LPc : {256L, 512L, 1024L}) {
            if (LPc > LP) continue;
            float *yb[2];
            CK(hipMalloc(&yb[0], (size_t)2 * NAP * KBP * LPc * 4)); CK(hipMalloc(&yb[1], (size_t)2 * NAP * KBP * LPc * 4));
            for (int nstream = 1; nstream <= 2; ++nstream) {
                auto chunked = [&](float *cube_out) {
                    for (long c = 0; c * LPc < LP; ++c) {
                        hipStream_t s_ = st2[nstream == 2 ? (c & 1) : 0];
                        float *y = yb[nstream == 2 ? (c & 1) : 0];
                        const long l0 = c * LPc;
                        DftH2Args q; q.KP = KP;
                        q.kind = 0; q.src = ilv + 2 * l0; q.ldb = 2 * KBP * LP; q.sB = 2 * LP; q.batch = hb; q.Kn = Na;
                        q.dst = y; q.ldc = 2 * KBP * LPc; q.sC = 2 * LPc; q.Rn = Na; q.rvalid = ha; q.N = (int)LPc;
                        q.e[0] = 1; q.e[1] = -1; q.e[2] = 1; q.e[3] = 1; q.e_alt[0] = 1; q.e_alt[1] = 1; q.e_alt[2] = 1; q.e_alt[3] = -1;
                        q.mhat = mhat; q.tpl = tpl; q.T = 4; q.LP = (int)LP; q.PL = (long)NAP * KBP; q.KBP = KBP; q.mix_l0 = (int)l0;
                        launch_dft_h2(s_, q, img, kA);
                        DftH2Args r; r.KP = KP;
                        r.kind = 2; r.src = y; r.ldb = 2 * LPc; r.sB = 2 * KBP * LPc; r.dst = cube_out + l0; r.ldc = NAP * LP; r.sC = LP;
                        r.e[0] = 1; r.e[1] = -1; r.e[2] = 1; r.e[3] = 1; r.Rn = Nb; r.rvalid = hb; r.N = (int)LPc; r.batch = Na;
                        launch_dft_h2(s_, r, img, kA);
                    }
                };
                for (int i = 0; i < 2; ++i) chunked(outB);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, st2[0]));
                for (int i = 0; i < 10; ++i) chunked(outB);
                CK(hipStreamSynchronize(st2[1]));
                CK(hipEventRecord(e1, st2[0])); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemsetAsync(acc, 0, 32, st));
                hipLaunchKernelGGL(diff_k, dim3(1024), dim3(256), 0, st, outB, outA, (long)ncube, acc);
                double hacc[4];
                CK(hipMemcpyAsync(hacc, acc, 32, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                printf("forward transform chunks of %4ld, %d stream(s)  %.4f ms   (vs whole: rel L2 %.3e)\n", LPc, nstream, ms / 10, std::sqrt(hacc[0] / hacc[1]));
            }
            CK(hipFree(yb[0])); CK(hipFree(yb[1]));
        }
    }
#ifndef NO_ADJMIX
    // ---- fused adjoint tail: complex pass along alpha + conj(H) + wavelength reduction against pass + float64 reference
    {
        const long PL = (long)NAP * KBP;
        float *Hi, *madjA, *madjB, *mpart;
        CK(hipMalloc(&Hi, nsp * 4)); CK(hipMalloc(&madjA, (size_t)8 * PL * 4)); CK(hipMalloc(&madjB, (size_t)8 * PL * 4));
        const size_t npart = dft_h2_adjmix_part_floats(LP, hb);
        CK(hipMalloc(&mpart, npart * 4));
        printf("adjmix partial buffer: %.1f MB\n", npart * 4 / 1e6);
        hipLaunchKernelGGL(ilv_k, dim3(2048), dim3(256), 0, st, ycol, ycol + plane, ilv, plane);      // Z[kb][a][l][2]
        hipLaunchKernelGGL(ilv_k, dim3(2048), dim3(256), 0, st, spec, spec + plane, Hi, plane);        // H[ka][kb][l][2]
        CK(hipMemsetAsync(madjA, 0, (size_t)8 * PL * 4, st)); CK(hipMemsetAsync(madjB, 0, (size_t)8 * PL * 4, st));
        DftH2Args q; q.KP = KP;
        q.kind = 0; q.src = ilv; q.ldb = 2 * LP; q.sB = 2 * NAP * LP; q.Kn = Na; q.dst = outI; q.ldc = 2 * KBP * LP; q.sC = 2 * LP;
        q.Rn = Na; q.rvalid = ha; q.N = (int)LP; q.batch = hb;
        q.e[0] = 1; q.e[1] = 1; q.e[2] = 1; q.e[3] = -1; q.e_alt[0] = 1; q.e_alt[1] = -1; q.e_alt[2] = 1; q.e_alt[3] = 1;
        CK(hipMemsetAsync(outI, 0, nsp * 4, st));
        if (int rc = launch_dft_h2(st, q, img, kA)) { printf("launch rc %d\n", rc); return 1; }
        hipLaunchKernelGGL(adjmix_ref_k, dim3((unsigned)PL), dim3(256), 0, st, outI, Hi, tpl, madjA, 4, LP, (long)KBP, PL);
        DftH2AdjMix am; am.hsrc = Hi; am.ldh = 2 * KBP * LP; am.sH = 2 * LP; am.tpl = tpl; am.T = 4; am.LPt = (int)LP; am.mpart = mpart;
        for (int i = 0; i < 3; ++i) if (int rc = launch_dft_h2_adjmix(st, q, am, madjB, PL, KBP, img, kA)) { printf("adjmix launch rc %d\n", rc); return 1; }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; ++i) launch_dft_h2_adjmix(st, q, am, madjB, PL, KBP, img, kA);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemsetAsync(acc, 0, 32, st));
        hipLaunchKernelGGL(diff_k, dim3(1024), dim3(256), 0, st, madjB, madjA, (long)8 * PL, acc);
        double hacc[4];
        CK(hipMemcpyAsync(hacc, acc, 32, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        printf("fused adjoint tail (pass + conj(H) + sum over l)  %.4f ms   vs float64 reference: rel L2 %.3e max|d| %.3e (max|ref| %.3e)\n",
               ms / 20, std::sqrt(hacc[0] / hacc[1]), hacc[2], hacc[3]);
    }
#endif
    return 0;
}
