// Probe: the r2c pass of dft_rx3 as ONE wave per SIMD (512-register budget) with a 4-deep ring of raw data loads issued
// three k-steps ahead, matrix tiles register-staged (so that the compiler keeps counted vmcnt waits), persistent
// workgroups (one per CU).  Timing only (zero operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 16, RS = 16, PIECE = 128 * RS, IMG = 3 * PIECE, BUF = 2 * IMG;
constexpr size_t LDS_BYTES = (size_t)2 * BUF * 2;

struct Args {
    const unsigned short *A[2]; long planeA; int lda;
    const float *src; long ldb, sB; int Kn;
    float *dst[2]; long ldc, sC;
    float e00, e11; int rvalid, N, batch;
};

__device__ __forceinline__ unsigned pack2(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &fh, bf16x8 &fm, bf16x8 &fl) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned u = __float_as_uint(x[j]);
        h[j] = u & 0xFFFF0000u;
        const float r = x[j] - __uint_as_float(h[j]);
        m[j] = __float_as_uint(r) & 0xFFFF0000u;
        l[j] = __float_as_uint(r - __uint_as_float(m[j]));
    }
    fh = __builtin_bit_cast(bf16x8, make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7])));
    fm = __builtin_bit_cast(bf16x8, make_uint4(pack2(m[0], m[1]), pack2(m[2], m[3]), pack2(m[4], m[5]), pack2(m[6], m[7])));
    fl = __builtin_bit_cast(bf16x8, make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7])));
}
#define MFMA6(acc_, ah_, am_, al_, bh_, bm_, bl_)                                   \
    {                                                                               \
        f32x16 c_ = acc_;                                                           \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am_, bm_, c_, 0, 0, 0);        \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bl_, c_, 0, 0, 0);        \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, bh_, c_, 0, 0, 0);        \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bm_, c_, 0, 0, 0);        \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am_, bh_, c_, 0, 0, 0);        \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bh_, c_, 0, 0, 0);        \
        acc_ = c_;                                                                  \
    }

#ifndef DEPTH
#define DEPTH 3
#endif
#ifndef WPS
#define WPS 1
#endif

__global__ __launch_bounds__(256, WPS) void r2c_v2_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int tilesX = g.N / 128, ntile = tilesX * g.batch;
    const unsigned ldb4 = (unsigned)(g.ldb * 4), ldc4 = (unsigned)(g.ldc * 4), c4 = (unsigned)(wave * 32 + l31) * 4u;
    const int kin = g.Kn / 2 + 1;
    const int arow = tid >> 1;
    const unsigned aoff = (unsigned)(arow * g.lda + 8 * (tid & 1)) * 2u;
    const unsigned awr = (unsigned)(arow * RS + 8 * ((tid & 1) ^ ((arow >> 3) & 1)));     // swizzled LDS position
    const char *A0 = (const char *)g.A[0], *A1 = (const char *)g.A[1];

    int tile = blockIdx.x;             // tile whose MFMAs run
    int ltile = tile;                  // tile the load stream is in
    int lstep = 0;                     // next k-step to load (0..7) of ltile
    const char *LB = (const char *)(g.src + (long)(ltile / tilesX) * g.sB + (ltile % tilesX) * 128);

    int hv = h;
    float r0x[8], r0q[8], r1x[8], r1q[8], r2x[8], r2q[8], r3x[8], r3q[8];
    uint4 pa0, pa1, pa2, pa3, pa4, pa5;
    f32x16 acc1[4], acc2[4];
    bf16x8 b0h, b0m, b0l, b1h, b1m, b1l;

#define V2_BLOAD(X_, Q_)                                                                                        \
    {                                                                                                           \
        if (ltile < ntile) {                                                                                    \
            const unsigned ok = (lstep * BK + 8 * hv) * ldb4 + c4, op = (g.Kn - lstep * BK - 8 * hv) * ldb4 + c4; \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                     \
                const int k = lstep * BK + 8 * hv + j;                                                           \
                const unsigned o = ok + j * ldb4;                                                               \
                const bool pv = (k >= 1) && (k < kin) && (2 * k != g.Kn);                                       \
                X_[j] = *(const float *)(LB + o);                                                               \
                Q_[j] = *(const float *)(LB + (pv ? op - j * ldb4 : o));                                        \
            }                                                                                                   \
        }                                                                                                       \
        if (++lstep == 8) {                                                                                     \
            lstep = 0;                                                                                          \
            ltile += gridDim.x;                                                                                 \
            LB = (const char *)(g.src + (long)(ltile / tilesX) * g.sB + (ltile % tilesX) * 128);               \
        }                                                                                                       \
    }
#define V2_FOLD(kt_, X_, Q_)                                                                                    \
    {                                                                                                           \
        float x0[8], x1[8];                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                         \
            const int k = (kt_) * BK + 8 * hv + j;                                                               \
            const bool pv = (k >= 1) && (k < kin) && (2 * k != g.Kn);                                           \
            x0[j] = X_[j] + (pv ? Q_[j] : 0.f);                                                                 \
            x1[j] = pv ? X_[j] - Q_[j] : 0.f;                                                                   \
        }                                                                                                       \
        split8(x0, b0h, b0m, b0l);                                                                              \
        split8(x1, b1h, b1m, b1l);                                                                              \
    }
#define V2_ALOAD(kt_)                                                                                           \
    {                                                                                                           \
        const unsigned ao = aoff + (unsigned)((kt_) * BK) * 2u;                                                 \
        pa0 = *(const uint4 *)(A0 + ao); pa1 = *(const uint4 *)(A0 + 2 * g.planeA + ao); pa2 = *(const uint4 *)(A0 + 4 * g.planeA + ao); \
        pa3 = *(const uint4 *)(A1 + ao); pa4 = *(const uint4 *)(A1 + 2 * g.planeA + ao); pa5 = *(const uint4 *)(A1 + 4 * g.planeA + ao); \
    }
#define V2_ASTORE(buf_)                                                                                         \
    {                                                                                                           \
        unsigned short *pA = lds + (buf_) * BUF + awr;                                                          \
        *(uint4 *)(pA) = pa0; *(uint4 *)(pA + PIECE) = pa1; *(uint4 *)(pA + 2 * PIECE) = pa2;                   \
        *(uint4 *)(pA + IMG) = pa3; *(uint4 *)(pA + IMG + PIECE) = pa4; *(uint4 *)(pA + IMG + 2 * PIECE) = pa5; \
    }
#define V2_MFMA(buf_)                                                                                           \
    {                                                                                                           \
        const unsigned short *ra = lds + (buf_) * BUF + l31 * RS + 8 * (h ^ ((l31 >> 3) & 1));                  \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                      \
            const unsigned short *p = ra + mt * 32 * RS;                                                        \
            const bf16x8 a0h = *(const bf16x8 *)(p), a0m = *(const bf16x8 *)(p + PIECE), a0l = *(const bf16x8 *)(p + 2 * PIECE); \
            MFMA6(acc1[mt], a0h, a0m, a0l, b0h, b0m, b0l)                                                       \
            const bf16x8 a1h = *(const bf16x8 *)(p + IMG), a1m = *(const bf16x8 *)(p + IMG + PIECE), a1l = *(const bf16x8 *)(p + IMG + 2 * PIECE); \
            MFMA6(acc2[mt], a1h, a1m, a1l, b1h, b1m, b1l)                                                       \
        }                                                                                                       \
    }
    // one k-step: loads DEPTH steps ahead into ring slot (kt_ + DEPTH) % 4, matrix tile of kt_+1, MFMAs of kt_, fold of kt_+1
#define V2_STEP(kt_, XL_, QL_, XN_, QN_)                                                                        \
    {                                                                                                           \
        asm volatile("" : "+v"(hv));   /* keeps per-step lane masks and offsets from being hoisted out of the loop */ \
        V2_ALOAD(((kt_) + 1) & 7);                                                                              \
        V2_BLOAD(XL_, QL_);                                                                                     \
        V2_MFMA((kt_) & 1);                                                                                     \
        if ((kt_) == 7) {                                                                                       \
            char *D0 = (char *)(g.dst[0] + (long)(tile / tilesX) * g.sC + (tile % tilesX) * 128);               \
            char *D1 = (char *)(g.dst[1] + (long)(tile / tilesX) * g.sC + (tile % tilesX) * 128);               \
            unsigned oo = c4 + (unsigned)(4 * h) * ldc4;                                                        \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int r = 0; r < 16; ++r) {   \
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;                                       \
                if (row < g.rvalid) {                                                                           \
                    *(float *)(D0 + oo) = g.e00 * acc1[mt][r];                                                  \
                    *(float *)(D1 + oo) = g.e11 * acc2[mt][r];                                                  \
                }                                                                                               \
                oo += ((r & 3) == 3) ? 5u * ldc4 : ldc4;                                                        \
            }                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int r = 0; r < 16; ++r) acc1[i][r] = acc2[i][r] = 0.f; \
            tile += gridDim.x;                                                                                  \
        }                                                                                                       \
        V2_ASTORE(((kt_) + 1) & 1);                                                                             \
        V2_FOLD(((kt_) + 1) & 7, XN_, QN_);                                                                     \
        __syncthreads();                                                                                        \
    }

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[i][r] = acc2[i][r] = 0.f;
    // prologue: fill the ring DEPTH deep
    V2_ALOAD(0);
    V2_BLOAD(r0x, r0q);
#if DEPTH >= 2
    V2_BLOAD(r1x, r1q);
#endif
#if DEPTH >= 3
    V2_BLOAD(r2x, r2q);
#endif
    V2_ASTORE(0);
    V2_FOLD(0, r0x, r0q);
    __syncthreads();
    while (tile < ntile) {
#if DEPTH == 3
        V2_STEP(0, r3x, r3q, r1x, r1q) V2_STEP(1, r0x, r0q, r2x, r2q) V2_STEP(2, r1x, r1q, r3x, r3q) V2_STEP(3, r2x, r2q, r0x, r0q)
        V2_STEP(4, r3x, r3q, r1x, r1q) V2_STEP(5, r0x, r0q, r2x, r2q) V2_STEP(6, r1x, r1q, r3x, r3q) V2_STEP(7, r2x, r2q, r0x, r0q)
#elif DEPTH == 2
        V2_STEP(0, r2x, r2q, r1x, r1q) V2_STEP(1, r3x, r3q, r2x, r2q) V2_STEP(2, r0x, r0q, r3x, r3q) V2_STEP(3, r1x, r1q, r0x, r0q)
        V2_STEP(4, r2x, r2q, r1x, r1q) V2_STEP(5, r3x, r3q, r2x, r2q) V2_STEP(6, r0x, r0q, r3x, r3q) V2_STEP(7, r1x, r1q, r0x, r0q)
#else
        V2_STEP(0, r1x, r1q, r1x, r1q) V2_STEP(1, r2x, r2q, r2x, r2q) V2_STEP(2, r3x, r3q, r3x, r3q) V2_STEP(3, r0x, r0q, r0x, r0q)
        V2_STEP(4, r1x, r1q, r1x, r1q) V2_STEP(5, r2x, r2q, r2x, r2q) V2_STEP(6, r3x, r3q, r3x, r3q) V2_STEP(7, r0x, r0q, r0x, r0q)
#endif
    }
}

int main() {
    const int Na = 251, Nb = 251, hb = 126, NAP = 256, NBP = 256, KBP = 128;
    const long LP = 4096;
    float *cube, *ycol; unsigned short *A;
    CK(hipMalloc(&cube, (size_t)NBP * NAP * LP * 4)); CK(hipMalloc(&ycol, (size_t)2 * NAP * KBP * LP * 4)); CK(hipMalloc(&A, (size_t)6 * 128 * 128 * 2));
    CK(hipMemset(cube, 0, (size_t)NBP * NAP * LP * 4)); CK(hipMemset(A, 0, (size_t)6 * 128 * 128 * 2));
    Args g;
    g.A[0] = A; g.A[1] = A + 3 * 128 * 128; g.planeA = 128 * 128; g.lda = 128;
    g.src = cube; g.ldb = NAP * LP; g.sB = 0; g.Kn = Nb;
    g.dst[0] = ycol; g.dst[1] = ycol + (long)KBP * NAP * LP; g.ldc = NAP * LP; g.sC = 0; g.e00 = 1; g.e11 = -1; g.rvalid = hb; g.N = (int)(Na * LP); g.batch = 1;
    CK(hipFuncSetAttribute((const void *)r2c_v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * WPS;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(r2c_v2_kernel, dim3(grid), dim3(256), LDS_BYTES, st, g);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(r2c_v2_kernel, dim3(grid), dim3(256), LDS_BYTES, st, g);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("r2c v2: waves/SIMD %d, loads %d k-steps ahead: %.4f ms\n", WPS, DEPTH, ms / 20);
    return 0;
}
