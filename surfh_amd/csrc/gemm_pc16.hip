// Two-piece fp16 NT GEMM on the matrix cores, producer / consumer workgroup (layout and roles of gemm_pc3.hip).
//
//   C[m][n] = sum_k A[m][k] * B[n][k]        (A fp32 in HBM, B given as its two fp16 pieces)
//
// Every operand is scaled by a power of two so that its largest magnitude sits at 2^14, then cut into two fp16 values by
// round-to-nearest:  x / s = h + l + e,  |l| <= 2^-11 |h|,  |e| <= 2^-23 |x / s|  (22 mantissa bits; below 2^-18 of the
// largest magnitude the relative precision decreases, the absolute error stays under 2^-39 of the largest magnitude).
// Three products are kept (lh, hl, hh -- each exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_f16); the dropped
// l*l and the remainders e are 2^-22 relative with random sign.  Measured against float64 the split error is 3e-9 (L2), a
// tenth of the six-product truncating bf16 split (which also carries a -4e-8 bias on non-negative data), for half the
// matrix-core work.  What remains is the fp32 accumulation itself.
//   * B is constant on this path (the spectral PSF): split once at plan creation (launch_split2h), scale sB a host constant.
//   * A is data: one scale PER ROW, so that an outlier in one row (a hot detector pixel) does not cost the other rows their
//     precision.  The kernel that writes A leaves per-wave maxima, a small pass reduces them to max|A[m][:]| per row (bit
//     patterns of non-negative floats); the producers derive each row's power of two and split on the fly, the consumers
//     undo it when they store.
#include "gemm_f32.h"
#include "lds_attr.h"
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 256, BK = 32;
constexpr int ROWS = BM + BN;
constexpr int PIECE = ROWS * BK;              // one fp16 piece of one stage (elements): 24 KB
constexpr int STAGE = 2 * PIECE;              // 48 KB
constexpr int NSTAGE = 2;
constexpr size_t LDS_BYTES = (size_t)NSTAGE * STAGE * sizeof(unsigned short);
constexpr int NF = BM / 32;                   // float4 loads per producer thread and K step (A rows, 32 per pass)
constexpr int NSP = BN / 64, NS = 2 * NSP;    // 16-byte loads of the pre-split B rows (64 per pass, 2 pieces)

// power of two that brings `amax` to [2^13, 2^14]; 1 for an all-zero operand
__device__ __forceinline__ float scale_of(float amax) {
    if (!(amax > 0.f)) return 1.f;
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127;      // floor(log2(amax)) for normal values
    int s = e - 13;
    s = s < -126 ? -126 : (s > 127 ? 127 : s);
    return __uint_as_float((unsigned)(s + 127) << 23);
}

__device__ __forceinline__ void store_split(unsigned short *dst, float4 v, float inv) {
    const float x0 = v.x * inv, x1 = v.y * inv, x2 = v.z * inv, x3 = v.w * inv;
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1, h2 = (_Float16)x2, h3 = (_Float16)x3;
    f16x4 h = {h0, h1, h2, h3};
    f16x4 l = {(_Float16)(x0 - (float)h0), (_Float16)(x1 - (float)h1), (_Float16)(x2 - (float)h2), (_Float16)(x3 - (float)h3)};
    *reinterpret_cast<f16x4 *>(dst) = h;
    *reinterpret_cast<f16x4 *>(dst + PIECE) = l;
}

__global__ __launch_bounds__(512, 1) void gemm_nt_f16x2_pc_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // work order: see gemm_pc3.hip (slab-major, B tile, A tile; one contiguous range per XCD)
    const int tilesM = g.M / BM, tilesN = (g.N + BN - 1) / BN, tiles = tilesM * tilesN;
    const long total = (long)tiles * g.splitK * g.batch, per = (total + 7) / 8;
    const long v = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((long)(blockIdx.x >> 3) >= per || v >= total) return;
    const int t = (int)(v % tiles), z = (int)(v / tiles);
    const int tm = t % tilesM, tn = t / tilesM;
    const int b = z / g.splitK, sk = z % g.splitK;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Kper = g.K / g.splitK, kbeg = sk * Kper, nk = Kper / BK;
    const unsigned *rmax = g.amax + (long)b * g.M + m0;              // max |A[m][:]| of this tile's rows (bit patterns)

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int t = tid - 256;
        const int r = t >> 3, c8 = t & 7;
        const int rs = t >> 2, c4 = t & 3;
        const float *fsrc[NF];
        int fpos[NF];
        float finv[NF];                                           // 1 / scale of the row: exact, a power of two
        const unsigned short *ssrc[NS];
        int spos[NS];
        const float *Ab = g.A0 + (long)b * g.sA + kbeg + 4 * c8;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int row = r + 32 * i;
            fsrc[i] = Ab + (long)(m0 + row) * g.lda;
            fpos[i] = row * BK + (((c8 >> 1) ^ ((row >> 2) & 3)) * 8) + (c8 & 1) * 4;
            finv[i] = 1.f / scale_of(__uint_as_float(rmax[row]));
        }
#pragma unroll
        for (int p = 0; p < NSP; ++p) {
            const int row = BM + rs + 64 * p;
            int n = n0 + row - BM;
            n = n < g.N ? n : g.N - 1;                             // ragged last tile: clamp, the columns are not stored
            const unsigned short *base = g.B16 + (long)b * g.sB + (long)n * g.ldb + kbeg + 8 * c4;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ssrc[2 * p + q] = base + q * g.pB16;
                spos[2 * p + q] = q * PIECE + row * BK + ((c4 ^ ((row >> 2) & 3)) * 8);
            }
        }
        float4 fcur[NF], fnxt[NF];
        u32x4 scur[NS], snxt[NS];
#define PH_LOAD(kt_, f_, s_)                                                                                            \
    {                                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NF; ++i) f_[i] = *reinterpret_cast<const float4 *>(fsrc[i] + (kt_) * BK); \
        _Pragma("unroll") for (int i = 0; i < NS; ++i) s_[i] = *reinterpret_cast<const u32x4 *>(ssrc[i] + (kt_) * BK);  \
    }
#define PH_STORE(st_, f_, s_)                                                                                           \
    {                                                                                                                   \
        unsigned short *base = lds + (st_) * STAGE;                                                                     \
        _Pragma("unroll") for (int i = 0; i < NF; ++i) store_split(base + fpos[i], f_[i], finv[i]);                     \
        _Pragma("unroll") for (int i = 0; i < NS; ++i) *reinterpret_cast<u32x4 *>(base + spos[i]) = s_[i];              \
    }
        PH_LOAD(0, fcur, scur);
        if (nk > 1) PH_LOAD(1, fnxt, snxt);
        PH_STORE(0, fcur, scur);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int i = 0; i < NF; ++i) fcur[i] = fnxt[i];
#pragma unroll
            for (int i = 0; i < NS; ++i) scur[i] = snxt[i];
            if (kt + 2 < nk) PH_LOAD(kt + 2, fnxt, snxt);
            if (kt + 1 < nk) PH_STORE((kt + 1) & 1, fcur, scur);
            __syncthreads();
        }
#undef PH_LOAD
#undef PH_STORE
        return;
    }

    // ---------------------------------------------------------------------- consumers
    __builtin_amdgcn_s_setprio(1);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int sw = (l31 >> 2) & 3;
    const int fa = (wm * 64 + l31) * BK;
    const int fb = (BM + wn * 128 + l31) * BK;
    char *Cb = reinterpret_cast<char *>(g.C + (long)b * g.sC + (long)sk * g.sCsplit + (long)m0 * g.ldc + n0);
    const unsigned ldc4 = (unsigned)(g.ldc * 4);
    const unsigned cbase = (unsigned)(wm * 64 + 4 * h) * ldc4 + (unsigned)(wn * 128 + l31) * 4u;
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned short *st = lds + (kt & 1) * STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int ch = ((2 * s2 + h) ^ sw) * 8;
            f16x8 a[2][2], bq[4][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][q] = *reinterpret_cast<const f16x8 *>(st + q * PIECE + fa + i * 32 * BK + ch);
#pragma unroll
                for (int j = 0; j < 4; ++j) bq[j][q] = *reinterpret_cast<const f16x8 *>(st + q * PIECE + fb + j * 32 * BK + ch);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], bq[j][0], c, 0, 0, 0);   // l*h
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], bq[j][1], c, 0, 0, 0);   // h*l
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], bq[j][0], c, 0, 0, 0);   // h*h
                    acc[i][j] = c;
                }
        }
        __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);
    {
        // undo both operand scales (powers of two: exact); all 32 row maxima of this lane requested together
        unsigned rm[2][16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) rm[i][r] = rmax[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        unsigned o_ = cbase;
        asm volatile("" : "+v"(o_));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sc = scale_of(__uint_as_float(rm[i][r])) * g.sB16;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n0 + wn * 128 + j * 32 + l31 < g.N) *reinterpret_cast<float *>(Cb + (o_ + 128u * j)) = acc[i][j][r] * sc;
                o_ += ((r & 3) == 3) ? 5u * ldc4 : ldc4;
            }
    }
}

// dst[q*plane + i] = fp16 piece q (h, l) of src[i] / scale, round to nearest
__global__ __launch_bounds__(256) void split2h_kernel(const float *__restrict__ src, unsigned short *__restrict__ dst, long n4, long plane,
                                                      float inv) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(src)[i];
        const float x0 = v.x * inv, x1 = v.y * inv, x2 = v.z * inv, x3 = v.w * inv;
        const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1, h2 = (_Float16)x2, h3 = (_Float16)x3;
        f16x4 h = {h0, h1, h2, h3};
        f16x4 l = {(_Float16)(x0 - (float)h0), (_Float16)(x1 - (float)h1), (_Float16)(x2 - (float)h2), (_Float16)(x3 - (float)h3)};
        *reinterpret_cast<f16x4 *>(dst + 4 * i) = h;
        *reinterpret_cast<f16x4 *>(dst + plane + 4 * i) = l;
    }
}

}  // namespace

// power-of-two scale for an operand whose largest magnitude is amax (host side of scale_of)
float gemm_f16x2_scale(float amax) {
    if (!(amax > 0.f)) return 1.f;
    int e = 0;
    std::frexp(amax, &e);                  // amax = f * 2^e, f in [0.5, 1)  ->  floor(log2(amax)) = e - 1
    int s = (e - 1) - 13;
    s = s < -126 ? -126 : (s > 127 ? 127 : s);
    return std::ldexp(1.f, s);
}

int launch_split2h(hipStream_t stream, const float *src, unsigned short *dst2, long n, long plane, float scale) {
    if (n % 4 || plane % 4 || !(scale > 0.f)) return (int)hipErrorInvalidValue;
    const long n4 = n / 4;
    long nb = (n4 + 255) / 256;
    hipLaunchKernelGGL(split2h_kernel, dim3((unsigned)(nb > 4096 ? 4096 : (nb < 1 ? 1 : nb))), dim3(256), 0, stream, src, dst2, n4, plane,
                       1.f / scale);
    return (int)hipGetLastError();
}

// C[M][N] = A[M][K] * B[N][K]^T ; M multiple of 128, N of 128 (ragged last 256-column tile handled), K of 32*splitK.
// B as fp16 pieces (g.B16, g.pB16, g.sB16); A fp32 with its per-row maxima g.amax[batch][M] (bit patterns).
int launch_gemm_nt_f16x2_pc(hipStream_t stream, const GemmArgs &g) {
    if (g.M % BM || g.N % 128 || g.K % (BK * g.splitK) || g.splitK < 1 || g.batch < 1 || g.accumulate || g.lda % 4 || g.ldb % 8 ||
        !g.B16 || g.pB16 % 8 || !(g.sB16 > 0.f) || !g.amax)
        return (int)hipErrorInvalidValue;
    if ((double)(BM + 1) * (double)g.ldc * 4.0 >= 2147483648.0) return (int)hipErrorInvalidValue;
    const long total = (long)(g.M / BM) * ((g.N + BN - 1) / BN) * g.batch * g.splitK;
    dim3 grid((unsigned)(8 * ((total + 7) / 8)));
    static unsigned long long attr_done = 0;
    if (int e = ensure_dynamic_lds(gemm_nt_f16x2_pc_kernel, LDS_BYTES, attr_done)) return e;
    hipLaunchKernelGGL(gemm_nt_f16x2_pc_kernel, grid, dim3(512), LDS_BYTES, stream, g);
    return (int)hipGetLastError();
}
