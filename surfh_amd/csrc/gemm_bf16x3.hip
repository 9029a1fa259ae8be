// fp32-accurate GEMM on the bf16 matrix cores of gfx950 by exact three-way splitting.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]        (both operands K-contiguous, fp32 in HBM)
//
// Every fp32 operand x is cut into three bf16 pieces x = h + m + l by TRUNCATION of successive
// remainders (24 mantissa bits = 3 x 8), so the split is exact.  Of the nine partial products the six
// of order >= 2^-16 are kept (hh, hm, mh, hl, lh, mm); the dropped ones (ml, lm, ll) are below
// 2^-23 relative, i.e. the size of the fp32 product rounding itself.  Accumulation is fp32 in the MFMA
// accumulators.  v_mfma_f32_32x32x16_bf16 retires 16x the multiply-adds per cycle of the fp32-input
// MFMA, so six of them cost 6/16 of the fp32 path.
// Operand maps of v_mfma_f32_32x32x16_bf16 (lane l, r = l&31, h = l>>5):
//   A fragment: A[row r][k = 8h + j], j = 0..7 ;  B fragment: B[k = 8h + j][col r] ;  D as the fp32 forms.
#include "gemm_f32.h"
#include "lds_attr.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int RSTR = 24;                      // LDS row stride in bf16 elements (16 + 8 pad = 48 B): conflict-free b128 reads
constexpr int PART = BM * RSTR;               // one bf16 piece of one operand tile (elements)
constexpr int OPER = 3 * PART;                // h, m, l
constexpr int BUF = 2 * OPER;                 // A and B
constexpr size_t LDS_BYTES = (size_t)2 * BUF * sizeof(unsigned short);

__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
    const unsigned u = __float_as_uint(x);
    h = u & 0xFFFF0000u;
    const float r = x - __uint_as_float(h);
    m = __float_as_uint(r) & 0xFFFF0000u;
    l = __float_as_uint(r - __uint_as_float(m));   // <= 8 significant bits: already a bf16 value
}

// pack the bf16 (upper halves) of two fp32 bit patterns into one dword: low = a, high = b
__device__ __forceinline__ unsigned pack2(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // one v_perm_b32

__device__ __forceinline__ void store_split(unsigned short *dst_h, float4 v) {
    unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
    split3(v.x, h0, m0, l0);
    split3(v.y, h1, m1, l1);
    split3(v.z, h2, m2, l2);
    split3(v.w, h3, m3, l3);
    *reinterpret_cast<uint2 *>(dst_h) = make_uint2(pack2(h0, h1), pack2(h2, h3));
    *reinterpret_cast<uint2 *>(dst_h + PART) = make_uint2(pack2(m0, m1), pack2(m2, m3));
    *reinterpret_cast<uint2 *>(dst_h + 2 * PART) = make_uint2(pack2(l0, l1), pack2(l2, l3));
}

__global__ __launch_bounds__(256, 2) void gemm_nt_bf16x3_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int tilesN = g.N / BN;
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
    const int b = blockIdx.z / g.splitK, sk = blockIdx.z % g.splitK;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Kper = g.K / g.splitK, kbeg = sk * Kper, nk = Kper / BK;
    const int lr = tid >> 2, lc = (tid & 3) * 4;       // loader: 64 rows x 4 float4 per pass, 2 passes
    const float *Ab = g.A0 + (long)b * g.sA, *Bb = g.B0 + (long)b * g.sB;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra0, ra1, rb0, rb1;
#define X3_GLOAD(kt_)                                                                                      \
    {                                                                                                      \
        const int k0 = kbeg + (kt_) * BK + lc;                                                             \
        ra0 = *reinterpret_cast<const float4 *>(Ab + (long)(m0 + lr) * g.lda + k0);                        \
        ra1 = *reinterpret_cast<const float4 *>(Ab + (long)(m0 + lr + 64) * g.lda + k0);                   \
        rb0 = *reinterpret_cast<const float4 *>(Bb + (long)(n0 + lr) * g.ldb + k0);                        \
        rb1 = *reinterpret_cast<const float4 *>(Bb + (long)(n0 + lr + 64) * g.ldb + k0);                   \
    }
#define X3_LSTORE(buf_)                                                                                    \
    {                                                                                                      \
        unsigned short *base = lds + (buf_) * BUF;                                                         \
        store_split(base + lr * RSTR + lc, ra0);                                                           \
        store_split(base + (lr + 64) * RSTR + lc, ra1);                                                    \
        store_split(base + OPER + lr * RSTR + lc, rb0);                                                    \
        store_split(base + OPER + (lr + 64) * RSTR + lc, rb1);                                             \
    }

    X3_GLOAD(0);
    X3_LSTORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        X3_GLOAD((kt + 1 < nk) ? kt + 1 : kt);
        const unsigned short *pa = lds + buf * BUF + (wm * 64 + l31) * RSTR + 8 * h;
        const unsigned short *pb = lds + buf * BUF + OPER + (wn * 64 + l31) * RSTR + 8 * h;
        bf16x8 a[2][3], bq[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                a[t][q] = *reinterpret_cast<const bf16x8 *>(pa + t * 32 * RSTR + q * PART);
                bq[t][q] = *reinterpret_cast<const bf16x8 *>(pb + t * 32 * RSTR + q * PART);
            }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x16 c = acc[mt][nt];
                // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][1], bq[nt][1], c, 0, 0, 0);   // m*m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][0], bq[nt][2], c, 0, 0, 0);   // h*l
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][2], bq[nt][0], c, 0, 0, 0);   // l*h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][0], bq[nt][1], c, 0, 0, 0);   // h*m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][1], bq[nt][0], c, 0, 0, 0);   // m*h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][0], bq[nt][0], c, 0, 0, 0);   // h*h
                acc[mt][nt] = c;
            }
        X3_LSTORE(buf ^ 1);
        __syncthreads();
    }
#undef X3_GLOAD
#undef X3_LSTORE

    float *Cb = g.C + (long)b * g.sC + (long)sk * g.sCsplit;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = n0 + wn * 64 + nt * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                Cb[(long)row * g.ldc + col] = acc[mt][nt][r];
            }
        }
}

}  // namespace

// C[M][N] = A[M][K] * B[N][K]^T ; M, N multiples of 128, K of 16*splitK.  GemmArgs.B0/ldb describe B as [N][K].
int launch_gemm_nt_bf16x3(hipStream_t stream, const GemmArgs &g) {
    if (g.M % BM || g.N % BN || g.K % (BK * g.splitK) || g.splitK < 1 || g.batch < 1 || g.accumulate) return (int)hipErrorInvalidValue;
    static unsigned long long attr_done = 0;
    if (int e = ensure_dynamic_lds(gemm_nt_bf16x3_kernel, LDS_BYTES, attr_done)) return e;
    dim3 grid((g.M / BM) * (g.N / BN), 1, g.batch * g.splitK);
    hipLaunchKernelGGL(gemm_nt_bf16x3_kernel, grid, dim3(256), LDS_BYTES, stream, g);
    return (int)hipGetLastError();
}
