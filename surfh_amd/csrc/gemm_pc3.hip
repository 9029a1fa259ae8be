// Split-bf16 NT GEMM (arithmetic: gemm_bf16x3.hip) as a producer / consumer workgroup.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]        (both operands K-contiguous, fp32 in HBM)
//
// One workgroup = 8 waves = one 128 x 256 tile, one workgroup per CU.  A CU's SIMD hosts waves w and w + 4 of the
// workgroup, so the roles are split by wave number:
//   waves 0-3  consumers: fragment reads + 96 MFMAs per K step (32), nothing else -- wave tile 64 x 128 (2 x 4 tiles of
//              32 x 32, 128 accumulator registers);
//   waves 4-7  producers: fp32 tile loads in whole 128-byte lines (issued two K steps ahead), exact 3-way split in registers, ds_write of the
//              bf16 pieces -- their VALU work issues beside the partner's MFMAs instead of in front of them.
// The split pieces of a K step live in LDS as [piece][row][32 k] with 64-byte rows whose four 16-byte chunks are
// XOR-ed with row bits 2..3: ds_write_b64 (producers) and ds_read_b128 (consumers) are both conflict-free without
// padding.  Two stages of 72 KB; one barrier per K step: during step i the producers fill stage (i+1)&1 while the
// consumers read stage i&1.
//
// Accumulation-chain length: every fp32 product enters the accumulator as six bf16 partial products, three of them
// 2^-16 of the leading one.  Once the running sum has grown by a few hundred leading terms, those three fall below half
// an ulp of it and are rounded away; on non-negative operands (the physical regime) that is a bias, measured with
// tools/exp/gemm_bias_main.hip: +3e-8 for chains of <= 1024 k, -1.4e-6 at 1536 k, -7.7e-6 at 3072 k, -1.7e-5 at 12288 k
// (the fp32-input MFMA has none).  Callers therefore keep K / splitK at about 1024 (plan.hip:pick_split).
#include "gemm_f32.h"
#include "lds_attr.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 256, BK = 32;
constexpr int ROWS = BM + BN;                 // rows of one stage: A rows 0..127, B rows 128..383
constexpr int PIECE = ROWS * BK;              // one bf16 piece of one stage (elements): 24 KB
constexpr int STAGE = 3 * PIECE;              // 72 KB
constexpr size_t LDS_BYTES = (size_t)2 * STAGE * sizeof(unsigned short);
constexpr int NLD = ROWS / 32;                // float4 loads per producer thread and K step: 32 rows per pass

__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
    const unsigned u = __float_as_uint(x);
    h = u & 0xFFFF0000u;
    const float r = x - __uint_as_float(h);
    m = __float_as_uint(r) & 0xFFFF0000u;
    l = __float_as_uint(r - __uint_as_float(m));
}
__device__ __forceinline__ unsigned pack2(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__device__ __forceinline__ void store_split(unsigned short *dst, float4 v) {
    unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
    split3(v.x, h0, m0, l0);
    split3(v.y, h1, m1, l1);
    split3(v.z, h2, m2, l2);
    split3(v.w, h3, m3, l3);
    *reinterpret_cast<uint2 *>(dst) = make_uint2(pack2(h0, h1), pack2(h2, h3));
    *reinterpret_cast<uint2 *>(dst + PIECE) = make_uint2(pack2(m0, m1), pack2(m2, m3));
    *reinterpret_cast<uint2 *>(dst + 2 * PIECE) = make_uint2(pack2(l0, l1), pack2(l2, l3));
}

// PRE = 0: fp32 operands, split by the producers; 1: B given as bf16 pieces (constant operand, split once);
// 2: both operands given as pieces -- the producers only move 16-byte chunks
template <int PRE>
__global__ __launch_bounds__(512, 1) void gemm_nt_bf16x3_pc_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Work items are numbered slab-major, then
    // B tile, then A tile, and XCD x takes the contiguous range [x P, (x+1) P): the workgroups that run together on an
    // XCD share a K slab and mostly a B tile, so the operands are fetched into that L2 once instead of once per tile.
    const int tilesM = g.M / BM, tilesN = (g.N + BN - 1) / BN, tiles = tilesM * tilesN;
    const long total = (long)tiles * g.splitK * g.batch, per = (total + 7) / 8;
    const long v = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((long)(blockIdx.x >> 3) >= per || v >= total) return;
    const int t = (int)(v % tiles), z = (int)(v / tiles);
    const int tm = t % tilesM, tn = t / tilesM;
    const int b = z / g.splitK, sk = z % g.splitK;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Kper = g.K / g.splitK, kbeg = sk * Kper, nk = Kper / BK;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int t = tid - 256;
        // fp32 rows: 32 per pass, 8 consecutive lanes read the 32 k of one row in whole 128-byte lines
        constexpr int NF = PRE == 0 ? ROWS / 32 : PRE == 1 ? BM / 32 : 0;
        // pre-split rows: 64 per pass, 4 consecutive lanes read the 64 bytes of one row of one piece
        constexpr int SROW0 = PRE == 1 ? BM : 0, NSP = PRE == 0 ? 0 : (ROWS - SROW0) / 64, NS = 3 * NSP;
        const int r = t >> 3, c8 = t & 7;
        const int rs = t >> 2, c4 = t & 3;
        const float *fsrc[NF > 0 ? NF : 1];
        int fpos[NF > 0 ? NF : 1];
        const unsigned short *ssrc[NS > 0 ? NS : 1];
        int spos[NS > 0 ? NS : 1];
        if constexpr (NF > 0) {
            const float *Ab = g.A0 + (long)b * g.sA + kbeg + 4 * c8;
            const float *Bb = g.B0 + (long)b * g.sB + kbeg + 4 * c8;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int row = r + 32 * i;                            // stage row: 0..127 A, 128..383 B
                if (row < BM) {
                    fsrc[i] = Ab + (long)(m0 + row) * g.lda;
                } else {
                    int n = n0 + row - BM;
                    n = n < g.N ? n : g.N - 1;                         // ragged last tile: clamp, the columns are not stored
                    fsrc[i] = Bb + (long)n * g.ldb;
                }
                // LDS position (elements): row * 32 + (16-byte chunk ^ row bits 2..3) * 8 + (c8 & 1) * 4
                fpos[i] = row * BK + (((c8 >> 1) ^ ((row >> 2) & 3)) * 8) + (c8 & 1) * 4;
            }
        }
        if constexpr (NS > 0) {
#pragma unroll
            for (int p = 0; p < NSP; ++p) {
                const int row = SROW0 + rs + 64 * p;
                const unsigned short *base;
                long pl;
                if (row < BM) {
                    base = g.A3 + (long)b * g.sA + (long)(m0 + row) * g.lda + kbeg + 8 * c4;
                    pl = g.pA3;
                } else {
                    int n = n0 + row - BM;
                    n = n < g.N ? n : g.N - 1;
                    base = g.B3 + (long)b * g.sB + (long)n * g.ldb + kbeg + 8 * c4;
                    pl = g.pB3;
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    ssrc[3 * p + q] = base + q * pl;
                    spos[3 * p + q] = q * PIECE + row * BK + ((c4 ^ ((row >> 2) & 3)) * 8);
                }
            }
        }
        float4 fcur[NF > 0 ? NF : 1], fnxt[NF > 0 ? NF : 1];
        u32x4 scur[NS > 0 ? NS : 1], snxt[NS > 0 ? NS : 1];
#define PC_LOAD(kt_, f_, s_)                                                                                              \
    {                                                                                                                     \
        if constexpr (NF > 0) { _Pragma("unroll") for (int i = 0; i < NF; ++i) f_[i] = *reinterpret_cast<const float4 *>(fsrc[i] + (kt_) * BK); } \
        if constexpr (NS > 0) { _Pragma("unroll") for (int i = 0; i < NS; ++i) s_[i] = *reinterpret_cast<const u32x4 *>(ssrc[i] + (kt_) * BK); }  \
    }
#define PC_STORE(st_, f_, s_)                                                                                             \
    {                                                                                                                     \
        unsigned short *base = lds + (st_) * STAGE;                                                                       \
        if constexpr (NF > 0) { _Pragma("unroll") for (int i = 0; i < NF; ++i) store_split(base + fpos[i], f_[i]); }      \
        if constexpr (NS > 0) { _Pragma("unroll") for (int i = 0; i < NS; ++i) *reinterpret_cast<u32x4 *>(base + spos[i]) = s_[i]; } \
    }
        // raw tiles are loaded two K steps ahead of the step the consumers work on (cur: kt+1, nxt: kt+2)
        PC_LOAD(0, fcur, scur);
        if (nk > 1) PC_LOAD(1, fnxt, snxt);
        PC_STORE(0, fcur, scur);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if constexpr (NF > 0) {
#pragma unroll
                for (int i = 0; i < NF; ++i) fcur[i] = fnxt[i];
            }
            if constexpr (NS > 0) {
#pragma unroll
                for (int i = 0; i < NS; ++i) scur[i] = snxt[i];
            }
            if (kt + 2 < nk) PC_LOAD(kt + 2, fnxt, snxt);
            if (kt + 1 < nk) PC_STORE((kt + 1) & 1, fcur, scur);   // into the stage the consumers do not read
            __syncthreads();
        }
#undef PC_LOAD
#undef PC_STORE
        return;
    }

    // ---------------------------------------------------------------------- consumers
    __builtin_amdgcn_s_setprio(1);             // the matrix stream is the critical path of the SIMD it shares with a producer
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;   // 64-row half of A, 128-row half of B
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragment addresses (elements) inside a piece: row * 32 + ((2 s + h) ^ row bits 2..3) * 8 for MFMA k-step s of the K step
    const int sw = (l31 >> 2) & 3;
    const int fa = (wm * 64 + l31) * BK;
    const int fb = (BM + wn * 128 + l31) * BK;
    // output addressing: uniform base + one 32-bit lane offset, rows reached by running increments
    char *Cb = reinterpret_cast<char *>(g.C + (long)b * g.sC + (long)sk * g.sCsplit + (long)m0 * g.ldc + n0);
    const unsigned ldc4 = (unsigned)(g.ldc * 4);
    const unsigned cbase = (unsigned)(wm * 64 + 4 * h) * ldc4 + (unsigned)(wn * 128 + l31) * 4u;
#define PC_FLUSH()                                                                                      \
    {                                                                                                   \
        unsigned o_ = cbase;                                                                            \
        asm volatile("" : "+v"(o_));                                                                    \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int r = 0; r < 16; ++r) {  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                               \
                if (n0 + wn * 128 + j * 32 + l31 < g.N) *reinterpret_cast<float *>(Cb + (o_ + 128u * j)) = acc[i][j][r]; \
            o_ += ((r & 3) == 3) ? 5u * ldc4 : ldc4;                                                    \
        }                                                                                               \
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned short *st = lds + (kt & 1) * STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int ch = ((2 * s2 + h) ^ sw) * 8;
            bf16x8 a[2][3], bq[4][3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][q] = *reinterpret_cast<const bf16x8 *>(st + q * PIECE + fa + i * 32 * BK + ch);
#pragma unroll
                for (int j = 0; j < 4; ++j) bq[j][q] = *reinterpret_cast<const bf16x8 *>(st + q * PIECE + fb + j * 32 * BK + ch);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x16 c = acc[i][j];
                    // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bq[j][1], c, 0, 0, 0);   // m*m
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bq[j][2], c, 0, 0, 0);   // h*l
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], bq[j][0], c, 0, 0, 0);   // l*h
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bq[j][1], c, 0, 0, 0);   // h*m
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bq[j][0], c, 0, 0, 0);   // m*h
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bq[j][0], c, 0, 0, 0);   // h*h
                    acc[i][j] = c;
                }
        }
        __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);
    PC_FLUSH()
#undef PC_FLUSH
}



__global__ __launch_bounds__(256) void split3_kernel(const float *__restrict__ src, unsigned short *__restrict__ dst, long n4, long plane) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(src)[i];
        unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
        split3(v.x, h0, m0, l0);
        split3(v.y, h1, m1, l1);
        split3(v.z, h2, m2, l2);
        split3(v.w, h3, m3, l3);
        *reinterpret_cast<uint2 *>(dst + 4 * i) = make_uint2(pack2(h0, h1), pack2(h2, h3));
        *reinterpret_cast<uint2 *>(dst + plane + 4 * i) = make_uint2(pack2(m0, m1), pack2(m2, m3));
        *reinterpret_cast<uint2 *>(dst + 2 * plane + 4 * i) = make_uint2(pack2(l0, l1), pack2(l2, l3));
    }
}

}  // namespace

int launch_split3(hipStream_t stream, const float *src, unsigned short *dst3, long n, long plane) {
    if (n % 4 || plane % 4) return (int)hipErrorInvalidValue;
    const long n4 = n / 4;
    long nb = (n4 + 255) / 256;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)(nb > 4096 ? 4096 : (nb < 1 ? 1 : nb))), dim3(256), 0, stream, src, dst3, n4, plane);
    return (int)hipGetLastError();
}

// C[M][N] = A[M][K] * B[N][K]^T ; M multiple of 128, N of 128 (a ragged last 256-column tile is handled), K of 32*splitK.
int launch_gemm_nt_bf16x3_pc(hipStream_t stream, const GemmArgs &g) {
    if (g.M % BM || g.N % 128 || g.K % (BK * g.splitK) || g.splitK < 1 || g.batch < 1 || g.accumulate || g.lda % 4 || g.ldb % 4)
        return (int)hipErrorInvalidValue;
    if ((g.A3 && !g.B3) || (g.A3 && (g.pA3 % 8 || g.lda % 8)) || (g.B3 && (g.pB3 % 8 || g.ldb % 8))) return (int)hipErrorInvalidValue;
    const int pre = g.A3 ? 2 : g.B3 ? 1 : 0;
    if ((double)(BM + 1) * (double)g.ldc * 4.0 >= 2147483648.0) return (int)hipErrorInvalidValue;
    const long total = (long)(g.M / BM) * ((g.N + BN - 1) / BN) * g.batch * g.splitK;
    dim3 grid((unsigned)(8 * ((total + 7) / 8)));
    static unsigned long long attr_done[3] = {0, 0, 0};
    if (pre == 0) {
        if (int e = ensure_dynamic_lds(gemm_nt_bf16x3_pc_kernel<0>, LDS_BYTES, attr_done[0])) return e;
        hipLaunchKernelGGL(gemm_nt_bf16x3_pc_kernel<0>, grid, dim3(512), LDS_BYTES, stream, g);
    } else if (pre == 1) {
        if (int e = ensure_dynamic_lds(gemm_nt_bf16x3_pc_kernel<1>, LDS_BYTES, attr_done[1])) return e;
        hipLaunchKernelGGL(gemm_nt_bf16x3_pc_kernel<1>, grid, dim3(512), LDS_BYTES, stream, g);
    } else {
        if (int e = ensure_dynamic_lds(gemm_nt_bf16x3_pc_kernel<2>, LDS_BYTES, attr_done[2])) return e;
        hipLaunchKernelGGL(gemm_nt_bf16x3_pc_kernel<2>, grid, dim3(512), LDS_BYTES, stream, g);
    }
    return (int)hipGetLastError();
}
