// Split-bf16 NT GEMM (arithmetic: gemm_bf16x3.hip) as a producer / consumer workgroup.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]        (both operands K-contiguous, fp32 in HBM)
//
// One workgroup = 8 waves = one 128 x 256 tile, one workgroup per CU.  A CU's SIMD hosts waves w and w + 4 of the
// workgroup, so the roles are split by wave number:
//   waves 0-3  consumers: fragment reads + 48 MFMAs per K step (16), nothing else -- wave tile 64 x 128 (2 x 4 tiles of
//              32 x 32, 128 accumulator registers);
//   waves 4-7  producers: fp32 tile loads (issued three K steps ahead), exact 3-way split in registers, ds_write of the
//              bf16 pieces -- their VALU work issues beside the partner's MFMAs instead of in front of them.
// The split pieces of a K step live in LDS as [piece][row][16 k] with 32-byte rows, the two 16-byte halves of a row
// swapped on rows with bit 3 set: ds_write_b64 (producers) and ds_read_b128 (consumers) are both conflict-free without
// padding.  Two stages of 36 KB; one barrier per K step: during step i the producers fill stage (i+1)&1 while the
// consumers read stage i&1.
#include "../../surfh_amd/csrc/gemm_f32.h"
#ifndef PC_EXP
#define PC_EXP 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BM = 128, BN = 256, BK = 16;
constexpr int ROWS = BM + BN;                 // rows of one stage: A rows 0..127, B rows 128..383
constexpr int PIECE = ROWS * BK;              // one bf16 piece of one stage (elements): 12 KB
constexpr int STAGE = 3 * PIECE;              // 36 KB
constexpr size_t LDS_BYTES = (size_t)2 * STAGE * sizeof(unsigned short);

__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
    const unsigned u = __float_as_uint(x);
#if PC_EXP == 2
    h = u; m = u; l = u; return;
#endif
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

__global__ __launch_bounds__(512, 1) void gemm_nt_bf16x3_pc_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesN = (g.N + BN - 1) / BN;
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
    const int b = blockIdx.z / g.splitK, sk = blockIdx.z % g.splitK;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Kper = g.K / g.splitK, kbeg = sk * Kper, nk = Kper / BK;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int t = tid - 256;
        const int r = t >> 2, c4 = t & 3;                         // row group (0..63), float4 index inside the 16 k
        const float *Ab = g.A0 + (long)b * g.sA + kbeg + 4 * c4;
        const float *Bb = g.B0 + (long)b * g.sB + kbeg + 4 * c4;
        // global row pointers: 2 A rows (r, r+64), 4 B rows (r, r+64, r+128, r+192; clamped on a ragged last tile)
        const float *pa0 = Ab + (long)(m0 + r) * g.lda, *pa1 = Ab + (long)(m0 + r + 64) * g.lda;
        const float *pb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = n0 + r + 64 * i;
            row = row < g.N ? row : g.N - 1;
            pb[i] = Bb + (long)row * g.ldb;
        }
        // LDS positions: row * 16 elements + (k half ^ row bit 3) * 8 + (c4 & 1) * 4
        auto lpos = [&](int row) { return row * BK + (((c4 >> 1) ^ ((row >> 3) & 1)) * 8) + (c4 & 1) * 4; };
        const int la0 = lpos(r), la1 = lpos(r + 64);
        int lb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lb[i] = lpos(BM + r + 64 * i);

        float4 c0, c1, c2, c3, c4v, c5, n0v, n1, n2, n3, n4, n5, f0, f1, f2, f3, f4, f5;
#define PC_LOAD(kt_, x0_, x1_, x2_, x3_, x4_, x5_)                                      \
    {                                                                                   \
        const int ko = (kt_) * BK;                                                      \
        x0_ = *reinterpret_cast<const float4 *>(pa0 + ko);                              \
        x1_ = *reinterpret_cast<const float4 *>(pa1 + ko);                              \
        x2_ = *reinterpret_cast<const float4 *>(pb[0] + ko);                            \
        x3_ = *reinterpret_cast<const float4 *>(pb[1] + ko);                            \
        x4_ = *reinterpret_cast<const float4 *>(pb[2] + ko);                            \
        x5_ = *reinterpret_cast<const float4 *>(pb[3] + ko);                            \
    }
#define PC_STORE(st_, x0_, x1_, x2_, x3_, x4_, x5_)                                     \
    {                                                                                   \
        unsigned short *base = lds + (st_) * STAGE;                                     \
        store_split(base + la0, x0_);                                                   \
        store_split(base + la1, x1_);                                                   \
        store_split(base + lb[0], x2_);                                                 \
        store_split(base + lb[1], x3_);                                                 \
        store_split(base + lb[2], x4_);                                                 \
        store_split(base + lb[3], x5_);                                                 \
    }
        // raw tiles are loaded three K steps ahead of the step the consumers work on (c: kt+1, n: kt+2, f: kt+3)
        PC_LOAD(0, c0, c1, c2, c3, c4v, c5);
        if (nk > 1) PC_LOAD(1, n0v, n1, n2, n3, n4, n5);
        if (nk > 2) PC_LOAD(2, f0, f1, f2, f3, f4, f5);
        PC_STORE(0, c0, c1, c2, c3, c4v, c5);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
#if PC_EXP == 5
            __syncthreads(); continue;
#endif
            c0 = n0v; c1 = n1; c2 = n2; c3 = n3; c4v = n4; c5 = n5;
            n0v = f0; n1 = f1; n2 = f2; n3 = f3; n4 = f4; n5 = f5;
#if PC_EXP != 1
            if (kt + 3 < nk) PC_LOAD(kt + 3, f0, f1, f2, f3, f4, f5);
#endif
#if PC_EXP == 3
            if (c0.x == 1234.5f) PC_STORE((kt + 1) & 1, c0, c1, c2, c3, c4v, c5);
#else
            if (kt + 1 < nk) PC_STORE((kt + 1) & 1, c0, c1, c2, c3, c4v, c5);
#endif
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
    // fragment addresses (elements) inside a piece: row * 16 + (h ^ row bit 3) * 8; (row >> 3) & 1 == (l31 >> 3) & 1 for all tiles
    const int hs = (h ^ ((l31 >> 3) & 1)) * 8;
    const int fa = (wm * 64 + l31) * BK + hs;
    const int fb = (BM + wn * 128 + l31) * BK + hs;
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned short *st = lds + (kt & 1) * STAGE;
        bf16x8 a[2][3], bq[4][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i][q] = *reinterpret_cast<const bf16x8 *>(st + q * PIECE + fa + i * 32 * BK);
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[j][q] = *reinterpret_cast<const bf16x8 *>(st + q * PIECE + fb + j * 32 * BK);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x16 c = acc[i][j];
#if PC_EXP == 4
                c[0] += (float)a[i][0][0] + (float)a[i][1][1] + (float)a[i][2][2] + (float)bq[j][0][0] + (float)bq[j][1][1] + (float)bq[j][2][2];
                acc[i][j] = c;
                continue;
#endif
                // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bq[j][1], c, 0, 0, 0);   // m*m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bq[j][2], c, 0, 0, 0);   // h*l
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], bq[j][0], c, 0, 0, 0);   // l*h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bq[j][1], c, 0, 0, 0);   // h*m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bq[j][0], c, 0, 0, 0);   // m*h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bq[j][0], c, 0, 0, 0);   // h*h
                acc[i][j] = c;
            }
        __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);

    float *Cb = g.C + (long)b * g.sC + (long)sk * g.sCsplit;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 128 + j * 32 + l31;
            if (col >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                Cb[(long)row * g.ldc + col] = acc[i][j][r];
            }
        }
}

}  // namespace

// C[M][N] = A[M][K] * B[N][K]^T ; M multiple of 128, N of 128 (a ragged last 256-column tile is handled), K of 16*splitK.
int launch_gemm_nt_bf16x3_pc(hipStream_t stream, const GemmArgs &g) {
    if (g.M % BM || g.N % 128 || g.K % (BK * g.splitK) || g.splitK < 1 || g.batch < 1 || g.accumulate || g.lda % 4 || g.ldb % 4)
        return (int)hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)gemm_nt_bf16x3_pc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((g.M / BM) * ((g.N + BN - 1) / BN), 1, g.batch * g.splitK);
    hipLaunchKernelGGL(gemm_nt_bf16x3_pc_kernel, grid, dim3(512), LDS_BYTES, stream, g);
    return (int)hipGetLastError();
}
