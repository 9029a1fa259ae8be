// Two-piece fp16 NT GEMM, all-consumer form: C[m][n] = sum_k A[m][k] * B[n][k], both operands given as their fp16 pieces
// (arithmetic of gemm_pc16.hip).  With the matrix-core work halved, the 128 x 256 producer / consumer tile is bound by
// operand delivery (48 KB per K step = 60 B/clk per CU, the width of the L1 fill path).  Here one workgroup = 8 waves = one
// 256 x 256 tile (wave tile 64 x 128, 4 x 2 waves): 64 KB of operands per K step for twice the flops, delivered by LDS-DMA
// (global_load_lds, no registers, no VALU) into the same XOR-swizzled [piece][row][32 k] image, two 64 KB stages.
#include "gemm_f32.h"
#include "lds_attr.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int ROWS = BM + BN;                 // stage rows: A 0..255, B 256..511
constexpr int PIECE = ROWS * BK;              // one fp16 piece of one stage (elements): 32 KB
constexpr int STAGE = 2 * PIECE;              // 64 KB
constexpr size_t LDS_BYTES = (size_t)2 * STAGE * sizeof(unsigned short);

// power of two that brings `amax` to [2^13, 2^14); 1 for an all-zero row (the same function as in gemm_pc16.hip)
__device__ __forceinline__ float f16x2_scale_of(float amax) {
    if (!(amax > 0.f)) return 1.f;
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127;
    int s = e - 13;
    s = s < -126 ? -126 : (s > 127 ? 127 : s);
    return __uint_as_float((unsigned)(s + 127) << 23);
}

// dst[q*plane + row*ld + k] = fp16 piece q of src[row*ld + k] / scale(rowmax[row]), round to nearest; ld multiple of 4
__global__ __launch_bounds__(256) void split_rows2h_kernel(const float *__restrict__ src, const unsigned *__restrict__ rowmax,
                                                           unsigned short *__restrict__ dst, int ld, long plane) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int row = blockIdx.y;
    const float inv = 1.f / f16x2_scale_of(__uint_as_float(rowmax[row]));
    const int k4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (k4 >= ld) return;
    const long o = (long)row * ld + k4;
    const float4 v = *reinterpret_cast<const float4 *>(src + o);
    const float x0 = v.x * inv, x1 = v.y * inv, x2 = v.z * inv, x3 = v.w * inv;
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1, h2 = (_Float16)x2, h3 = (_Float16)x3;
    f16x4 h = {h0, h1, h2, h3};
    f16x4 l = {(_Float16)(x0 - (float)h0), (_Float16)(x1 - (float)h1), (_Float16)(x2 - (float)h2), (_Float16)(x3 - (float)h3)};
    *reinterpret_cast<f16x4 *>(dst + o) = h;
    *reinterpret_cast<f16x4 *>(dst + plane + o) = l;
}

__global__ __launch_bounds__(512, 1) void gemm_nt_f16x2_cc_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN, tiles = tilesM * tilesN;
    const long total = (long)tiles * g.splitK * g.batch, per = (total + 7) / 8;
    const long v = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((long)(blockIdx.x >> 3) >= per || v >= total) return;
    const int t = (int)(v % tiles), z = (int)(v / tiles);
    const int tm = t % tilesM, tn = t / tilesM;
    const int b = z / g.splitK, sk = z % g.splitK;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Kper = g.K / g.splitK, kbeg = sk * Kper, nk = Kper / BK;

    // DMA: one wave instruction moves 64 x 16 B = 16 rows of one piece (row = lane >> 2, LDS chunk = lane & 3, which holds
    // the global 16-byte chunk (lane & 3) ^ ((row >> 2) & 3) of that row).  A stage has 2 pieces x 512 rows = 64 such
    // groups; wave w moves groups w, w + 8, ... (8 per K step).
    const unsigned short *gsrc[8];
    int gdst[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int grp = wave + 8 * i;              // 0..63: piece = grp >> 5, 16-row group inside the piece = grp & 31
        const int q = grp >> 5, row = (grp & 31) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        const unsigned short *base;
        if (row < BM) {
            int m = m0 + row;
            m = m < g.M ? m : g.M - 1;             // ragged last tile: clamp, the rows are not stored
            base = g.A3 + (long)b * g.sA + q * g.pA3 + (long)m * g.lda;
        } else {
            int n = n0 + row - BM;
            n = n < g.N ? n : g.N - 1;
            base = g.B16 + (long)b * g.sB + q * g.pB16 + (long)n * g.ldb;
        }
        gsrc[i] = base + kbeg + 8 * chunk;
        gdst[i] = q * PIECE + (grp & 31) * 16 * BK;            // wave-uniform LDS base of the group (lane-linear behind it)
    }
#define CC_DMA(kt_, st_)                                                                                             \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc[i] + (kt_) * BK),  \
                                             (__attribute__((address_space(3))) void *)(lds + (st_) * STAGE + gdst[i]), 16, 0, 0); \
    }

    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;       // 4 x 2 waves, wave tile 64 x 128
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

    // ragged last row tile: the waves whose 64 rows lie beyond M only move data.  Waves w and w + 4 share a SIMD and differ
    // by two 64-row blocks, so a half-empty tile costs half the matrix-core time.
    const bool active = m0 + wm * 64 < g.M;
    // Block-scaled A (g.bscale): the scale of a row changes from one K segment to the next (the same boundaries for every
    // row).  The accumulators are kept in units of the current segment's scale and rescaled -- by an exact power of two --
    // when a boundary is crossed.
    const float *bs = g.bscale;                    // batch 1 only (checked by the launcher)
    int seg = 0, seg_end = 1 << 30;                // current segment and the k at which it ends
    if (bs) {
        const int col = kbeg / g.segLinP, in = kbeg % g.segLinP;
        seg = col * g.segChunks + in / 1024;
        const int e1 = col * g.segLinP + (in / 1024 + 1) * 1024, e2 = (col + 1) * g.segLinP;
        seg_end = e1 < e2 ? e1 : e2;
    }
    CC_DMA(0, 0);
    __syncthreads();                               // drains the DMA (vmcnt(0)) of every wave
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) CC_DMA(kt + 1, (kt + 1) & 1);   // the other stage was last read before the previous barrier
        const unsigned short *st = lds + (kt & 1) * STAGE;
        if (bs && kbeg + kt * BK >= seg_end) {     // workgroup-uniform
            const int k = kbeg + kt * BK, col = k / g.segLinP, in = k % g.segLinP;
            const int nseg = col * g.segChunks + in / 1024;
            const int e1 = col * g.segLinP + (in / 1024 + 1) * 1024, e2 = (col + 1) * g.segLinP;
            seg_end = e1 < e2 ? e1 : e2;
            if (active) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        row = row < g.M ? row : g.M - 1;
                        const float ratio = bs[(long)seg * g.M + row] / bs[(long)nseg * g.M + row];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j][r] *= ratio;
                    }
            }
            seg = nseg;
        }
        if (active)
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
        __syncthreads();                           // next stage complete (vmcnt(0)), this one free
    }
#undef CC_DMA
    if (active) {
        char *Cb = reinterpret_cast<char *>(g.C + (long)b * g.sC + (long)sk * g.sCsplit + (long)m0 * g.ldc + n0);
        const unsigned ldc4 = (unsigned)(g.ldc * 4);
        unsigned o_ = (unsigned)(wm * 64 + 4 * h) * ldc4 + (unsigned)(wn * 128 + l31) * 4u;
        // undo both operand scales (powers of two: exact): A was split row by row with the scale of its row maximum
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float rsc[16];                 // the 16 row scales of this half requested together
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                row = row < g.M ? row : g.M - 1;
                rsc[r] = bs ? bs[(long)seg * g.M + row] : f16x2_scale_of(__uint_as_float(g.amax[(long)b * g.M + row]));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float sc = rsc[r] * g.sB16;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (row < g.M && n0 + wn * 128 + j * 32 + l31 < g.N) *reinterpret_cast<float *>(Cb + (o_ + 128u * j)) = acc[i][j][r] * sc;
                o_ += ((r & 3) == 3) ? 5u * ldc4 : ldc4;
            }
        }
    }
}

}  // namespace

int launch_split_rows2h(hipStream_t stream, const float *src, const unsigned *rowmax, unsigned short *dst2, int rows, int ld, long plane) {
    if (ld % 4 || plane % 4 || rows < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(split_rows2h_kernel, dim3((ld / 4 + 255) / 256, rows), dim3(256), 0, stream, src, rowmax, dst2, ld, plane);
    return (int)hipGetLastError();
}

// A as fp16 pieces A3[q*pA3 + m*lda + k] of A[m][k] / scale(amax[m]) (launch_split_rows2h), B as pieces B16 of B / sB16
int launch_gemm_nt_f16x2_cc(hipStream_t stream, const GemmArgs &g) {
    if (g.M % 64 || g.N % 128 || g.K % (BK * g.splitK) || g.splitK < 1 || g.batch < 1 || g.accumulate || g.lda % 8 || g.ldb % 8 || !g.A3 ||
        !g.B16 || g.pA3 % 8 || g.pB16 % 8 || (!g.amax && !g.bscale) || !(g.sB16 > 0.f))
        return (int)hipErrorInvalidValue;
    if (g.bscale && (g.batch != 1 || g.segLinP < BK || g.segLinP % BK || g.segChunks != (g.segLinP + 1023) / 1024 || g.K % g.segLinP))
        return (int)hipErrorInvalidValue;
    if ((double)(BM + 1) * (double)g.ldc * 4.0 >= 2147483648.0) return (int)hipErrorInvalidValue;
    const long total = (long)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * g.batch * g.splitK;
    dim3 grid((unsigned)(8 * ((total + 7) / 8)));
    static unsigned long long attr_done = 0;
    if (int e = ensure_dynamic_lds(gemm_nt_f16x2_cc_kernel, LDS_BYTES, attr_done)) return e;
    hipLaunchKernelGGL(gemm_nt_f16x2_cc_kernel, grid, dim3(512), LDS_BYTES, stream, g);
    return (int)hipGetLastError();
}
