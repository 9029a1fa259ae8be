// Two-piece fp16 NT GEMM on the matrix cores: C[m][n] = sum_k A[m][k] * B[n][k], both operands given as their fp16 pieces.
//
// Arithmetic.  Every operand is scaled by a power of two so that its largest magnitude sits at 2^14, then cut into two fp16
// values by round-to-nearest:  x / s = h + l + e,  |l| <= 2^-11 |h|,  |e| <= 2^-23 |x / s|  (22 mantissa bits; below 2^-18
// of the largest magnitude the relative precision decreases, the absolute error stays under 2^-39 of the largest
// magnitude).  Three products are kept (lh, hl, hh -- each exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_f16);
// the dropped l*l and the remainders e are 2^-22 relative with random sign.  Measured against float64 the split error is
// 3e-9 (L2) for half the matrix-core work of a three-piece bf16 split; what remains is the fp32 accumulation itself.
//   * B is constant on this path (the spectral PSF): split once at plan creation (launch_split2h), scale sB16 a host constant.
//   * A is data: one scale PER ROW (launch_split_rows2h, amax) or per (row, K segment) (the gather writes the pieces itself,
//     bscale), so that an outlier in one row (a hot detector pixel) does not cost the other rows their precision.
//
// Tile.  One workgroup = 8 consumer waves = one 256 x 256 tile (wave tile 64 x 128, 4 x 2 waves): 64 KB of operands per K
// step of 32, delivered by LDS-DMA (global_load_lds, no registers, no VALU) into an XOR-swizzled [piece][row][32 k] image,
// two 64 KB stages.  (A 128 x 256 producer / consumer tile needs 48 KB per K step for half the flops: 60 B/clk per CU, the
// width of the L1 fill path.)
#include "gemm_f32.h"
#include "lds_attr.h"
#include <cmath>
#include <cstdlib>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int ROWS = BM + BN;                 // stage rows: A 0..255, B 256..511
constexpr int PIECE = ROWS * BK;              // one fp16 piece of one stage (elements): 32 KB
constexpr int STAGE = 2 * PIECE;              // 64 KB
constexpr size_t LDS_BYTES = (size_t)2 * STAGE * sizeof(unsigned short);

// power of two that brings `amax` to [2^13, 2^14); 1 for an all-zero row (the gather uses the same function: f16x2_block_scale, kernels.hip)
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

// The kernel's K loop is hand-scheduled (described in front of the kernel): hipcc puts s_waitcnt lgkmcnt(0) behind every group
// of LDS reads and vmcnt(0) in front of LDS accesses while a DMA is in flight.  Register budget: 128 accumulators + 64
// fragment registers -- the DMA sources are one 32-bit lane offset per 128-row group on scalar bases and the block scales of
// A sit in LDS (K segments of this slab x 256 rows).
#ifndef CC_EXP
#define CC_EXP 0          // experiment mask (tools/exp): 1 no DMA inside the loop, 2 no fragment reads, 4 no MFMAs
#endif
constexpr int CC2_MAXSEG = 64;                                        // K segments one slab may cross (launcher checks)
constexpr size_t LDS2_BYTES = LDS_BYTES + (size_t)CC2_MAXSEG * BM;    // the scales are powers of two: one exponent byte each

// a pointer the compiler keeps in scalar registers (wave-uniform by construction)
__device__ __forceinline__ const char *cc2_uniform(const char *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char *)(((unsigned long long)hi << 32) | lo);
}
// segment of column k of the block-scaled A and the k at which it ends; k is wave-uniform, the results are scalars
__device__ __forceinline__ int cc2_seg_of(int k, int segLinP, int segChunks) {
    const int col = k / segLinP, in = k % segLinP;
    return __builtin_amdgcn_readfirstlane(col * segChunks + in / 1024);
}
__device__ __forceinline__ int cc2_seg_end(int k, int segLinP) {
    const int col = k / segLinP, in = k % segLinP;
    const int e1 = col * segLinP + (in / 1024 + 1) * 1024, e2 = (col + 1) * segLinP;
    return __builtin_amdgcn_readfirstlane(e1 < e2 ? e1 : e2);
}
// old / new for two power-of-two scales (f16x2_scale_of) given by their exponent fields, exact; difference clamped to the normal range
__device__ __forceinline__ float cc2_pow2_ratio(unsigned eo, unsigned en) {
    int e = (int)eo - (int)en + 127;
    e = e < 1 ? 1 : (e > 254 ? 254 : e);
    return __uint_as_float((unsigned)e << 23);
}
// 1 KiB of one piece (16 rows x 64 B) from global memory straight into LDS at byte address `l` (wave-uniform)
__device__ __forceinline__ void cc2_dma(const char *base, unsigned off, unsigned l) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(l) : "memory");
}

// the accumulators are written by assembly MFMAs the compiler knows nothing about: before ordinary code reads them, let the
// last one drain
#define CC_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory")

// ---------------------------------------------------------------------------------------------------------------
// The K loop on v_mfma_f32_16x16x32_f16, as a fixed interleaved stream of volatile inline assembly (the compiler keeps
// the order and adds no waits of its own; LDS reads return in order, vmcnt is one in-order counter):
//   * a K step is one MFMA deep (k = 32): a lane holds 8 consecutive k of one row (row = lane & 15, k chunk = lane >> 4),
//     and the LDS image is swizzled for that read pattern: chunk c of row r at position c ^ (-(r >> 2) & 3) -- conflict-free
//     ds_read_b128, no padding; the DMA writes the image lane-linearly, so the swizzle is a permutation of the 16-byte
//     chunks each lane fetches from its row's 64-byte segment;
//   * wave tile 64 x 128 = 4 x 8 blocks of 16 x 16, accumulator block = 4 registers (rows 4 (lane >> 4) + r, column lane & 15);
//   * stream of a K step: four units of 24 MFMAs (unit u = column blocks 2u, 2u + 1; inside a unit row block by row block,
//     products l*h, h*l, h*h per block).  Behind MFMAs 0..3 of a unit the B fragments of the next unit; in unit 3 the A
//     fragments of the next K step, each row block's pair right behind the last MFMA that uses the old ones (the registers are
//     free then), and the eight DMA pieces of K step kt + 2 into the stage just left;
//   * one barrier per K step, in front of unit 3: every wave then holds the last fragments of stage kt & 1 in registers and
//     K step kt + 1 has landed (it is each wave's newest DMA).  One instruction stream for every kt: in the last two steps the
//     DMA repeats the last K step into a stage nobody reads any more and the reads fetch fragments nobody uses.
// 128 accumulator + 64 fragment registers.  The round-1 loop (32x32x16 MFMAs, compiler-scheduled) spent 0.123 ms on the DMA
// alone and 0.142 ms on the MFMAs alone and took their SUM; hand-scheduled it reached the matrix cores' rate on constant
// operands (0.157 ms, 1.5 PFLOP/s issued) and 0.24 ms on random ones: the chip lowers its clock under the bit toggling of real
// data.  The 16x16x32 shape draws less per flop: 0.225 ms on random operands (0.170 on constant ones) -- tools/exp/cc_main.hip.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// FAR: a K step whose products are kept as h*h only (see "K-step classes" in front of the kernel): one MFMA per block instead of
// three, no l fragments, no l pieces in the DMA
template <int U, int N, bool DMA, bool FAR>
__device__ __forceinline__ void cc16_slot(f32x4v (&acc)[4][8], f16x8 (&A)[4][2], const f16x8 (&Bc)[2][2], f16x8 (&Bn)[2][2], unsigned ran,
                                          unsigned rbn, const char *ka, const char *kb, long pA2, long pB2, const unsigned (&offA)[2],
                                          const unsigned (&offB)[2], unsigned ls) {
    constexpr int i = N / 6, jj = (N / 3) & 1, j = 2 * U + jj, p = N % 3;
    if constexpr (!(CC_EXP & 4) && (!FAR || p == 2))
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(A[i][p == 0 ? 1 : 0]), "v"(Bc[jj][p == 1 ? 1 : 0]));
    if constexpr (N < 4 && !(CC_EXP & 2)) {        // B fragments of the next unit (unit 3: of the next K step's unit 0, other stage)
        constexpr int jn = 2 * ((U + 1) & 3) + (N >> 1), q = N & 1;
        if constexpr (!FAR || q == 0)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(Bn[N >> 1][q]) : "v"(rbn), "n"(q * PIECE * 2 + jn * 16 * BK * 2));
    }
    if constexpr (U == 3 && N % 6 == 5 && !(CC_EXP & 2)) {      // row block i is done with its A fragments: the next K step's follow
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[i][0]) : "v"(ran), "n"(i * 16 * BK * 2));
        if constexpr (!FAR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[i][1]) : "v"(ran), "n"(PIECE * 2 + i * 16 * BK * 2));
    }
    constexpr int d = (U == 3 && N % 3 != 2 && N < 12) ? N - N / 3 : -1;       // pieces 0..7 behind MFMAs 0, 1, 3, 4, 6, 7, 9, 10
    if constexpr (DMA && d >= 0 && !(CC_EXP & 1) && (!FAR || d < 4)) {
        constexpr int q = d >> 2, i2 = (d >> 1) & 1;
        if constexpr (d & 1) cc2_dma(kb + q * pB2, offB[i2], ls + (unsigned)((q * PIECE + (BM + i2 * 128) * BK) * 2));
        else cc2_dma(ka + q * pA2, offA[i2], ls + (unsigned)((q * PIECE + i2 * 128 * BK) * 2));
    }
}
template <int U, bool DMA, bool FAR, int N = 0>
__device__ __forceinline__ void cc16_unit(f32x4v (&acc)[4][8], f16x8 (&A)[4][2], const f16x8 (&Bc)[2][2], f16x8 (&Bn)[2][2], unsigned ran,
                                          unsigned rbn, const char *ka, const char *kb, long pA2, long pB2, const unsigned (&offA)[2],
                                          const unsigned (&offB)[2], unsigned ls) {
    if constexpr (N < 24) {
        // unit 0: row block 3's fragments of this K step were requested last in the previous unit 3 (two reads; one in a FAR step),
        // before the read of the K-step list entry and this unit's four (two) B reads: wait for them here
        if constexpr (U == 0 && N == 18 && !(CC_EXP & 2)) {
            if constexpr (FAR) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(A[3][0]), "+v"(A[3][1]));
            else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(A[3][0]), "+v"(A[3][1]));
        }
        cc16_slot<U, N, DMA, FAR>(acc, A, Bc, Bn, ran, rbn, ka, kb, pA2, pB2, offA, offB, ls);
        cc16_unit<U, DMA, FAR, N + 1>(acc, A, Bc, Bn, ran, rbn, ka, kb, pA2, pB2, offA, offB, ls);
    }
}
// wait until at most n_ LDS reads of this wave are outstanding (the B fragments of the next unit are operands)
#define CC16_WAITB(n_, B_)                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(B_[0][0]), "+v"(B_[1][0]), "+v"(B_[0][1]), "+v"(B_[1][1]))

// K-step classes.  The constant operand of this path, the spectral response, decays away from its diagonal: for a tile of 256
// of its rows most K steps hold only the far tails.  The launcher may pass, per tile column, two lists of K steps (klist):
// "near" steps keep the three products, "far" steps only h*h -- a relative error of 2^-10 at worst on terms whose total weight
// the host bounded (plan.hip: build_klist; far steps of a row sum to < 2^-8 of its l1 norm and < 2^-10 of its l2 norm).  A far
// step issues a third of the MFMAs and moves half the operand bytes: 0.44 of the time of a near one (tools/exp/cc_main.hip).
// Without lists every step of the slab is near, in ascending order.  List entry = K step | segment of A's block scales << 16.
constexpr int CC2_MAXLIST = 2048;                                     // K steps of one slab (launcher checks)
constexpr size_t LDS3_BYTES = LDS2_BYTES + (size_t)CC2_MAXLIST * sizeof(int);

// The products of one launch: work items (tile, slab, batch entry) of product i are first[i] .. first[i + 1] - 1 of one list,
// and XCD x takes a contiguous eighth of that list -- the products run side by side, each on the XCDs its items fall to.
// Measured on config 3 (four bands): the adjoint's products, 1.5-1.8 rounds of workgroups each on their own, 0.52 -> 0.41 ms per
// step (0.43 with the products one after the other inside the launch, each spread over all XCDs); the forward's products are
// one full round each and lose the L2 sharing of 252 concurrent workgroups on the same operands: 0.40 -> 0.51 ms -- they
// stay one launch per channel.
struct GemmGroup {
    GemmArgs g[GEMM_GROUP_MAX];
    long first[GEMM_GROUP_MAX + 1];
    int n;
};

__global__ __launch_bounds__(512, 1) void gemm_nt_f16x2_cc_kernel(GemmGroup grp) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    // XCD x takes a contiguous range of the launch's work items (workgroups sharing an L2 share their operands)
    const long gtotal = grp.first[grp.n], per = (gtotal + 7) / 8;
    const long vg = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((long)(blockIdx.x >> 3) >= per || vg >= gtotal) return;
    int gi = 0;
    while (gi + 1 < grp.n && vg >= grp.first[gi + 1]) ++gi;
    const long v = vg - grp.first[gi];
    const GemmArgs g = grp.g[gi];
    unsigned char *sctab = reinterpret_cast<unsigned char *>(lds + 2 * STAGE);     // [segment][256 rows] exponent fields of A's block scales
    int *klds = reinterpret_cast<int *>(sctab + CC2_MAXSEG * BM);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tile columns: 256 consecutive rows of B, or (permP > 0) Q = 256 / permP consecutive rows of each of permP neighbouring columns
    const int Q = g.permP ? BN / g.permP : BN, tilesL = g.permP ? g.permLin / Q : 0, ncol = g.permP ? g.N / g.permLin : 0;
    const int tilesM = (g.M + BM - 1) / BM, tilesN = g.permP ? (ncol + g.permP - 1) / g.permP * tilesL : (g.N + BN - 1) / BN, tiles = tilesM * tilesN;
    // row of B (= column of C) of tile-local column v of tile column tn_; -1: none (a column group beyond the last)
    auto bcol = [&](int tn_, int v) __attribute__((always_inline)) {
        if (!g.permP) { const int n = tn_ * BN + v; return n < g.N ? n : -1; }
        const int c = (tn_ / tilesL) * g.permP + v / Q;
        return c < ncol ? c * g.permLin + (tn_ % tilesL) * Q + v % Q : -1;
    };
    const int t = (int)(v % tiles), z = (int)(v / tiles);
    const int tm = t % tilesM, tn = t / tilesM;
    const int b = z / g.splitK, sk = z % g.splitK;
    const int m0 = tm * BM;
    const float *bs = g.bscale;

    // this slab's K steps: near ones first, then the far ones, each in ascending order
    int nnear, nfar, seg0 = 0, nseg = 0;
    if (g.klist) {
        const int *kl = g.klist + (long)tn * g.klistStride;
        const int NN = kl[0], NF = kl[1];
        const int a0 = (int)((long)NN * sk / g.splitK), a1 = (int)((long)NN * (sk + 1) / g.splitK);
        const int f0 = (int)((long)NF * sk / g.splitK), f1 = (int)((long)NF * (sk + 1) / g.splitK);
        nnear = __builtin_amdgcn_readfirstlane(a1 - a0);
        nfar = __builtin_amdgcn_readfirstlane(f1 - f0);
        for (int e = tid; e < nnear; e += 512) klds[e] = kl[2 + a0 + e];
        for (int e = tid; e < nfar; e += 512) klds[nnear + e] = kl[2 + NN + f0 + e];
        if (bs) nseg = (g.K / g.segLinP) * g.segChunks;               // every segment of A (launcher: <= CC2_MAXSEG)
    } else {
        const int Kper = g.K / g.splitK, kbeg = sk * Kper;
        nnear = Kper / BK;
        nfar = 0;
        if (bs) {
            seg0 = cc2_seg_of(kbeg, g.segLinP, g.segChunks);
            nseg = cc2_seg_of(kbeg + Kper - 1, g.segLinP, g.segChunks) - seg0 + 1;
        }
        for (int e = tid; e < nnear; e += 512) {
            const int k = kbeg + e * BK;
            klds[e] = (k / BK) | (bs ? ((k / g.segLinP) * g.segChunks + (k % g.segLinP) / 1024) << 16 : 0);
        }
    }
    for (int e = tid; e < nseg * BM; e += 512) {
        int row = m0 + (e & (BM - 1));
        row = row < g.M ? row : g.M - 1;
        sctab[e] = (unsigned char)(__float_as_uint(bs[(long)(seg0 + (e >> 8)) * g.M + row]) >> 23);
    }

    // DMA: LDS slot lane & 3 of row lane >> 2 (of a 16-row group) holds the global chunk (lane & 3) ^ (-(lane >> 4) & 3)
    unsigned offA[2], offB[2];
    {
        const int chunk = (lane & 3) ^ ((-(lane >> 4)) & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = m0 + (wave + 8 * i) * 16 + (lane >> 2);
            m = m < g.M ? m : g.M - 1;
            offA[i] = (unsigned)(((long)m * g.lda + 8 * chunk) * 2);
            int n = bcol(tn, (wave + 8 * i) * 16 + (lane >> 2));
            n = n >= 0 ? n : g.N - 1;
            offB[i] = (unsigned)(((long)n * g.ldb + 8 * chunk) * 2);
        }
    }
    const char *baseA = cc2_uniform(reinterpret_cast<const char *>(g.A3 + (long)b * g.sA));
    const char *baseB = cc2_uniform(reinterpret_cast<const char *>(g.B16 + (long)b * g.sB));
    const long pA2 = g.pA3 * 2, pB2 = g.pB16 * 2;
    const unsigned ldsw = (unsigned)(size_t)lds + (unsigned)(wave * 16 * BK * 2);
    const unsigned kaddr = (unsigned)(size_t)klds;
    const int l15 = lane & 15, kc = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;       // 4 x 2 waves, wave tile 64 x 128
    f32x4v acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // fragment byte addresses in stage 0 (the stage is toggled by XOR with its size)
    const unsigned ra0 = (unsigned)(size_t)lds + (unsigned)(((wm * 64 + l15) * BK + ((kc ^ ((-(l15 >> 2)) & 3)) * 8)) * 2);
    const unsigned fbd = (unsigned)((BM + wn * 128 - wm * 64) * BK * 2);
    const bool active = m0 + wm * 64 < g.M;
    const int srow = wm * 64 + 4 * kc;             // first of the lane's accumulator rows inside the tile (row block 0)
    int seg = -1;                                  // segment whose scales the accumulators carry (-1: none yet, they are zero)

    f16x8 A[4][2], Bx[2][2], By[2][2];             // A [row block][piece] of the current K step; B [column block][piece] of the even / odd unit
#define CC_RD(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(off_))
    // the eight (FAR: four) DMA pieces of K step ks_ into stage st_
#define CC_DMA(FAR_, ks_, st_)                                                                                         \
    {                                                                                                                  \
        const char *ka = baseA + (long)(ks_) * (BK * 2), *kb = baseB + (long)(ks_) * (BK * 2);                         \
        const unsigned ls = ldsw + stofs((FAR_), (st_));                                                               \
        _Pragma("unroll") for (int q = 0; q < ((FAR_) ? 1 : 2); ++q) {                                                 \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                            \
                cc2_dma(ka + q * pA2, offA[i], ls + (q * PIECE + i * 128 * BK) * 2);                                   \
                cc2_dma(kb + q * pB2, offB[i], ls + (q * PIECE + (BM + i * 128) * BK) * 2);                            \
            }                                                                                                          \
        }                                                                                                              \
    }
    // byte offset of ring stage s: two stages of both pieces, or (far pass: h pieces only) four stages in the same 128 KB --
    // far stage s takes the place of piece s & 1 of stage s >> 1
    auto stofs = [](bool far, int s) __attribute__((always_inline)) {
        return far ? (unsigned)((s >> 1) * (STAGE * 2) + (s & 1) * (PIECE * 2)) : (unsigned)(s * (STAGE * 2));
    };
    __syncthreads();                               // lists and scales are in LDS

    // one pass over n list entries from klds[lb]
    // A ring of D stages: the DMA of K step kt + D is issued in step kt (unit 3, into the stage step kt has just left) and must
    // have landed at the barrier of step kt + D - 1.  A near step lasts about 2.5 us, a far one 1 us -- less than a trip to
    // memory, so the far pass runs four stages deep.
    auto kloop = [&](auto far_tag, const int lb, const int n) __attribute__((always_inline)) {
        constexpr bool FAR = decltype(far_tag)::value;
        constexpr int D = FAR ? 4 : 2;             // ring depth; a step's DMA group is 4 (far) or 8 instructions
        if (n <= 0) return;                        // workgroup-uniform
        int e[D];
#pragma unroll
        for (int s = 0; s < D; ++s) {
            e[s] = __builtin_amdgcn_readfirstlane(klds[lb + (s < n ? s : n - 1)]);
            CC_DMA(FAR, e[s] & 0xFFFF, s);
        }
        if (FAR) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        if (!active) {
            for (int kt = 0; kt < n; ++kt) {
                if (FAR) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                const int ed = __builtin_amdgcn_readfirstlane(klds[lb + (kt + D < n ? kt + D : n - 1)]);
                CC_DMA(FAR, ed & 0xFFFF, kt & (D - 1));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        unsigned ra = ra0, rb = ra0 + fbd;
        if (!(CC_EXP & 2)) {                       // fragments of stage 0: A, B of unit 0
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                CC_RD(A[i][0], ra, 0 + 0);
                if (!FAR) CC_RD(A[i][1], ra, PIECE * 2);
                ra += 16 * BK * 2;
            }
            ra -= 4 * 16 * BK * 2;
            CC_RD(Bx[0][0], rb, 0); CC_RD(Bx[1][0], rb, 16 * BK * 2);
            if (!FAR) { CC_RD(Bx[0][1], rb, PIECE * 2); CC_RD(Bx[1][1], rb, PIECE * 2 + 16 * BK * 2); }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[1][0]), "+v"(A[1][1]), "+v"(A[2][0]), "+v"(A[2][1]), "+v"(A[3][0]),
                         "+v"(A[3][1]), "+v"(Bx[0][0]), "+v"(Bx[1][0]), "+v"(Bx[0][1]), "+v"(Bx[1][1]));
        }
        for (int kt = 0; kt < n; ++kt) {
            const int nseg_ = e[0] >> 16;
            if (bs && nseg_ != seg) {              // workgroup-uniform: the accumulators move to the scales of this step's segment
                if (seg >= 0) {
                    // the scales are per (row, segment), but neighbouring segments mostly carry the same ones: nothing to do then
                    const unsigned char *to = sctab + (seg - seg0) * BM + srow, *tn_ = sctab + (nseg_ - seg0) * BM + srow;
                    unsigned so[4], sn[4];
                    bool same = true;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        so[i] = *reinterpret_cast<const unsigned *>(to + i * 16);        // rows srow + 16 i + 0..3
                        sn[i] = *reinterpret_cast<const unsigned *>(tn_ + i * 16);
                        same = same && so[i] == sn[i];
                    }
                    if (__builtin_amdgcn_ballot_w64(!same) != 0ull) {                     // wave-uniform
                        CC_MFMA_DRAIN();
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4v rt = {cc2_pow2_ratio(so[i] & 255u, sn[i] & 255u), cc2_pow2_ratio((so[i] >> 8) & 255u, (sn[i] >> 8) & 255u),
                                               cc2_pow2_ratio((so[i] >> 16) & 255u, (sn[i] >> 16) & 255u), cc2_pow2_ratio(so[i] >> 24, sn[i] >> 24)};
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[i][j] *= rt;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                seg = nseg_;
            }
            // list entry of K step kt + D (the DMA of unit 3): requested here, complete at the wait behind unit 0
            unsigned ev;
            asm volatile("ds_read_b32 %0, %1" : "=v"(ev) : "v"(kaddr + 4u * (unsigned)(lb + (kt + D < n ? kt + D : n - 1))));
            const char *kz = baseA;                // unit 0 .. 2 issue no DMA
            const unsigned ls2 = ldsw + stofs(FAR, kt & (D - 1));
            cc16_unit<0, false, FAR>(acc, A, Bx, By, ra, rb, kz, kz, pA2, pB2, offA, offB, ls2);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(By[0][0]), "+v"(By[1][0]), "+v"(By[0][1]), "+v"(By[1][1]), "+v"(ev));
            const int en = __builtin_amdgcn_readfirstlane((int)ev);
            const char *ka2 = baseA + (long)(en & 0xFFFF) * (BK * 2), *kb2 = baseB + (long)(en & 0xFFFF) * (BK * 2);
            cc16_unit<1, false, FAR>(acc, A, By, Bx, ra, rb, kz, kz, pA2, pB2, offA, offB, ls2);
            CC16_WAITB(0, Bx);
            cc16_unit<2, false, FAR>(acc, A, Bx, By, ra, rb, kz, kz, pA2, pB2, offA, offB, ls2);
            CC16_WAITB(0, By);
            // K step kt + 1 landed (younger DMA groups may still be in flight); every wave holds the last fragments of this
            // step's stage in registers
            if (FAR) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            {
                const unsigned nx = stofs(FAR, (kt + 1) & (D - 1));
                ra = ra0 + nx; rb = ra0 + fbd + nx;
            }
            // unit 3: reads from the next stage (B of the next unit 0, A of the next K step), DMA of K step kt + D
            cc16_unit<3, true, FAR>(acc, A, By, Bx, ra, rb, ka2, kb2, pA2, pB2, offA, offB, ls2);
            if (FAR) { CC16_WAITB(1, Bx); }        // all but row block 3's A fragments (awaited inside the next unit 0)
            else { CC16_WAITB(2, Bx); }
#pragma unroll
            for (int s = 0; s + 1 < D; ++s) e[s] = e[s + 1];
            e[D - 1] = en;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };
    kloop(std::false_type{}, 0, nnear);
    if (nfar > 0) {
        __syncthreads();                           // every wave is done with the stages of the near pass
        kloop(std::true_type{}, nnear, nfar);
    }
    if (!active) return;
    CC_MFMA_DRAIN();
#undef CC_DMA
#undef CC_RD
    {
        char *Cb = reinterpret_cast<char *>(g.C + (long)b * g.sC + (long)sk * g.sCsplit + (long)m0 * g.ldc);
        int cj[8];                                 // column of C of the lane's element of column block j (-1: none); blocks of 16 stay whole
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c0 = bcol(tn, wn * 128 + j * 16);
            cj[j] = c0 >= 0 ? c0 + l15 : -1;
        }
        const unsigned ldc4 = (unsigned)(g.ldc * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float rsc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = srow + i * 16 + r;
                int row = m0 + lr;
                row = row < g.M ? row : g.M - 1;
                rsc[r] = bs ? (seg >= 0 ? __uint_as_float((unsigned)sctab[(seg - seg0) * BM + lr] << 23) : 1.f) : f16x2_scale_of(__uint_as_float(g.amax[(long)b * g.M + row]));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + srow + i * 16 + r;
                const float sc = rsc[r] * g.sB16;
                const unsigned o_ = (unsigned)(srow + i * 16 + r) * ldc4;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (row < g.M && cj[j] >= 0 && cj[j] < g.N) *reinterpret_cast<float *>(Cb + (o_ + (unsigned)cj[j] * 4u)) = acc[i][j][r] * sc;
            }
        }
    }
}
#undef CC16_WAITB

// dst[q*plane + i] = fp16 piece q of src[i] * inv (round to nearest), four elements per thread
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

int launch_split_rows2h(hipStream_t stream, const float *src, const unsigned *rowmax, unsigned short *dst2, int rows, int ld, long plane) {
    if (ld % 4 || plane % 4 || rows < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(split_rows2h_kernel, dim3((ld / 4 + 255) / 256, rows), dim3(256), 0, stream, src, rowmax, dst2, ld, plane);
    return (int)hipGetLastError();
}

// A as fp16 pieces A3[q*pA3 + m*lda + k] of A[m][k] / scale(amax[m]) (launch_split_rows2h), B as pieces B16 of B / sB16
// argument checks of one product; *items = its work items (tiles x slabs x batch entries)
static int check_gemm_nt_f16x2_cc(const GemmArgs &g, long *items) {
    if (g.M % 64 || g.N % 128 || g.K % (BK * g.splitK) || g.splitK < 1 || g.batch < 1 || g.accumulate || g.lda % 8 || g.ldb % 8 || !g.A3 ||
        !g.B16 || g.pA3 % 8 || g.pB16 % 8 || (!g.amax && !g.bscale) || !(g.sB16 > 0.f))
        return (int)hipErrorInvalidValue;
    if (g.bscale && (g.batch != 1 || g.segLinP < BK || g.segLinP % BK || g.segChunks != (g.segLinP + 1023) / 1024 || g.K % g.segLinP))
        return (int)hipErrorInvalidValue;
    if ((double)(BM + 1) * (double)g.ldc * 4.0 >= 2147483648.0) return (int)hipErrorInvalidValue;
    // the DMA addresses a row by a 32-bit byte offset from the operand's base
    if ((double)g.M * (double)g.lda * 2.0 >= 4294967296.0 || (double)g.N * (double)g.ldb * 2.0 >= 4294967296.0) return (int)hipErrorInvalidValue;
    if (g.klist) {             // every slab may visit every segment; a slab's K steps sit in LDS
        if (g.batch != 1 || g.klistStride < 2 + g.K / BK || g.K / BK > CC2_MAXLIST || g.K / BK > 0xFFFF) return (int)hipErrorInvalidValue;
        if (g.bscale && (g.K / g.segLinP) * g.segChunks > CC2_MAXSEG) return (int)hipErrorInvalidValue;
    } else {
        const int Kper = g.K / g.splitK;
        if (Kper / BK > CC2_MAXLIST || g.K / BK > 0xFFFF) return (int)hipErrorInvalidValue;
        if (g.bscale) {        // the scales of the K segments a slab crosses sit in LDS
            auto seg_of = [&](int k) { return (k / g.segLinP) * g.segChunks + (k % g.segLinP) / 1024; };
            for (int sk = 0; sk < g.splitK; ++sk)
                if (seg_of(sk * Kper + Kper - 1) - seg_of(sk * Kper) + 1 > CC2_MAXSEG) return (int)hipErrorInvalidValue;
        }
    }
    long tilesN = (g.N + BN - 1) / BN;
    if (g.permP) {
        if ((g.permP != 1 && g.permP != 2 && g.permP != 4 && g.permP != 8) || g.permLin < 1 || g.permLin % (BN / g.permP) || g.N % g.permLin)
            return (int)hipErrorInvalidValue;
        tilesN = (long)((g.N / g.permLin + g.permP - 1) / g.permP) * (g.permLin / (BN / g.permP));
    }
    *items = (long)((g.M + BM - 1) / BM) * tilesN * g.batch * g.splitK;
    return 0;
}

int launch_gemm_nt_f16x2_cc_group(hipStream_t stream, const GemmArgs *gs, int n) {
    if (n < 1 || n > GEMM_GROUP_MAX) return (int)hipErrorInvalidValue;
    GemmGroup grp;
    grp.n = n;
    grp.first[0] = 0;
    for (int i = 0; i < n; ++i) {
        long items = 0;
        if (int e = check_gemm_nt_f16x2_cc(gs[i], &items)) return e;
        grp.g[i] = gs[i];
        grp.first[i + 1] = grp.first[i] + items;
    }
    for (int i = n; i < GEMM_GROUP_MAX; ++i) grp.first[i + 1] = grp.first[n];
    const long total = grp.first[n];
    if (total >= 2147483647L - 8) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)(8 * ((total + 7) / 8)));
    static unsigned long long attr_done = 0;
    if (int e = ensure_dynamic_lds(gemm_nt_f16x2_cc_kernel, LDS3_BYTES, attr_done)) return e;
    hipLaunchKernelGGL(gemm_nt_f16x2_cc_kernel, grid, dim3(512), LDS3_BYTES, stream, grp);
    return (int)hipGetLastError();
}

int launch_gemm_nt_f16x2_cc(hipStream_t stream, const GemmArgs &g) { return launch_gemm_nt_f16x2_cc_group(stream, &g, 1); }
