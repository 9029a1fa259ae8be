// EXPERIMENT (not part of the product library; see README.md here): wave-specialised form of the split-bf16 folded DFT pass
// (arithmetic, operand layout and arguments of surfh_amd/csrc/dft_rx3.h).  Correct (it passed tests/test_gpu_parity.py as the
// product path), but no faster than dft_rx3.hip: 0.58-0.92 ms per pass on config 3 against 0.48-0.79.
//
// One persistent workgroup of 8 waves per CU walks a contiguous range of tiles; a tile is 128 output rows x 128 columns
// (lambda), 64 columns for the packed complex pass.  Waves 4-7 are PRODUCERS, waves 0-3 CONSUMERS; wave w + 4 and wave w share a
// SIMD, so the producer's vector work issues beside its partner's MFMAs.  One raw s_barrier per k-step hands a stage over.
//
// What each wave may wait for decides the structure.  vmcnt is ONE in-order counter per wave: a wave with slow operations in
// flight (stores draining at the chip's write rate; data loads two k-steps ahead) cannot wait for a younger, short one
// without waiting for everything in front of it.  And a kernel that contains LDS-DMA (global_load_lds) makes hipcc put
// s_waitcnt vmcnt(0) in front of EVERY LDS access of every wave (it cannot tell which DMA an access may alias).  Measured
// dead ends, each correct: consumers that issued the matrix DMA and the stores (every k-step after an epilogue waited for
// that tile's stores: time = compute + stores + loads); data as the MFMA A operand (row-per-lane float4 stores: 64 scattered
// 16-byte pieces per instruction, slower than 128 dword stores); producers with LDS-DMA rings (the inserted waits drained them).
//   * Producers use plain global loads into TWO register sets and write them to LDS two k-steps later (raw fp32 rows of the
//     fold -- mirror rows included -- as whole 512-byte row pieces, the cos / sin matrix tile, the template weights of the
//     lane's column): the compiler counts these loads exactly, one set stays in flight while the other is written.  They then
//     read their lane's rows from LDS, fold, form the spectral mix, split into three bf16 pieces and store them to LDS ALREADY
//     AS MFMA B FRAGMENTS (lane-linear 16-byte stores).
//   * Consumers never wait on vmcnt: fragment reads, 48 MFMAs per k-step, and per tile an epilogue that sends the
//     accumulators through the (just consumed) B-fragment stage of the wave to turn "lane = column" into 16-byte rows:
//     32 float4 stores per lane and tile (eight 128-byte row segments per instruction) instead of 128 one-dword stores, which
//     queue (< 63 outstanding) and drain while the next tile is computed.
#include "dft_ws.h"
#include "../../surfh_amd/csrc/lds_attr.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));      // (arrays of HIP's float4 / uint4 structs are copied with memcpy and end up in scratch)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// loads through pointers the compiler cannot prove global (selected per thread, carried across the loop) would be flat loads,
// which also count in lgkmcnt and would be drained by every LDS wait
#define WS_GLOBAL(T_, p_) (*((const __attribute__((address_space(1))) T_ *)(p_)))

namespace {

constexpr int BK = 16, RS = 16;
constexpr int PIECE = 128 * RS;               // one bf16 piece of one 128-row matrix tile (elements)
constexpr int IMG = 3 * PIECE;
constexpr int ABUF = 2 * IMG;                 // both matrices: 24 KB
constexpr int NA = 2;
constexpr int BFRAG = 64 * 8;                 // one B fragment of one wave: 64 lanes x 8 bf16 = 1 KB
constexpr int BWAVE = 6 * BFRAG;              // (stream 0: h, m, l; stream 1: h, m, l)
constexpr int BBUF = 4 * BWAVE;               // 24 KB
constexpr int NB = 2;
#ifndef WS_EXP                                // tools/exp only, bit mask: 1 no raw-data DMA, 2 no MFMAs, 4 no stores, 8 no fold / split arithmetic, 16 half the MFMAs, 32 no matrix loads, 64 no L2 prefetch
#define WS_EXP 0
#endif
constexpr int MIX_ROWS_MAX = 1024;            // rows (k) of the spectral-mix table held in LDS: 2 x 1024 x 16 B = 32 KB at most (160 KB in all)
constexpr size_t LDS_MAIN = (size_t)(NA * ABUF + NB * BBUF) * sizeof(unsigned short);

__device__ __forceinline__ unsigned pack2(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// exact split of 8 values into three bf16x8 fragments (h, m, l)
__device__ __forceinline__ void split8(const float (&x)[8], uint4 &fh, uint4 &fm, uint4 &fl) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned u = __float_as_uint(x[j]);
        h[j] = u & 0xFFFF0000u;
        const float r = x[j] - __uint_as_float(h[j]);
        m[j] = __float_as_uint(r) & 0xFFFF0000u;
        l[j] = __float_as_uint(r - __uint_as_float(m[j]));
    }
    fh = make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
    fm = make_uint4(pack2(m[0], m[1]), pack2(m[2], m[3]), pack2(m[4], m[5]), pack2(m[6], m[7]));
    fl = make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7]));
}

// raw workgroup barrier between two compiler memory fences: no memory operation moves across it, and no s_waitcnt is
// added to it (the waits each role needs stand in front of it by hand)
#define WS_BARRIER()                                    \
    {                                                   \
        asm volatile("" ::: "memory");                  \
        __builtin_amdgcn_s_barrier();                   \
        asm volatile("" ::: "memory");                  \
    }

#if WS_EXP & 2
#define MFMA6(acc_, ah_, am_, al_, bh_, bm_, bl_)                                   \
    {                                                                               \
        asm volatile("" ::"v"(ah_), "v"(am_), "v"(al_), "v"(bh_), "v"(bm_), "v"(bl_)); \
    }
#elif WS_EXP & 16
#define MFMA6(acc_, ah_, am_, al_, bh_, bm_, bl_)                                   \
    {                                                                               \
        f32x16 c_ = acc_;                                                           \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, bl_, c_, 0, 0, 0);        \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am_, bm_, c_, 0, 0, 0);        \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bh_, c_, 0, 0, 0);        \
        acc_ = c_;                                                                  \
    }
#else
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
#endif

// KIND 0: two source streams, folded, PACKED (complex pass, both output components from one read of a 64-column tile);
//      1: one real source feeding both streams (r2c); 2: two streams, no fold (c2r).
template <int KIND, bool MIX>
__global__ __launch_bounds__(512, 2) void dft_ws_kernel(DftWsArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 3;                                   // which 32 lanes-worth of columns of the tile
    const int l31 = lane & 31, h = lane >> 5;
    constexpr bool packed = (KIND == 0);
    constexpr int TN = packed ? 64 : 128;                      // columns of a tile
    const int var = packed ? (l31 >> 4) : 0;                   // packed: which variant this lane's column copy carries
    const int lcol = packed ? cw * 16 + (l31 & 15) : cw * 32 + l31;
    const int tilesX = g.N / TN, tilesY = g.MP / 128;
    const long ntile = (long)tilesX * tilesY * g.batch;
    const int nk = g.KP / BK;
    // this workgroup's tiles: the contiguous range [t0, t1) (neighbouring tiles share the spectral-mix table and memory pages)
    const long t0 = ntile * (long)blockIdx.x / (long)gridDim.x, t1 = ntile * ((long)blockIdx.x + 1) / (long)gridDim.x;
    const int ntl = (int)(t1 - t0);
    const int S = ntl * nk;                                    // k-steps this workgroup runs, over all its tiles
    if (S == 0) return;
    unsigned short *ldsA = lds, *ldsB = lds + NA * ABUF;
    float4 *mtab = reinterpret_cast<float4 *>(lds + NA * ABUF + NB * BBUF);      // [2 mix_rows] mhat column of the tile's kb (fused spectral mix)
    const int mixn = g.mix_rows * 2;
    // k_beta of local tile i (uniform): the spectral-mix table changes where it does
    auto kb_of = [&](int i) { return (int)((((t0 + i) % tilesX) * TN) / g.LP); };

    if (wave < 4) {
        // ============================================================== consumers
        f32x16 acc1[4], acc2[4];
        WS_BARRIER();                                              // end of the producers' prologue
        WS_BARRIER();                                              // end of body 0: matrix tile and B fragments of k-step 0 are in LDS
        int ti = 0, kt = 0;
        for (int s = 0; s < S; ++s) {
            if (kt == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[i][r] = acc2[i][r] = 0.f;
            }
            // the producers reload the mix table between two barriers when the next tile belongs to another k_beta
            if (MIX && kt == nk - 1 && ti + 1 < ntl && kb_of(ti + 1) != kb_of(ti)) WS_BARRIER();
            {
                const unsigned short *ra = ldsA + (s & 1) * ABUF + l31 * RS + 8 * (h ^ ((l31 >> 3) & 1));
                const unsigned short *rb = ldsB + (s & 1) * BBUF + cw * BWAVE + lane * 8;
                const bf16x8 b0h = *reinterpret_cast<const bf16x8 *>(rb), b0m = *reinterpret_cast<const bf16x8 *>(rb + BFRAG),
                             b0l = *reinterpret_cast<const bf16x8 *>(rb + 2 * BFRAG), b1h = *reinterpret_cast<const bf16x8 *>(rb + 3 * BFRAG),
                             b1m = *reinterpret_cast<const bf16x8 *>(rb + 4 * BFRAG), b1l = *reinterpret_cast<const bf16x8 *>(rb + 5 * BFRAG);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const unsigned short *p = ra + mt * 32 * RS;
                    const bf16x8 a0h = *reinterpret_cast<const bf16x8 *>(p);
                    const bf16x8 a0m = *reinterpret_cast<const bf16x8 *>(p + PIECE);
                    const bf16x8 a0l = *reinterpret_cast<const bf16x8 *>(p + 2 * PIECE);
                    MFMA6(acc1[mt], a0h, a0m, a0l, b0h, b0m, b0l)
                    const bf16x8 a1h = *reinterpret_cast<const bf16x8 *>(p + IMG);
                    const bf16x8 a1m = *reinterpret_cast<const bf16x8 *>(p + IMG + PIECE);
                    const bf16x8 a1l = *reinterpret_cast<const bf16x8 *>(p + IMG + 2 * PIECE);
                    MFMA6(acc2[mt], a1h, a1m, a1l, b1h, b1m, b1l)
                }
            }
            if (kt == nk - 1) {
                // Epilogue of tile ti.  An accumulator holds a column (lane) x 16 rows (registers); half a 32-row block at a time
                // goes through this wave's 6 KB of the B-fragment stage it has just consumed (free until the next barrier) as
                // [16 rows][32 columns] floats, and comes back as float4 row pieces: lane L = row L >> 3, columns 4 (L & 7)...
                const long t = t0 + ti;
                const int tx = (int)(t % tilesX), ty = (int)((t / tilesX) % tilesY);
                const long bz = t / ((long)tilesX * tilesY);
                const int n0 = tx * TN, em0 = ty * 128;
                // packed: a second-variant lane holds (acc1, acc2) = (A[0] X_second, A[1] X_first), i.e. that variant's products swapped
                const float e00 = var ? g.e_alt[1] : g.e00, e01 = var ? g.e_alt[0] : g.e01;
                const float e10 = var ? g.e_alt[3] : g.e10, e11 = var ? g.e_alt[2] : g.e11;
                float *stg = reinterpret_cast<float *>(ldsB + (s & 1) * BBUF + cw * BWAVE);       // two buffers of 16 x 32 floats
                const int rrow = lane >> 3, cg = lane & 7;
                const int rvar = packed ? (cg >> 2) : 0;
                float *Dv = (rvar ? g.dst_alt : g.dst[0]) + bz * g.sC + n0 + (packed ? cw * 16 + 4 * (cg & 3) : cw * 32 + 4 * cg);
                float *D1 = (g.dst[1] ? g.dst[1] : g.dst[0]) + bz * g.sC + n0 + cw * 32 + 4 * cg;   // SPLIT mode (never packed)
#pragma unroll
                for (int hr = 0; hr <= 16; ++hr) {
                    if (hr < 16) {                                 // write half-round hr: block mt, rows 16 half + (0..15), output o
                        const int mt = hr >> 2, half = (hr >> 1) & 1, o = hr & 1;
                        float *w = stg + (hr & 1) * 512 + (4 * h) * 32 + l31;
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) {
                            const float a1 = acc1[mt][8 * half + rr], a2 = acc2[mt][8 * half + rr];
                            float v;
                            if (g.mode == 0) v = o ? (e10 * a1 + e11 * a2) : (e00 * a1 + e01 * a2);
                            else v = o ? (e11 * a2) : (e00 * a1);
                            w[((rr & 3) + 8 * (rr >> 2)) * 32] = v;
                        }
                    }
                    if (hr > 0) {                                  // read half-round hr - 1 back as rows and store them
                        const int q = hr - 1, mt = q >> 2, half = (q >> 1) & 1, o = q & 1;
                        const float *rd = stg + (q & 1) * 512 + rrow * 32 + 4 * cg;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const float4 v = *reinterpret_cast<const float4 *>(rd + i * 256);
                            const int row = em0 + mt * 32 + 16 * half + 8 * i + rrow;
                            if (WS_EXP & 4) {
                                asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                            } else if (g.mode == 0) {
                                if (o == 0) {
                                    if (row < g.rvalid) *reinterpret_cast<float4 *>(Dv + (long)row * g.ldc) = v;
                                } else {
                                    if (row < g.rvalid && row >= 1 && 2 * row != g.Rn) *reinterpret_cast<float4 *>(Dv + (long)(g.Rn - row) * g.ldc) = v;
                                }
                            } else {
                                if (row < g.rvalid) *reinterpret_cast<float4 *>((o ? D1 : Dv) + (long)row * g.ldc) = v;
                            }
                        }
                    }
                }
            }
            if (s + 1 < S) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                WS_BARRIER();
            }
            if (++kt == nk) { kt = 0; ++ti; }
        }
        return;
    }

    // ================================================================== producers
    const int pw = wave - 4, ptid = tid - 256;
    const int kin = g.Kn / 2 + 1;
    float4 tw = make_float4(0.f, 0.f, 0.f, 0.f);                   // template weights of the lane's column (MIX)
    int hv = h;
    const unsigned ldb4 = (unsigned)(g.ldb * 4), c4 = (unsigned)lcol * 4u;
    const long ldbB = g.ldb * 4;
    // matrix tiles: position 2*row + c of a piece holds the k-half c ^ ((row >> 3) & 1) of the row, which makes the fragment
    // reads (ds_read_b128, 16-lane groups) conflict-free.  Producer wave pw fills rows 32 pw .. 32 pw + 31, lane-linear.
    const int amrow = ptid >> 1;
    const unsigned aoff = (unsigned)(amrow * g.lda + 8 * ((lane & 1) ^ ((amrow >> 3) & 1))) * 2u;

    // Three register sets; set f % 3 holds what body f needs -- the lane's 8 (+ 8 mirror) raw rows of k-step f (k = 16 kt + 8 h + j,
    // straight from HBM: consecutive lanes are consecutive columns, every instruction two 128-byte segments), the lane's 16
    // bytes x 6 of the matrix tile, the template weights of its column -- and is refilled for k-step f + 3 at the end of that body
    // (plain loads: the compiler counts them, the other two sets' stay in flight).  Behind the end of the stream the cursor stops.
    float rx0[8], ri0[8], rq0[8], rp0[8], rx1[8], ri1[8], rq1[8], rp1[8], rx2[8], ri2[8], rq2[8], rp2[8];
    f32x4 stw0, stw1, stw2;
    u32x4 sA0[6], sA1[6], sA2[6];
    int li = 0, lkt = 0, lstep = 0;                                // load cursor: k-step lstep = k-step lkt of local tile li
    const char *LB0 = nullptr, *LB1 = nullptr;
    const char *LA0 = nullptr, *LA1 = nullptr;
    int ln0 = 0;
#define WS_LSETUP()                                                                                                 \
    {                                                                                                               \
        const long t_ = t0 + li;                                                                                    \
        const long bz_ = t_ / ((long)tilesX * tilesY);                                                              \
        const int m0_ = (int)((t_ / tilesX) % tilesY) * 128;                                                        \
        ln0 = (int)(t_ % tilesX) * TN;                                                                              \
        LB0 = reinterpret_cast<const char *>(g.src[0] + bz_ * g.sB + ln0);                                          \
        LB1 = reinterpret_cast<const char *>(g.src[1] + bz_ * g.sB + ln0);                                          \
        LA0 = reinterpret_cast<const char *>(g.A[0] + (long)m0_ * g.lda) + aoff;                                    \
        LA1 = reinterpret_cast<const char *>(g.A[1] + (long)m0_ * g.lda) + aoff;                                    \
    }
    // Row part of every address in 64-bit scalar pointers, lane part in one small VGPR.  Mirror row of k is
    // (Kn - 16 kt - 8 - j) + 8 (1 - h); read unconditionally (its weight is zero where there is no mirror), except k = 0 whose
    // "mirror" Kn may not exist.
#define WS_LOAD(x_)                                                                                                 \
    {                                                                                                               \
        const char *rk0 = LB0 + (long)(lkt * BK) * ldbB, *rk1 = LB1 + (long)(lkt * BK) * ldbB;                      \
        const char *rp0_ = LB0 + (long)(g.Kn - lkt * BK - 8) * ldbB, *rp1_ = LB1 + (long)(g.Kn - lkt * BK - 8) * ldbB; \
        const unsigned vk = (unsigned)(8 * hv) * ldb4 + c4, vp = (unsigned)(8 * (1 - hv)) * ldb4 + c4;              \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                            \
            if (WS_EXP & 1) { rx##x_[j] = ri##x_[j] = rq##x_[j] = rp##x_[j] = (float)(lstep + j); continue; }       \
            rx##x_[j] = WS_GLOBAL(float, rk0 + j * ldbB + vk);                                                      \
            if (KIND != 1) ri##x_[j] = WS_GLOBAL(float, rk1 + j * ldbB + vk);                                       \
            if (KIND != 2) {                                                                                       \
                const unsigned q = (j == 0 && lkt == 0) ? c4 : vp;                                                 \
                rq##x_[j] = WS_GLOBAL(float, rp0_ - j * ldbB + q);                                                  \
                if (KIND != 1) rp##x_[j] = WS_GLOBAL(float, rp1_ - j * ldbB + q);                                   \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                            \
            sA##x_[q] = (WS_EXP & 32) ? u32x4{(unsigned)q, 1u, 2u, 3u} : WS_GLOBAL(u32x4, LA0 + 2 * q * g.planeA + (unsigned)(lkt * BK) * 2u); \
            sA##x_[3 + q] = (WS_EXP & 32) ? u32x4{(unsigned)q, 4u, 5u, 6u} : WS_GLOBAL(u32x4, LA1 + 2 * q * g.planeA + (unsigned)(lkt * BK) * 2u); \
        }                                                                                                          \
        if (MIX) stw##x_ = WS_GLOBAL(f32x4, g.tplT + ln0 % g.LP + lcol);                                            \
        if (lstep + 1 < S) {                                                                                       \
            ++lstep;                                                                                               \
            if (++lkt == nk) { lkt = 0; ++li; WS_LSETUP(); }                                                       \
        }                                                                                                          \
    }
#define WS_A_WRITE(x_, step_)                                                                                       \
    {                                                                                                               \
        u32x4 *lb = reinterpret_cast<u32x4 *>(ldsA + ((step_) & 1) * ABUF + pw * 512) + lane;                       \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                            \
            lb[q * (PIECE / 8)] = sA##x_[q];                                                                       \
            lb[(IMG + q * PIECE) / 8] = sA##x_[3 + q];                                                             \
        }                                                                                                          \
    }
    // spectral-mix table of local tile i_'s k_beta ([k][re/im] x 4 templates, laid out by mix_table_kernel): global -> LDS
#define WS_MIXTAB_LOAD(i_)                                                                                          \
    {                                                                                                               \
        const float4 *src = g.mixtab + (long)kb_of(i_) * mixn;                                                      \
        for (int e = ptid; e < mixn; e += 256) {                                                                    \
            const f32x4 v_ = WS_GLOBAL(f32x4, src + e);                                                             \
            *reinterpret_cast<f32x4 *>(mtab + e) = v_;                                                              \
        }                                                    \
    }
    // fold (and mix) the raw rows of k-step fkt of tile fi (LDS stage rs_) into the two data streams, split, store as B fragments
#define WS_FOLD(x_, st_)                                                                                            \
    {                                                                                                               \
        float x0[8], x1[8];                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                            \
            const int k = fkt * BK + 8 * hv + j;                                                                    \
            float ar = rx##x_[j], ai_ = (KIND == 1) ? rx##x_[j] : ri##x_[j];                                        \
            float br = (KIND == 2) ? 0.f : rq##x_[j], bi = (KIND == 2) ? 0.f : (KIND == 1) ? rq##x_[j] : rp##x_[j]; \
            if (MIX) {                                                                                             \
                const float4 mr = mtab[2 * k], mi = mtab[2 * k + 1];                                               \
                const float sr = tw.x * mr.x + tw.y * mr.y + tw.z * mr.z + tw.w * mr.w;                            \
                const float si = tw.x * mi.x + tw.y * mi.y + tw.z * mi.z + tw.w * mi.w;                            \
                const float hr = ar, hi = ai_;                                                                     \
                ar = hr * sr - hi * si;                                                                            \
                ai_ = hr * si + hi * sr;                                                                           \
            }                                                                                                      \
            if (KIND == 2) {                                                                                       \
                x0[j] = ar;                                                                                        \
                x1[j] = ai_;                                                                                       \
            } else {                                                                                               \
                const bool pv = (k >= 1) && (k < kin) && (2 * k != g.Kn);                                          \
                if (MIX) {                                                                                         \
                    const int kp = pv ? g.Kn - k : k;                                                              \
                    const float4 mr = mtab[2 * kp], mi = mtab[2 * kp + 1];                                         \
                    const float sr = tw.x * mr.x + tw.y * mr.y + tw.z * mr.z + tw.w * mr.w;                        \
                    const float si = tw.x * mi.x + tw.y * mi.y + tw.z * mi.z + tw.w * mi.w;                        \
                    const float hr = br, hi = bi;                                                                  \
                    br = hr * sr - hi * si;                                                                        \
                    bi = hr * si + hi * sr;                                                                        \
                }                                                                                                  \
                const float g0 = var ? g.fold_alt[0] : g.fold[0], g1 = var ? g.fold_alt[1] : g.fold[1];            \
                const float f0 = pv ? g0 : 0.f, f1 = pv ? g1 : 0.f;                                                \
                const float w0 = (!pv && g0 < 0.f) ? 0.f : 1.f, w1 = (!pv && g1 < 0.f) ? 0.f : 1.f;                \
                const float s0 = w0 * ar + f0 * br, s1v = w1 * ai_ + f1 * bi;                                      \
                x0[j] = var ? s1v : s0;       /* the second variant pairs its streams with the other matrix */      \
                x1[j] = var ? s0 : s1v;                                                                            \
            }                                                                                                      \
        }                                                                                                          \
        uint4 f0h, f0m, f0l, f1h, f1m, f1l;                                                                        \
        if (WS_EXP & 8) {                                                                                          \
            f0h = make_uint4(__float_as_uint(x0[0]), __float_as_uint(x0[1]), __float_as_uint(x0[2]), __float_as_uint(x0[3])); \
            f0m = make_uint4(__float_as_uint(x0[4]), __float_as_uint(x0[5]), __float_as_uint(x0[6]), __float_as_uint(x0[7])); \
            f1h = make_uint4(__float_as_uint(x1[0]), __float_as_uint(x1[1]), __float_as_uint(x1[2]), __float_as_uint(x1[3])); \
            f1m = make_uint4(__float_as_uint(x1[4]), __float_as_uint(x1[5]), __float_as_uint(x1[6]), __float_as_uint(x1[7])); \
            f0l = f0h; f1l = f1h;                                                                                  \
        } else {                                                                                                   \
            split8(x0, f0h, f0m, f0l);                                                                              \
            split8(x1, f1h, f1m, f1l);                                                                              \
        }                                                                                                          \
        uint4 *wb = reinterpret_cast<uint4 *>(ldsB + (st_) * BBUF + cw * BWAVE) + lane;                            \
        wb[0] = f0h; wb[64] = f0m; wb[128] = f0l; wb[192] = f1h; wb[256] = f1m; wb[320] = f1l;                     \
    }
    // body f (beside the consumers' k-step f - 1): register set x_ = f & 1 -> LDS (raw rows of k-step f + 1, matrix tile of
    // k-step f), refill it (k-steps f + 3 / f + 2), fold k-step f
#define WS_BODY(x_, p_)                                                                                               \
    {                                                                                                               \
        if (MIX && fkt == 0 && fi > 0 && kb_of(fi) != kb_of(fi - 1)) {                                              \
            /* another k_beta: every wave has left the old table behind (previous barrier); fetch the new one, publish it */ \
            WS_MIXTAB_LOAD(fi);                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
            WS_BARRIER();                                                                                           \
        }                                                                                                          \
        if (MIX && fkt == 0) tw = make_float4(stw##x_[0], stw##x_[1], stw##x_[2], stw##x_[3]);                      \
        WS_A_WRITE(x_, f);                                                                                          \
        WS_FOLD(x_, f & 1);                                                                                         \
        WS_LOAD(x_);                                                                                                \
        if (++fkt == nk) { fkt = 0; ++fi; }                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
        WS_BARRIER();                                                                                               \
        ++f;                                                                                                        \
    }

    // ---- prologue: sets 0 / 1 / 2 <- k-steps 0 / 1 / 2; the mix table of tile 0 into LDS
    WS_LSETUP();
    WS_LOAD(0);
    WS_LOAD(1);
    WS_LOAD(2);
    if (MIX) WS_MIXTAB_LOAD(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WS_BARRIER();
    int fi = 0, fkt = 0, f = 0;
    while (f + 3 <= S) {
        asm volatile("" : "+v"(hv));   // keeps the per-lane fold selectors from being hoisted out of the loop
        WS_BODY(0, 0)
        WS_BODY(1, 1)
        WS_BODY(2, 2)
    }
    if (f < S) WS_BODY(0, 0)
    if (f < S) WS_BODY(1, 1)
#undef WS_LSETUP
#undef WS_LOAD
#undef WS_A_WRITE
#undef WS_MIXTAB_LOAD
#undef WS_FOLD
#undef WS_BODY
}

}  // namespace

// Whether launch_dft_ws can run this pass (the callers fall back to launch_dft_rx3 otherwise)
bool dft_ws_can(const DftWsArgs &g) {
    if (g.MP % 128 || g.KP % BK || g.batch < 1) return false;
    if (g.nvar != 1 && !(g.nvar == 2 && g.packed)) return false;
    const bool folded = g.fold[0] != 0.f || g.fold[1] != 0.f || (g.nvar == 2 && (g.fold_alt[0] != 0.f || g.fold_alt[1] != 0.f));
    const int kind = !folded ? 2 : (g.src[0] == g.src[1] ? 1 : 0);
    if (kind != 2 && (g.Kn < 2 || g.KP > g.Kn + 16)) return false;     // folded: mirror rows Kn - k of every k < KP exist (clamped at 0)
    if (kind == 0 && g.nvar != 2) return false;       // the complex pass is built in its packed form only
    if (kind != 0 && g.nvar != 1) return false;
    if ((long)g.ldb % 4 || (long)g.ldc % 4 || (long)g.sB % 4 || (long)g.sC % 4) return false;     // 16-byte DMA and float4 stores
    if (g.nvar == 2) {
        const long dalt = g.dst_alt ? (long)(g.dst_alt - g.dst[0]) : -1;
        if (g.mode != 0 || g.N % 64 || dalt < 0 || dalt % 4 || !g.A_alt[0]) return false;
        // the packed form keeps A[0], A[1] for both variants: the second variant's matrices must be the first one's, swapped
        if (g.A_alt[0] != g.A[1] || g.A_alt[1] != g.A[0]) return false;
    } else if (g.N % 128) return false;
    if (g.mhat) {      // fused spectral mix: needs the table built by launch_dft_ws_mix_table and whole-DMA table sizes
        if (kind != 0 || g.LP % 64 || g.T < 1 || g.T > 4 || g.KP / BK < 2 || !g.mixtab || !g.tplT ||
            g.mix_rows % 128 || g.mix_rows < g.Kn || g.mix_rows < g.KP || g.mix_rows > MIX_ROWS_MAX)
            return false;
    }
    return true;
}

int launch_dft_ws(hipStream_t stream, const DftWsArgs &g) {
    if (!dft_ws_can(g)) return (int)hipErrorInvalidValue;
    if (g.mode == 1 && !g.dst[1]) return (int)hipErrorInvalidValue;
    const bool folded = g.fold[0] != 0.f || g.fold[1] != 0.f || (g.nvar == 2 && (g.fold_alt[0] != 0.f || g.fold_alt[1] != 0.f));
    const int kind = !folded ? 2 : (g.src[0] == g.src[1] ? 1 : 0);
    static int cus_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (!cus_of[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return (int)hipErrorInvalidDevice;
        cus_of[dev] = cus;
    }
    DftWsArgs a = g;
    a.packed = (g.nvar == 2) ? 1 : 0;
    a.strided = 0;
    const long ntile = (long)(g.N / (a.packed ? 64 : 128)) * (g.MP / 128) * g.batch;
    dim3 grid((unsigned)(ntile < cus_of[dev] ? ntile : cus_of[dev]));
    const size_t mix_bytes = g.mhat ? (size_t)2 * g.mix_rows * sizeof(float4) : 0;
    static unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    if (g.mhat) {
        if (int e = ensure_dynamic_lds(dft_ws_kernel<0, true>, LDS_MAIN + (size_t)2 * MIX_ROWS_MAX * sizeof(float4), d3)) return e;
        hipLaunchKernelGGL((dft_ws_kernel<0, true>), grid, dim3(512), LDS_MAIN + mix_bytes, stream, a);
    } else if (kind == 0) {
        if (int e = ensure_dynamic_lds(dft_ws_kernel<0, false>, LDS_MAIN, d0)) return e;
        hipLaunchKernelGGL((dft_ws_kernel<0, false>), grid, dim3(512), LDS_MAIN, stream, a);
    } else if (kind == 1) {
        if (int e = ensure_dynamic_lds(dft_ws_kernel<1, false>, LDS_MAIN, d1)) return e;
        hipLaunchKernelGGL((dft_ws_kernel<1, false>), grid, dim3(512), LDS_MAIN, stream, a);
    } else {
        if (int e = ensure_dynamic_lds(dft_ws_kernel<2, false>, LDS_MAIN, d2)) return e;
        hipLaunchKernelGGL((dft_ws_kernel<2, false>), grid, dim3(512), LDS_MAIN, stream, a);
    }
    return (int)hipGetLastError();
}

// mixtab[kb][2 k + c] = (mhat[t][c][k][kb], t = 0..3), zero for t >= T and for k >= Kn: the per-tile table of the fused
// spectral mix in the layout the pass kernel moves to LDS by DMA (one contiguous block of 2 * mix_rows float4 per kb)
namespace {
__global__ __launch_bounds__(256) void mix_table_kernel(const float *__restrict__ mhat, float4 *__restrict__ out, int T, int Kn, int nkb,
                                                        long PL, long KBP, int mix_rows) {
    const int kb = blockIdx.x * 64 + (threadIdx.x & 63);          // kb fastest: contiguous reads of mhat
    const int e = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (kb >= nkb || e >= 2 * mix_rows) return;
    const int k = e >> 1, c = e & 1;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < Kn)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < T) v[t] = mhat[((long)t * 2 + c) * PL + (long)k * KBP + kb];
    out[(long)kb * 2 * mix_rows + e] = make_float4(v[0], v[1], v[2], v[3]);
}
}  // namespace

int dft_ws_mix_rows(int Kn, int KP) { return ((Kn > KP ? Kn : KP) + 127) / 128 * 128; }

int launch_dft_ws_mix_table(hipStream_t stream, const float *mhat, float *mixtab, int T, int Kn, int nkb, long PL, long KBP, int mix_rows) {
    if (T < 1 || T > 4 || nkb < 1 || mix_rows < Kn || mix_rows % 128) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)((nkb + 63) / 64), (unsigned)((2 * mix_rows + 3) / 4));
    hipLaunchKernelGGL(mix_table_kernel, grid, dim3(256), 0, stream, mhat, reinterpret_cast<float4 *>(mixtab), T, Kn, nkb, PL, KBP, mix_rows);
    return (int)hipGetLastError();
}
