// Wave-specialised form of the split-bf16 folded DFT pass (arithmetic, operand layout and arguments of dft_rx3.h).
//
// One persistent workgroup of 8 waves per CU.  Waves 4-7 are PRODUCERS: each owns 32 columns (lambda) of the tile, loads the
// 8 (+8 mirror) rows of its lane's column of every k-step straight from HBM, three k-steps ahead of their use (a register
// ring, so that two k-steps of loads per wave are always in flight), folds them, optionally forms the spectral mix, cuts the
// values into their three bf16 pieces and writes them to LDS ALREADY AS MFMA B FRAGMENTS (lane-linear 16-byte stores: no
// transposition, no bank conflicts).  Waves 0-3 are CONSUMERS: they move the cos / sin matrix tiles of the next k-step into
// LDS by LDS-DMA, read A and B fragments and issue the 48 MFMAs of a k-step -- nothing else, until the tile's epilogue
// stores.  A producer shares its SIMD with the consumer of the same number, so its VALU work issues beside the
// partner's MFMAs instead of in front of them (in dft_rx3.hip both were one wave's instruction stream: the pass took the
// SUM of its matrix-core, fold / split and memory times).  One raw s_barrier per k-step hands a stage over; the pipeline runs
// across tile seams; stores drain in the background (the consumers never wait for them: at a tile's last k-step the
// LDS-DMA of the next stage is known to have landed once 63 younger stores have been issued, s_waitcnt vmcnt(63)).
#include "dft_rx3.h"
#include "lds_attr.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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
#ifndef WS_RING
#define WS_RING 3
#endif
#ifndef WS_EXP                                // tools/exp only, bit mask: 1 no data loads, 2 no MFMAs, 4 no stores, 8 no fold / split arithmetic, 16 half the MFMAs
#define WS_EXP 0
#endif
constexpr int RING = WS_RING;                 // k-steps of raw loads a producer holds (RING - 1 in flight behind the one in use)
constexpr int MIX_ROWS_MAX = 896;             // rows (k) of a tile's spectral-mix table: 2 buffers x (2 x 896 + 128) x 16 B = 60 KB of LDS at most
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

// KIND 0: two source streams, folded (complex pass); 1: one real source feeding both streams (r2c);
//      2: two streams, no fold (c2r).  Units are tiles: nvar == 1, or nvar == 2 packed (see dft_rx3.h).
template <int KIND, bool MIX>
__global__ __launch_bounds__(512, 2) void dft_ws_kernel(DftRx3Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 3;                                   // which 32 columns of the tile
    const int l31 = lane & 31, h = lane >> 5;
    const bool packed = (KIND == 0) && g.packed;
    const int TN = packed ? 64 : 128;
    const int var = packed ? (l31 >> 4) : 0;
    const int lcol = packed ? cw * 16 + (l31 & 15) : cw * 32 + l31;
    const int tilesX = g.N / TN, tilesY = g.MP / 128;
    const int ntile = tilesX * tilesY * g.batch;
    const int nk = g.KP / BK;
    const int ntl = ((int)blockIdx.x < ntile) ? (ntile - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int S = ntl * nk;                                    // k-steps this workgroup runs, over all its tiles
    const long ldbB = g.ldb * 4, ldcB = g.ldc * 4;
    const unsigned ldb4 = (unsigned)ldbB, ldc4 = (unsigned)ldcB, c4 = (unsigned)lcol * 4u;
    if (S == 0) return;                                        // (the launcher never starts more workgroups than tiles)
    unsigned short *ldsA = lds, *ldsB = lds + NA * ABUF;
    float4 *mtab = reinterpret_cast<float4 *>(lds + NA * ABUF + NB * BBUF);   // [2][mixs]: mhat column of a tile ([k][re/im] x 4 templates) + template weights of its columns
    const int mixn = g.mix_rows * 2, mixs = mixn + 128;

    if (wave < 4) {
        // ============================================================== consumers
        const int arow = tid >> 1;                                 // = 32 * wave + (lane >> 1): the matrix row this lane moves
        const unsigned aoff = (unsigned)(arow * g.lda + 8 * ((lane & 1) ^ ((arow >> 3) & 1))) * 2u;
        // matrix tiles: global -> LDS by DMA, lane-linear; position 2*row + c holds the k-half c ^ ((row >> 3) & 1) of the row,
        // which makes the fragment reads (ds_read_b128, 16-lane groups) conflict-free.  Wave w fills rows 32w..32w+31.
#define WS_DMA(i_, kt_, st_)                                                                                       \
    {                                                                                                              \
        const int t_ = (int)blockIdx.x + (i_) * (int)gridDim.x;                                                    \
        const int m0_ = ((t_ / tilesX) % tilesY) * 128;                                                            \
        const char *A0 = reinterpret_cast<const char *>(g.A[0] + (long)m0_ * g.lda);                               \
        const char *A1 = reinterpret_cast<const char *>(g.A[1] + (long)m0_ * g.lda);                               \
        const unsigned ao = aoff + (unsigned)((kt_) * BK) * 2u;                                                    \
        unsigned short *lb = ldsA + (st_) * ABUF + wave * 512;                                                     \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                            \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A0 + 2 * q * g.planeA + ao), \
                                             (__attribute__((address_space(3))) void *)(lb + q * PIECE), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A1 + 2 * q * g.planeA + ao), \
                                             (__attribute__((address_space(3))) void *)(lb + IMG + q * PIECE), 16, 0, 0); \
        }                                                                                                          \
    }
        // fused spectral mix: the tile's column of mhat ([k][re/im] x 4 templates, laid out by mix_table_kernel) and the template
        // weights of its columns, global -> LDS buffer i_ & 1 by DMA (no registers): mixn + TN float4
#define WS_MIXDMA(i_)                                                                                              \
    {                                                                                                              \
        const int t_ = (int)blockIdx.x + (i_) * (int)gridDim.x;                                                    \
        const int n0_ = (t_ % tilesX) * TN;                                                                        \
        const float4 *src = g.mixtab + (long)(n0_ / g.LP) * mixn;                                                  \
        float4 *dstb = mtab + ((i_) & 1) * mixs;                                                                   \
        for (int e = wave * 64; e < mixn; e += 256)        /* mixn is a multiple of 64: whole wave instructions */ \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + e + lane),     \
                                             (__attribute__((address_space(3))) void *)(dstb + e), 16, 0, 0);      \
        if (wave * 64 < TN)                                                                                        \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g.tplT + (n0_ % g.LP) + wave * 64 + lane), \
                                             (__attribute__((address_space(3))) void *)(dstb + mixn + wave * 64), 16, 0, 0); \
    }
        // The data is the A operand and the matrix the B operand of every MFMA (the same register fragments as the other way
        // round): the accumulators then hold the TRANSPOSED output tile -- lane = output row (within its 32-row block mt),
        // registers = 32 wavelengths, four consecutive ones per register quad -- so the epilogue is 32 float4 stores per lane
        // with no transposition.  (With lane = wavelength it was 128 one-dword stores: the wave sat on its full vmcnt queue.)
        f32x16 acc1[4], acc2[4];
        if (MIX) {
            WS_MIXDMA(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        WS_BARRIER();                                              // slot -2 (the producers' prologue)
        WS_DMA(0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WS_BARRIER();                                              // slot -1: matrix tiles and B fragments of step 0 are in LDS
        int ti = 0, kt = 0;
        for (int s = 0; s < S; ++s) {
            if (kt == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[i][r] = acc2[i][r] = 0.f;
            }
            int nkt = kt + 1, nti = ti;
            if (nkt == nk) { nkt = 0; ++nti; }
            if (s + 1 < S) WS_DMA(nti, nkt, (s + 1) & 1);          // that stage was last read before the previous barrier
            // table of the tile whose first step is s + 3: its buffer was last read two tiles ago; it lands with this slot's wait
            // and is read from slot s + 2 on
            if (MIX && kt == nk - 3 && ti + 1 < ntl) WS_MIXDMA(ti + 1);
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
                    MFMA6(acc1[mt], b0h, b0m, b0l, a0h, a0m, a0l)       // data x matrix
                    const bf16x8 a1h = *reinterpret_cast<const bf16x8 *>(p + IMG);
                    const bf16x8 a1m = *reinterpret_cast<const bf16x8 *>(p + IMG + PIECE);
                    const bf16x8 a1l = *reinterpret_cast<const bf16x8 *>(p + IMG + 2 * PIECE);
                    MFMA6(acc2[mt], b1h, b1m, b1l, a1h, a1m, a1l)
                }
            }
            // the DMA of the next stage has landed (and the previous tile's stores have drained) before this tile's stores are
            // queued behind it: nothing this wave waits for is ever younger than a store
            if (s + 1 < S) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (kt == nk - 1) {
                // epilogue of tile ti.  Register quad q of an accumulator = wavelengths 8 q + 4 h + (0..3) of the wave's 32
                // (packed: quads 0, 1 = the 16 wavelengths under the first variant's fold, quads 2, 3 = under the second's)
                const int t = (int)blockIdx.x + ti * (int)gridDim.x;
                const int tx = t % tilesX, ty = (t / tilesX) % tilesY;
                const long bz = t / (tilesX * tilesY);
                const int n0 = tx * TN, em0 = ty * 128;
                float *D0 = g.dst[0] + bz * g.sC + n0;
                float *D1 = (g.dst[1] ? g.dst[1] : g.dst[0]) + bz * g.sC + n0;
                float *DA = packed ? g.dst_alt + bz * g.sC + n0 : D0;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = em0 + mt * 32 + l31;
                    const bool ok = row < g.rvalid, okm = ok && row >= 1 && 2 * row != g.Rn;
                    const long ro = (long)row * g.ldc, rm = (long)(g.Rn - row) * g.ldc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool v2 = packed && q >= 2;                       // second variant: products swapped (see dft_rx3.hip)
                        const float e00 = v2 ? g.e_alt[1] : g.e00, e01 = v2 ? g.e_alt[0] : g.e01;
                        const float e10 = v2 ? g.e_alt[3] : g.e10, e11 = v2 ? g.e_alt[2] : g.e11;
                        const int col = packed ? cw * 16 + 8 * (q & 1) + 4 * h : cw * 32 + 8 * q + 4 * h;
                        float *d0 = (v2 ? DA : D0) + col;
                        const float a0 = acc1[mt][4 * q], a1 = acc1[mt][4 * q + 1], a2 = acc1[mt][4 * q + 2], a3 = acc1[mt][4 * q + 3];
                        const float b0 = acc2[mt][4 * q], b1 = acc2[mt][4 * q + 1], b2 = acc2[mt][4 * q + 2], b3 = acc2[mt][4 * q + 3];
                        if (WS_EXP & 4) {
                            asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
                        } else if (g.mode == 0) {
                            if (ok) *reinterpret_cast<float4 *>(d0 + ro) = make_float4(e00 * a0 + e01 * b0, e00 * a1 + e01 * b1, e00 * a2 + e01 * b2, e00 * a3 + e01 * b3);
                            if (okm) *reinterpret_cast<float4 *>(d0 + rm) = make_float4(e10 * a0 + e11 * b0, e10 * a1 + e11 * b1, e10 * a2 + e11 * b2, e10 * a3 + e11 * b3);
                        } else {
                            if (ok) {
                                *reinterpret_cast<float4 *>(D0 + col + ro) = make_float4(e00 * a0, e00 * a1, e00 * a2, e00 * a3);
                                *reinterpret_cast<float4 *>(D1 + col + ro) = make_float4(e11 * b0, e11 * b1, e11 * b2, e11 * b3);
                            }
                        }
                    }
                }
            }
            if (s + 1 < S) WS_BARRIER();
            kt = nkt;
            ti = nti;
        }
#undef WS_DMA
#undef WS_MIXDMA
        return;
    }

    // ================================================================== producers
    const int kin = g.Kn / 2 + 1;
    float rx[RING][8], ri[RING][8], rq[RING][8], rp[RING][8];     // raw rows (stream 0 / 1) and their mirror rows, per ring slot
    float4 tw = make_float4(0.f, 0.f, 0.f, 0.f);                   // template weights of the lane's column (MIX)
    int hv = h;

    // load cursor: k-step lstep of this workgroup's stream = k-step lkt of local tile li; it stops at the last step (the loads
    // behind the end of the stream repeat that step: no branch around a vector load, so the compiler keeps counted waits)
    int li = 0, lkt = 0, lstep = 0;
    const char *LB0 = nullptr, *LB1 = nullptr;
#define WS_LSETUP()                                                                                                 \
    {                                                                                                               \
        const int t_ = (int)blockIdx.x + li * (int)gridDim.x;                                                       \
        const long bz_ = t_ / (tilesX * tilesY);                                                                    \
        const int n0_ = (t_ % tilesX) * TN;                                                                         \
        LB0 = reinterpret_cast<const char *>(g.src[0] + bz_ * g.sB + n0_);                                          \
        LB1 = reinterpret_cast<const char *>(g.src[1] + bz_ * g.sB + n0_);                                          \
    }
    // raw loads of this lane's 8 k of k-step lkt (k = 16 kt + 8 h + j) and of their mirror rows; branch-free so that all of
    // them are in flight together.  Row part of every address in 64-bit scalar pointers, lane part in one small VGPR.
    // Mirror row of k is (Kn - 16 kt - 8 - j) + 8 (1 - h); read unconditionally (its weight is zero where there is no mirror),
    // except k = 0 whose "mirror" Kn may not exist.
#define WS_LOAD(r_)                                                                                                 \
    {                                                                                                               \
        const char *rk0 = LB0 + (long)(lkt * BK) * ldbB, *rk1 = LB1 + (long)(lkt * BK) * ldbB;                      \
        const char *rp0 = LB0 + (long)(g.Kn - lkt * BK - 8) * ldbB, *rp1 = LB1 + (long)(g.Kn - lkt * BK - 8) * ldbB; \
        const unsigned vk = (unsigned)(8 * hv) * ldb4 + c4, vp = (unsigned)(8 * (1 - hv)) * ldb4 + c4;              \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                            \
            if (WS_EXP & 1) { rx[r_][j] = ri[r_][j] = rq[r_][j] = rp[r_][j] = (float)(lstep + j); continue; }     \
            rx[r_][j] = *reinterpret_cast<const float *>(rk0 + j * ldbB + vk);                                     \
            if (KIND != 1) ri[r_][j] = *reinterpret_cast<const float *>(rk1 + j * ldbB + vk);                      \
            if (KIND != 2) {                                                                                       \
                const unsigned q = (j == 0 && lkt == 0) ? c4 : vp;                                                 \
                rq[r_][j] = *reinterpret_cast<const float *>(rp0 - j * ldbB + q);                                  \
                if (KIND != 1) rp[r_][j] = *reinterpret_cast<const float *>(rp1 - j * ldbB + q);                   \
            }                                                                                                      \
        }                                                                                                          \
        if (lstep + 1 < S) {                                                                                       \
            ++lstep;                                                                                               \
            if (++lkt == nk) { lkt = 0; ++li; WS_LSETUP(); }                                                       \
        }                                                                                                          \
    }
    // fold (and mix) ring slot r_ = k-step fkt of tile fi into the two data streams, split, store as B fragments of stage st_
#define WS_FOLD(r_, st_)                                                                                            \
    {                                                                                                               \
        const float4 *mb = mtab + (fi & 1) * mixs;                                                                  \
        if (MIX && fkt == 0) tw = mb[mixn + lcol];                                                                  \
        float x0[8], x1[8];                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                            \
            const int k = fkt * BK + 8 * hv + j;                                                                    \
            float ar = rx[r_][j], ai = (KIND == 1) ? rx[r_][j] : ri[r_][j];                                         \
            if (MIX) {                                                                                             \
                const float4 mr = mb[2 * k], mi = mb[2 * k + 1];                                                   \
                const float sr = tw.x * mr.x + tw.y * mr.y + tw.z * mr.z + tw.w * mr.w;                            \
                const float si = tw.x * mi.x + tw.y * mi.y + tw.z * mi.z + tw.w * mi.w;                            \
                const float hr = ar, hi = ai;                                                                      \
                ar = hr * sr - hi * si;                                                                            \
                ai = hr * si + hi * sr;                                                                            \
            }                                                                                                      \
            if (KIND == 2) {                                                                                       \
                x0[j] = ar;                                                                                        \
                x1[j] = ai;                                                                                        \
            } else {                                                                                               \
                const bool pv = (k >= 1) && (k < kin) && (2 * k != g.Kn);                                          \
                float br = rq[r_][j], bi = (KIND == 1) ? rq[r_][j] : rp[r_][j];                                     \
                if (MIX) {                                                                                         \
                    const int kp = pv ? g.Kn - k : k;                                                              \
                    const float4 mr = mb[2 * kp], mi = mb[2 * kp + 1];                                             \
                    const float sr = tw.x * mr.x + tw.y * mr.y + tw.z * mr.z + tw.w * mr.w;                        \
                    const float si = tw.x * mi.x + tw.y * mi.y + tw.z * mi.z + tw.w * mi.w;                        \
                    const float hr = br, hi = bi;                                                                  \
                    br = hr * sr - hi * si;                                                                        \
                    bi = hr * si + hi * sr;                                                                        \
                }                                                                                                  \
                const float g0 = var ? g.fold_alt[0] : g.fold[0], g1 = var ? g.fold_alt[1] : g.fold[1];            \
                const float f0 = pv ? g0 : 0.f, f1 = pv ? g1 : 0.f;                                                \
                const float w0 = (!pv && g0 < 0.f) ? 0.f : 1.f, w1 = (!pv && g1 < 0.f) ? 0.f : 1.f;                \
                const float s0 = w0 * ar + f0 * br, s1v = w1 * ai + f1 * bi;                                       \
                x0[j] = var ? s1v : s0;       /* the second variant pairs its streams with the other matrix */      \
                x1[j] = var ? s0 : s1v;                                                                            \
            }                                                                                                      \
        }                                                                                                          \
        uint4 f0h, f0m, f0l, f1h, f1m, f1l;                                                                        \
        if (WS_EXP & 8) {                                                                                          \
            f0h = make_uint4(__float_as_uint(rx[r_][0]), __float_as_uint(rx[r_][1]), __float_as_uint(rx[r_][2]), __float_as_uint(rx[r_][3])); \
            f0m = make_uint4(__float_as_uint(rx[r_][4]), __float_as_uint(rx[r_][5]), __float_as_uint(rx[r_][6]), __float_as_uint(rx[r_][7])); \
            f0l = f0h; f1h = f0m; f1m = f0h; f1l = f0m;                                                            \
            if (KIND != 1) { f1h.x ^= __float_as_uint(ri[r_][0] + ri[r_][1] + ri[r_][2] + ri[r_][3] + ri[r_][4] + ri[r_][5] + ri[r_][6] + ri[r_][7]); } \
            if (KIND != 2) { f1m.x ^= __float_as_uint(rq[r_][0] + rq[r_][1] + rq[r_][2] + rq[r_][3] + rq[r_][4] + rq[r_][5] + rq[r_][6] + rq[r_][7]); } \
            if (KIND == 0) { f1l.x ^= __float_as_uint(rp[r_][0] + rp[r_][1] + rp[r_][2] + rp[r_][3] + rp[r_][4] + rp[r_][5] + rp[r_][6] + rp[r_][7]); } \
        } else {                                                                                                   \
        split8(x0, f0h, f0m, f0l);                                                                                  \
        split8(x1, f1h, f1m, f1l);                                                                                  \
        }                                                                                                          \
        uint4 *wb = reinterpret_cast<uint4 *>(ldsB + (st_) * BBUF + cw * BWAVE) + lane;                            \
        wb[0] = f0h; wb[64] = f0m; wb[128] = f0l; wb[192] = f1h; wb[256] = f1m; wb[320] = f1l;                     \
        if (++fkt == nk) { fkt = 0; ++fi; }                                                                        \
    }

    // ---- slot -2: the first RING k-steps of loads in flight
    WS_LSETUP();
#pragma unroll
    for (int r = 0; r < RING; ++r) WS_LOAD(r);
    WS_BARRIER();
    // ---- slots -1 .. S-2: in slot f - 1 fold k-step f (ring slot f % RING), then refill that ring slot with k-step f + RING
    int fi = 0, fkt = 0, f0 = 0;
    for (; f0 + RING <= S; f0 += RING) {
        asm volatile("" : "+v"(hv));   // keeps the per-lane fold selectors from being hoisted out of the loop
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            WS_FOLD(r, (f0 + r) & 1);
            WS_LOAD(r);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WS_BARRIER();
        }
    }
#pragma unroll
    for (int r = 0; r < RING - 1; ++r)      // the last S % RING k-steps: nothing left to load
        if (f0 + r < S) {
            WS_FOLD(r, (f0 + r) & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WS_BARRIER();
        }
#undef WS_LSETUP
#undef WS_LOAD
#undef WS_FOLD
}

}  // namespace

// Whether launch_dft_ws can run this pass (the callers fall back to launch_dft_rx3 otherwise)
bool dft_ws_can(const DftRx3Args &g) {
    if (g.MP % 128 || g.KP % BK || g.batch < 1) return false;
    if (g.nvar != 1 && !(g.nvar == 2 && g.packed)) return false;
    const bool folded = g.fold[0] != 0.f || g.fold[1] != 0.f || (g.nvar == 2 && (g.fold_alt[0] != 0.f || g.fold_alt[1] != 0.f));
    const int kind = !folded ? 2 : (g.src[0] == g.src[1] ? 1 : 0);
    if (g.nvar == 2) {
        const long dalt = g.dst_alt ? (long)(g.dst_alt - g.dst[0]) * 4 : -1;
        if (kind != 0 || g.mode != 0 || g.N % 64 || dalt < 0 || dalt >= 2147483648L || !g.A_alt[0]) return false;
        // the packed form keeps A[0], A[1] for both variants: the second variant's matrices must be the first one's, swapped
        if (g.A_alt[0] != g.A[1] || g.A_alt[1] != g.A[0]) return false;
    } else if (g.N % 128) return false;
    if (g.mhat) {      // fused spectral mix: needs the table built by launch_dft_ws_mix_table and whole-DMA table sizes
        if (kind != 0 || g.nvar != 2 || g.LP % 128 || g.T < 1 || g.T > 4 || g.KP / BK < 3 || !g.mixtab || !g.tplT ||
            g.mix_rows % 32 || g.mix_rows < g.Kn || g.mix_rows < g.KP || g.mix_rows > MIX_ROWS_MAX)
            return false;
    }
    return true;
}

int launch_dft_ws(hipStream_t stream, const DftRx3Args &g) {
    if (!dft_ws_can(g)) return (int)hipErrorInvalidValue;
    if (g.mode == 1 && !g.dst[1]) return (int)hipErrorInvalidValue;
    if (8.0 * (double)g.ldb * 4.0 + 1024.0 >= 4294967296.0 || 4.0 * (double)g.ldc * 4.0 + 1024.0 >= 4294967296.0) return (int)hipErrorInvalidValue;
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
    DftRx3Args a = g;
    a.packed = (g.nvar == 2) ? 1 : 0;
    a.strided = 0;
    const long ntile = (long)(g.N / (a.packed ? 64 : 128)) * (g.MP / 128) * g.batch;
    dim3 grid((unsigned)(ntile < cus_of[dev] ? ntile : cus_of[dev]));
    const size_t mix_bytes = g.mhat ? (size_t)2 * (2 * g.mix_rows + 128) * sizeof(float4) : 0;
    static unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    if (g.mhat) {
        if (int e = ensure_dynamic_lds(dft_ws_kernel<0, true>, LDS_MAIN + (size_t)2 * (2 * MIX_ROWS_MAX + 128) * sizeof(float4), d3)) return e;
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

int dft_ws_mix_rows(int Kn, int KP) { return ((Kn > KP ? Kn : KP) + 31) / 32 * 32; }

int launch_dft_ws_mix_table(hipStream_t stream, const float *mhat, float *mixtab, int T, int Kn, int nkb, long PL, long KBP, int mix_rows) {
    if (T < 1 || T > 4 || nkb < 1 || mix_rows < Kn || mix_rows % 32) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)((nkb + 63) / 64), (unsigned)((2 * mix_rows + 3) / 4));
    hipLaunchKernelGGL(mix_table_kernel, grid, dim3(256), 0, stream, mhat, reinterpret_cast<float4 *>(mixtab), T, Kn, nkb, PL, KBP, mix_rows);
    return (int)hipGetLastError();
}
