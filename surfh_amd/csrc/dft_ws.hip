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
constexpr int RING = 3;                       // k-steps of raw loads a producer holds (RING - 1 in flight behind the one in use)
constexpr int MIXE = 4;                       // mix-table entries a producer thread stages per tile (256 threads): Kn <= 512
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
    float4 *mtab = reinterpret_cast<float4 *>(lds + NA * ABUF + NB * BBUF);   // [2][mixn]: mhat column of a tile, [k][re/im] x 4 templates
    const int mixn = (g.Kn > g.KP ? g.Kn : g.KP) * 2;

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
        f32x16 acc1[4], acc2[4];
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
            bool many_stores = false;
            if (kt == nk - 1) {
                // epilogue of tile ti: uniform base + 32-bit lane offset (the launcher checks the pitches)
                const int t = (int)blockIdx.x + ti * (int)gridDim.x;
                const int tx = t % tilesX, ty = (t / tilesX) % tilesY;
                const long bz = t / (tilesX * tilesY);
                const int n0 = tx * TN, em0 = ty * 128;
                // packed: a second-variant lane holds (acc1, acc2) = (A[0] X_second, A[1] X_first), i.e. that variant's products swapped
                const float e00 = var ? g.e_alt[1] : g.e00, e01 = var ? g.e_alt[0] : g.e01;
                const float e10 = var ? g.e_alt[3] : g.e10, e11 = var ? g.e_alt[2] : g.e11;
                char *D0 = reinterpret_cast<char *>(g.dst[0] + bz * g.sC + n0);
                const unsigned dvar = var ? (unsigned)((g.dst_alt - g.dst[0]) * 4) : 0u;      // second variant's array, as a lane offset
                char *D1 = reinterpret_cast<char *>((g.dst[1] ? g.dst[1] : g.dst[0]) + bz * g.sC + n0);
                char *Dk0 = D0 + (long)em0 * ldcB, *Dk1 = D1 + (long)em0 * ldcB, *Dm0 = D0 + (long)(g.Rn - em0 - 4) * ldcB;
                const unsigned lo = (unsigned)(4 * h) * ldc4 + c4 + dvar, lm = (unsigned)(4 * (1 - h)) * ldc4 + c4 + dvar;
                const long s1 = ldcB, s5 = 5 * ldcB;
                many_stores = g.rvalid - em0 >= 126;               // then at least 63 store instructions follow the DMA above
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = em0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float a1 = acc1[mt][r], a2 = acc2[mt][r];
                        if (row < g.rvalid) {
                            if (g.mode == 0) {
                                *reinterpret_cast<float *>(Dk0 + lo) = e00 * a1 + e01 * a2;
                                if (row >= 1 && 2 * row != g.Rn) *reinterpret_cast<float *>(Dm0 + lm) = e10 * a1 + e11 * a2;
                            } else {
                                *reinterpret_cast<float *>(Dk0 + lo) = e00 * a1;
                                *reinterpret_cast<float *>(Dk1 + lo) = e11 * a2;
                            }
                        }
                        const long st = ((r & 3) == 3) ? s5 : s1;
                        Dk0 += st; Dk1 += st; Dm0 -= st;
                    }
            }
            if (s + 1 < S) {
                // the DMA of the next stage must have landed; the stores behind it may keep draining
                if (many_stores) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WS_BARRIER();
            }
            kt = nkt;
            ti = nti;
        }
#undef WS_DMA
        return;
    }

    // ================================================================== producers
    const int ptid = tid - 256;
    const int kin = g.Kn / 2 + 1;
    float rx[RING][8], ri[RING][8], rq[RING][8], rp[RING][8];     // raw rows (stream 0 / 1) and their mirror rows, per ring slot
    float4 twr[RING];                                              // template weights of the lane's column, loaded with a tile's first k-step
    float4 tw = make_float4(0.f, 0.f, 0.f, 0.f);
    float mt4[MIXE][4];                                            // staged mix-table entries of the next tile
    int hv = h;

    // load cursor (step lf): tile li, k-step lkt, source bases of that tile
    int li = 0, lkt = 0;
    const char *LB0 = nullptr, *LB1 = nullptr;
    int ln0 = 0;
#define WS_LSETUP()                                                                                                 \
    {                                                                                                               \
        const int t_ = (int)blockIdx.x + li * (int)gridDim.x;                                                       \
        const int tx_ = t_ % tilesX;                                                                                \
        const long bz_ = t_ / (tilesX * tilesY);                                                                    \
        ln0 = tx_ * TN;                                                                                             \
        LB0 = reinterpret_cast<const char *>(g.src[0] + bz_ * g.sB + ln0);                                          \
        LB1 = reinterpret_cast<const char *>(g.src[1] + bz_ * g.sB + ln0);                                          \
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
            rx[r_][j] = *reinterpret_cast<const float *>(rk0 + j * ldbB + vk);                                     \
            if (KIND != 1) ri[r_][j] = *reinterpret_cast<const float *>(rk1 + j * ldbB + vk);                      \
            if (KIND != 2) {                                                                                       \
                const unsigned q = (j == 0 && lkt == 0) ? c4 : vp;                                                 \
                rq[r_][j] = *reinterpret_cast<const float *>(rp0 - j * ldbB + q);                                  \
                if (KIND != 1) rp[r_][j] = *reinterpret_cast<const float *>(rp1 - j * ldbB + q);                   \
            }                                                                                                      \
        }                                                                                                          \
        if (MIX && lkt == 0) {                                                                                     \
            const int l = (ln0 % g.LP) + lcol;                                                                     \
            float t4[4];                                                                                           \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) t4[t] = (t < g.T) ? g.tpl[(long)t * g.LP + l] : 0.f;     \
            twr[r_] = make_float4(t4[0], t4[1], t4[2], t4[3]);                                                     \
        }                                                                                                          \
        if (++lkt == nk) { lkt = 0; ++li; if (li < ntl) WS_LSETUP(); }                                             \
    }
    // mix table of local tile i_: global loads into registers (issue), registers into LDS buffer i_ & 1 (write)
#define WS_MIX_ISSUE(i_)                                                                                            \
    {                                                                                                               \
        const int t_ = (int)blockIdx.x + (i_) * (int)gridDim.x;                                                     \
        const int kb = ((t_ % tilesX) * TN) / g.LP;                                                                 \
        _Pragma("unroll") for (int u = 0; u < MIXE; ++u) {                                                         \
            const int e = ptid + 256 * u, k = e >> 1, c = e & 1;                                                   \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                          \
                mt4[u][t] = (e < mixn && t < g.T && k < g.Kn) ? g.mhat[((long)t * 2 + c) * g.PL + (long)k * g.KBP + kb] : 0.f; \
        }                                                                                                          \
    }
#define WS_MIX_WRITE(i_)                                                                                            \
    {                                                                                                               \
        float4 *mb = mtab + ((i_) & 1) * mixn;                                                                      \
        _Pragma("unroll") for (int u = 0; u < MIXE; ++u) {                                                         \
            const int e = ptid + 256 * u;                                                                          \
            if (e < mixn) mb[e] = make_float4(mt4[u][0], mt4[u][1], mt4[u][2], mt4[u][3]);                         \
        }                                                                                                          \
    }
    // fold (and mix) ring slot r_ = k-step fkt of tile fi into the two data streams, split, store as B fragments of stage st_
#define WS_FOLD(r_, st_)                                                                                            \
    {                                                                                                               \
        if (MIX && fkt == 0) tw = twr[r_];                                                                          \
        const float4 *mb = mtab + (fi & 1) * mixn;                                                                  \
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
        split8(x0, f0h, f0m, f0l);                                                                                  \
        split8(x1, f1h, f1m, f1l);                                                                                  \
        uint4 *wb = reinterpret_cast<uint4 *>(ldsB + (st_) * BBUF + cw * BWAVE) + lane;                            \
        wb[0] = f0h; wb[64] = f0m; wb[128] = f0l; wb[192] = f1h; wb[256] = f1m; wb[320] = f1l;                     \
        if (++fkt == nk) { fkt = 0; ++fi; }                                                                        \
    }

    // ---- slot -2: first RING k-steps of loads in flight, mix table of tile 0
    if (ntl > 0) WS_LSETUP();
#pragma unroll
    for (int r = 0; r < RING; ++r)
        if (r < S) WS_LOAD(r);
    if (MIX && ntl > 0) {
        WS_MIX_ISSUE(0);
        WS_MIX_WRITE(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WS_BARRIER();
    // ---- slots -1 .. S-2: in slot f - 1 fold k-step f (ring slot f % RING), then refill that ring slot with k-step f + RING
    int fi = 0, fkt = 0;
    for (int f0 = 0; f0 < S; f0 += RING) {
        asm volatile("" : "+v"(hv));   // keeps the per-lane fold selectors from being hoisted out of the loop
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            const int f = f0 + r;
            if (f < S) {
                WS_FOLD(r, f & 1);
                if (f + RING < S) WS_LOAD(r);
                if (MIX) {
                    // tile i's table is read from slot (first step of i) - 1 on: registers filled two slots, LDS one slot earlier
                    const int a = f + 2, b = f + 1;          // slots f - 1 = s_i0 - 3 and s_i0 - 2  <=>  s_i0 = f + 2 and f + 1
                    if (a % nk == 0 && a < S) WS_MIX_ISSUE(a / nk);
                    if (b % nk == 0 && b < S && b > 0) WS_MIX_WRITE(b / nk);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                WS_BARRIER();
            }
        }
    }
#undef WS_LSETUP
#undef WS_LOAD
#undef WS_MIX_ISSUE
#undef WS_MIX_WRITE
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
    if (g.mhat) {
        const int mixn = (g.Kn > g.KP ? g.Kn : g.KP) * 2;
        if (kind != 0 || g.nvar != 2 || g.LP % 128 || g.T < 1 || g.T > 4 || mixn > 256 * MIXE || g.KP / BK < 3) return false;
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
    const int mixn = (g.Kn > g.KP ? g.Kn : g.KP) * 2;
    const size_t mix_bytes = g.mhat ? (size_t)2 * mixn * sizeof(float4) : 0;
    static unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    if (g.mhat) {
        if (int e = ensure_dynamic_lds(dft_ws_kernel<0, true>, LDS_MAIN + (size_t)2 * 256 * MIXE * sizeof(float4), d3)) return e;
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
