// Two-piece fp16 DFT pass with LDS-resident matrices, decoupled waves and interleaved complex arrays (see dft_h2.h).
// One persistent workgroup of eight waves per CU; a wave owns all 128 output rows of its 32 lane-columns (32 real
// columns, or 16 complex columns x 2 components), i.e. 2 products x 4 row tiles x 16 = 128 accumulator registers,
// and walks its own sequence of tiles: after the initial copy of the matrix image there is no workgroup barrier
// (the fused spectral mix adds one per change of k_beta).  Each wave pipelines its stream of k-steps: raw loads two
// steps ahead, fold / scale / split one step ahead, MFMAs on the current step; a tile's stores are issued after its
// last MFMA group, behind loads that are already in flight.
#include "dft_h2.h"
#include "lds_attr.h"
#include <cmath>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef H2_EXP
#define H2_EXP 0        // tools/exp: 2 no stores, 4 no MFMAs, 8 no loads
#endif

namespace {

constexpr int BK = 16, RS = 16;
constexpr int PIECE = 128 * RS;                       // halfs of one (matrix, piece, k-step) block: 4 KB
constexpr int KT = DFT_H2_KT;
constexpr int IMG_HALFS = (int)DFT_H2_IMAGE_HALFS;    // 65536 halfs = 128 KB
constexpr int MIX_ROWS = 256;                         // rows of one spectral-mix table ([k][re/im] float4): 8 KB
constexpr size_t LDS_IMG = (size_t)IMG_HALFS * 2;
constexpr size_t LDS_MIX = (size_t)2 * MIX_ROWS * 2 * sizeof(float4);
#ifndef H2_WAVES
#define H2_WAVES 8
#endif
constexpr int NWAVES = H2_WAVES, NTHREADS = 64 * NWAVES;
constexpr int E_TARGET = 10, E_LIMIT = 15;            // a column's largest scaled magnitude: set below 2^10, rescaled at 2^15

#define MFMA3(acc_, ah_, al_, bh_, bl_)                                             \
    {                                                                               \
        f32x16 c_ = acc_;                                                           \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_, bh_, c_, 0, 0, 0);         \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bl_, c_, 0, 0, 0);         \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bh_, c_, 0, 0, 0);         \
        acc_ = c_;                                                                  \
    }

// 8 scaled values -> (hi, lo) fp16 fragments
__device__ __forceinline__ void split8h(const float (&x)[8], int e, f16x8 &fh, f16x8 &fl) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xs = __builtin_amdgcn_ldexpf(x[j], e);
        const _Float16 hh = (_Float16)xs;
        fh[j] = hh;
        fl[j] = (_Float16)(xs - (float)hh);
    }
}

// the value held by the neighbouring lane (lane ^ 1): the other component of the same complex column
__device__ __forceinline__ float pair_swap(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
}

template <int KIND, bool MIX>
__global__ __launch_bounds__(NTHREADS) void dft_h2_kernel(DftH2Args g, const uint4 *__restrict__ img, int kA, int NS, long NT) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile indices and descriptors stay scalar
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int TNW = (KIND == 0) ? 16 : 32;                 // columns of a wave tile (complex columns for kind 0)
    constexpr int SRC_T = (KIND == 2) ? 64 : 32, DST_T = (KIND == 1) ? 64 : 32;   // floats of a tile's row segment
    const int var = (KIND == 0) ? (l31 & 1) : 0;               // kind 0: the component this lane owns
    const int lcol = (KIND == 0) ? (l31 >> 1) : l31;
    const int tilesX = g.N / TNW;
    const unsigned ldb4 = (unsigned)(g.ldb * 4), ldc4 = (unsigned)(g.ldc * 4);
    const unsigned c4 = (unsigned)l31 * (KIND == 2 ? 8u : 4u), d4 = (unsigned)l31 * (KIND == 1 ? 8u : 4u);
    const int nk = g.KP / BK;
    const int kin = g.Kn / 2 + 1;
    float4 *mixbuf = reinterpret_cast<float4 *>(lds + IMG_HALFS);

    // the matrix image: global -> LDS, once
    {
        uint4 *l4 = reinterpret_cast<uint4 *>(lds);
        for (int i = tid; i < IMG_HALFS / 8; i += NTHREADS) l4[i] = img[i];
    }
    // this workgroup's contiguous range of super-tiles (8 wave tiles each); wave w takes tile 8 s + w
    const int s0 = (int)((long)NS * blockIdx.x / gridDim.x), s1 = (int)((long)NS * (blockIdx.x + 1) / gridDim.x);
    int ntw = s1 - s0;
    if (ntw > 0 && (long)NWAVES * (s1 - 1) + wave >= NT) --ntw;

    // fused spectral mix: the (k, kb) column of mhat, all k, as [k][template](re, im) in LDS, one table per kb parity
    const float4 *mtab = mixbuf;
    float4 tw = make_float4(0.f, 0.f, 0.f, 0.f);
    int mix_kb = -1;
#define H2_MIXTAB(kb_)                                                                                          \
    {                                                                                                           \
        float4 *mt_ = mixbuf + ((kb_) & 1) * (MIX_ROWS * 2);                                                     \
        const int ne = (g.Kn > g.KP ? g.Kn : g.KP) * 2;                                                         \
        const float msc_ = ((kb_) == 0 || 2 * (kb_) == g.mix_Nb) ? g.mhat_self : g.mhat_pair;                   \
        for (int e_ = tid; e_ < ne; e_ += NTHREADS) {                                                           \
            const int k = e_ >> 1, tp = e_ & 1;        /* float4 = (re, im) of templates 2 tp and 2 tp + 1 */   \
            float v[4];                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
                const int t = 2 * tp + (i >> 1), c = i & 1;                                                     \
                v[i] = (t < g.T && k < g.Kn) ? g.mhat[((long)t * 2 + c) * g.PL + (long)k * g.KBP + (kb_)] : 0.f; \
            }                                                                                                   \
            mt_[e_] = make_float4(v[0] * msc_, v[1] * msc_, v[2] * msc_, v[3] * msc_);                          \
        }                                                                                                       \
        mtab = mt_;                                                                                             \
        mix_kb = (kb_);                                                                                         \
    }
    // per-tile state of the fold (MIX only): template weights of this lane's wavelength, table of the tile's kb;
    // a change of kb is reached by all eight waves of the workgroup (same tile sequence): the table buffer of parity
    // kb & 1 was last read two values of kb ago, i.e. before the previous barrier
#define H2_FSETUP(t_)                                                                                           \
    if (MIX) {                                                                                                  \
        const int fn0 = ((t_) % tilesX) * TNW;                                                                   \
        /* whole-cube launch: column n = kb LP + l; wavelength-chunk launch (batch > 1): batch entry = kb */     \
        const int kb = g.batch > 1 ? (t_) / tilesX : fn0 / g.LP;                                                \
        const int l = g.mix_l0 + (g.batch > 1 ? fn0 : fn0 % g.LP) + lcol;                                       \
        float t4[4];                                                                                            \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) t4[t] = (t < g.T) ? g.tpl[(long)t * g.LP + l] : 0.f;      \
        tw = make_float4(t4[0], t4[1], t4[2], t4[3]);                                                           \
        if (kb != mix_kb) {                                                                                     \
            H2_MIXTAB(kb);                                                                                      \
            __syncthreads();                                                                                    \
        }                                                                                                       \
    }

    // Addressing: buffer loads / stores -- a descriptor per tile (scalar), the row as a 32-bit scalar byte offset, the
    // lane part (k half, column) in one 32-bit VGPR: no 64-bit vector address arithmetic.  The launcher checks that
    // every row offset of a tile fits in 32 bits.
    __amdgpu_buffer_rsrc_t R0;
#define H2_RSRC(ptr_) __builtin_amdgcn_make_buffer_rsrc((void *)(ptr_), 0, 0xFFFFFFFF, 0x00020000)
#define H2_BLOAD(v_, s_) ((H2_EXP & 8) ? __uint_as_float((v_) + (s_)) : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R0, (int)(v_), (int)(s_), 0)))
#define H2_BLOAD2(v_, s_) __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(R0, (int)(v_), (int)(s_), 0))
#define H2_LSETUP(t_)                                                                                           \
    {                                                                                                           \
        const int tx = (t_) % tilesX;                                                                           \
        const long bz_ = (t_) / tilesX;                                                                         \
        R0 = H2_RSRC(g.src + bz_ * g.sB + (long)tx * SRC_T);                                                    \
    }
    // raw loads of this lane's 8 k of k-step kt_ (k = 16 kt + 8 h + j) and, for the folded kinds, of their mirror rows
    // (Kn - k): branch-free, all in flight together.  The mirror is read unconditionally (zero weight where there is
    // none), except k = 0 whose "mirror" row Kn may not exist
#define H2_LOAD(kt_)                                                                                            \
    {                                                                                                           \
        const unsigned sk = (unsigned)((kt_) * BK) * ldb4, sp = (unsigned)(g.Kn - (kt_) * BK - 8) * ldb4;       \
        const unsigned vk = (unsigned)(8 * hv) * ldb4 + c4, vp = (unsigned)(8 * (1 - hv)) * ldb4 + c4;          \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                         \
            if (KIND == 2) {                                                                                    \
                const f32x2 v2 = H2_BLOAD2(vk, sk + (unsigned)j * ldb4);                                        \
                xr[j] = v2[0];                                                                                  \
                qr[j] = v2[1];                                                                                  \
            } else {                                                                                            \
                xr[j] = H2_BLOAD(vk, sk + (unsigned)j * ldb4);                                                  \
                const unsigned q = (j == 0 && (kt_) == 0) ? c4 : vp;                                            \
                qr[j] = H2_BLOAD(q, sp - (unsigned)j * ldb4);                                                   \
            }                                                                                                   \
        }                                                                                                       \
    }
    // fold (and mix) the raw values into the two data streams of k-step kt_
#define H2_FOLD_BODY(kt_, PV_)                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                         \
            const int k = (kt_) * BK + 8 * hv + j;                                                              \
            const bool pv = PV_;                                                                                \
            float a = xr[j], b = qr[j];                                                                         \
            if (MIX) {      /* own component of (re + i im) * s: re' = re sr - im si, im' = im sr + re si */       \
                const float ap = pair_swap(a), bp = pair_swap(b);                                               \
                const int kp = pv ? g.Kn - k : k;                                                               \
                /* s = sum_t tw[t] (re, im)[t] as packed pairs: two v_pk_fma_f32 per table float4 */                 \
                const float4 m01 = mtab[2 * k], m23 = mtab[2 * k + 1], n01 = mtab[2 * kp], n23 = mtab[2 * kp + 1]; \
                const f32x2 sv = tw.x * f32x2{m01.x, m01.y} + tw.y * f32x2{m01.z, m01.w} + tw.z * f32x2{m23.x, m23.y} + tw.w * f32x2{m23.z, m23.w}; \
                const f32x2 uv = tw.x * f32x2{n01.x, n01.y} + tw.y * f32x2{n01.z, n01.w} + tw.z * f32x2{n23.x, n23.y} + tw.w * f32x2{n23.z, n23.w}; \
                a = a * sv[0] + sgv * (ap * sv[1]);                                                             \
                b = b * uv[0] + sgv * (bp * uv[1]);                                                             \
            }                                                                                                   \
            const float ev = a + (pv ? b : 0.f), od = pv ? a - b : 0.f;                                         \
            x0[j] = ev;                                                                                         \
            x1[j] = (KIND == 0) ? pair_swap(od) : od;          /* kind 0: the odd part of the other component */   \
        }
#define H2_FOLD(kt_)                                                                                            \
    {                                                                                                           \
        if (KIND == 2) {                                                                                        \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                     \
                x0[j] = xr[j];                                                                                  \
                x1[j] = qr[j];                                                                                  \
            }                                                                                                   \
        } else if ((kt_) >= 1 && ((kt_) + 1) * BK <= kin && (g.Kn & 1)) {                                       \
            /* every k of the step has its mirror row (wave-uniform; all but the first and the last step of an odd length): \
               the same values without the per-element tests */                                                 \
            H2_FOLD_BODY(kt_, true)                                                                             \
        } else {                                                                                                \
            H2_FOLD_BODY(kt_, (k >= 1) && (k < kin) && (2 * k != g.Kn))                                         \
        }                                                                                                       \
    }
    // exponent p of this column's largest folded magnitude (|x| < 2^p), over both streams and both k halves
#define H2_MAXEXP(p_)                                                                                           \
    {                                                                                                           \
        float m_ = 0.f;                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) m_ = fmaxf(m_, fmaxf(fabsf(x0[j]), fabsf(x1[j])));        \
        const unsigned mu_ = __float_as_uint(m_);                                                               \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(mu_, mu_, false, false);                              \
        m_ = fmaxf(m_, fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1])));                                \
        p_ = __builtin_amdgcn_frexp_expf(m_);                                                                   \
    }
#define H2_MFMA(m_, kt_, acc_, bh_, bl_)                                                                        \
    {                                                                                                           \
        const unsigned short *ra = lds + ((m_) * 2 * KT + (kt_)) * PIECE + l31 * RS + 8 * (h ^ ((l31 >> 3) & 1)); \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                      \
            const unsigned short *p = ra + mt * 32 * RS;                                                        \
            const f16x8 ah = *reinterpret_cast<const f16x8 *>(p);                                               \
            const f16x8 al = *reinterpret_cast<const f16x8 *>(p + KT * PIECE);                                  \
            if (!(H2_EXP & 4)) MFMA3(acc_[mt], ah, al, bh_, bl_)                                                \
            else acc_[mt][0] += (float)bh_[mt] + (float)bl_[mt] + (float)ah[0] + (float)al[0];                  \
        }                                                                                                       \
    }

    int hv = h;
    const float sgv = var ? 1.f : -1.f;
    float xr[8], qr[8];                // raw values: rows k and mirror rows (kind 2: re and im of rows k)
    float x0[8], x1[8];
    f16x8 c0h, c0l, c1h, c1l;          // fragments of the current k-step
    f32x16 acc1[4], acc2[4];

    if (MIX) {                         // first table (the barrier also publishes the matrix image)
        if (ntw > 0) {
            const int t0_ = NWAVES * s0 + wave;
            const int kb0 = g.batch > 1 ? t0_ / tilesX : ((t0_ % tilesX) * TNW) / g.LP;
            H2_MIXTAB(kb0);
        }
    }
    __syncthreads();
    if (ntw <= 0) return;              // MIX: NT % 8 == 0, so all waves of a workgroup leave together

    // Software pipeline over this wave's stream of k-steps.  Inside a step: MFMAs of the first matrix, fold of the next
    // step's raw values, loads of the step after that, the first stream's fragments rebuilt in place, MFMAs of the
    // second matrix, the second stream's fragments rebuilt.  At a tile seam the next tile's first two k-steps are
    // requested before the stores of the epilogue.
    // tiles are counted in the launch's own numbering vt (super-tile, wave); with a list of super-tiles the real tile follows from it
    const int *vl = g.vlist;
#define H2_RT(vt_) (vl ? NWAVES * vl[(vt_) >> 3] + ((vt_) & 7) : (vt_))
    int vt = NWAVES * s0 + wave;
    const int vtend = vt + NWAVES * ntw;               // this wave's tiles: vt, vt + 8, ... < vtend
    int tile = H2_RT(vt);
    int e = 0, en = 0;                                  // block exponents of the current / next k-step's column

    H2_LSETUP(tile);
    H2_LOAD(0);
    H2_FSETUP(tile);
    H2_FOLD(0);
    {
        int p;
        H2_MAXEXP(p);
        e = E_TARGET - p;
    }
    split8h(x0, e, c0h, c0l);
    split8h(x1, e, c1h, c1l);
    H2_LOAD(1);
    while (true) {
        asm volatile("" : "+v"(hv));   // keeps the per-lane fold selectors from being hoisted out of the tile loop
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] = acc2[i][r] = 0.f;
        const bool more = vt + NWAVES < vtend;
        const int next = more ? H2_RT(vt + NWAVES) : tile;
        // per-tile limits (wave-uniform): k-steps that can be non-zero, output rows that anybody reads
        const int chunk_ = g.tabLP ? (int)(((long)(tile % tilesX) * TNW % g.tabLP) >> 7) : 0;
        const int nkt = g.ktab ? g.ktab[chunk_] : nk;
        const int mtn = g.rtab ? (g.rtab[chunk_] + 31) / 32 : 4;       // row tiles of 32 that anybody reads
        for (int kt = 0; kt + 1 < nkt; ++kt) {
            H2_MFMA(0, kt, acc1, c0h, c0l);
            H2_FOLD(kt + 1);           // raw holds k-step kt + 1 of this tile
            int p;
            H2_MAXEXP(p);
            const bool need = p + e > E_LIMIT;
            en = need ? E_TARGET - p : e;
            const int d = en - e;
            if (kt + 2 < nkt) {
                H2_LOAD(kt + 2);
            } else if (more) {
                H2_LSETUP(next);
                H2_LOAD(0);
            }
            split8h(x0, en, c0h, c0l);
            H2_MFMA(1, kt, acc2, c1h, c1l);
            split8h(x1, en, c1h, c1l);
            if (__builtin_amdgcn_ballot_w64(d != 0) != 0ull) {
                // a column's next k-step would overflow its fp16 range: its accumulators move to the lower exponent
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc1[mt][r] = __builtin_amdgcn_ldexpf(acc1[mt][r], d);
                        acc2[mt][r] = __builtin_amdgcn_ldexpf(acc2[mt][r], d);
                    }
            }
            e = en;
        }
        // last k-step of the tile; raw holds k-step 0 of the next tile (if any)
        H2_MFMA(0, nkt - 1, acc1, c0h, c0l);
        if (more) {
            H2_FSETUP(next);
            H2_FOLD(0);
            int p;
            H2_MAXEXP(p);
            en = E_TARGET - p;         // fresh exponent: the next tile's accumulators start from zero
            H2_LOAD(1);
            split8h(x0, en, c0h, c0l);
        }
        H2_MFMA(1, nkt - 1, acc2, c1h, c1l);
        if (more) split8h(x1, en, c1h, c1l);
        {
            // epilogue: descriptor of the tile's destination, rows as running scalar offsets
            const int tx = tile % tilesX;
            const long bz = tile / tilesX;
            const float f = __builtin_amdgcn_ldexpf(1.f, -e - kA);
            const float e0 = f * (var ? g.e_alt[0] : g.e[0]), e1 = f * (var ? g.e_alt[1] : g.e[1]);
            const float e2 = f * (var ? g.e_alt[2] : g.e[2]), e3 = f * (var ? g.e_alt[3] : g.e[3]);
            const __amdgpu_buffer_rsrc_t W0 = H2_RSRC(g.dst + bz * g.sC + (long)tx * DST_T);
            const unsigned lo = (unsigned)(4 * h) * ldc4 + d4, lm = (unsigned)(4 * (1 - h)) * ldc4 + d4;
            unsigned sk0 = 0u, sm0 = (unsigned)(g.Rn - 4) * ldc4;     // row / mirror row (Rn - 4 - row) byte offsets
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (mt >= mtn) break;          // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float a1 = acc1[mt][r], a2 = acc2[mt][r];
                    if (row < g.rvalid && (!(H2_EXP & 2) || g.batch < 0)) {
                        if (KIND == 1) {
                            const f32x2 v2 = {e0 * a1, e3 * a2};
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v2), W0, (int)lo, (int)sk0, 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e0 * a1 + e1 * a2), W0, (int)lo, (int)sk0, 0);
                            if (row >= 1 && 2 * row != g.Rn)
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e2 * a1 + e3 * a2), W0, (int)lm, (int)sm0, 0);
                        }
                    }
                    const unsigned st = ((r & 3) == 3) ? 5u * ldc4 : ldc4;
                    sk0 += st; sm0 -= st;
                }
            }
        }
        if (!more) break;
        e = en;
        tile = next;
        vt += NWAVES;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Adjoint tail, fused: the complex pass along alpha of rfft2 (kind 0) whose epilogue, instead of storing the spectrum
// Y[ka][kb][l], forms conj(H) Y and reduces it over the wavelengths with the template weights -- the reference's
// `sum_l tpl[t,l] conj(sotf) rfft2(cube)` (spectroModel.py:175-181) -- so the 1 GB spectrum is neither written nor read
// back and `specmix_adj_kernel` disappears.
//
// The MFMA operands swap roles: the data fragments go in as A, the matrix fragments as B.  The fragment registers are the
// same, but the accumulator then has lane = output row (k_alpha) and registers = the tile's 32 lane-columns, i.e. a
// register quad is (re, im) of two adjacent wavelengths of ONE row: the product with conj(H) needs no cross-lane move and
// the sum over wavelengths runs inside the lane.  A lane carries sum_l tpl[t][l] z[l] for its rows (4 templates x re/im x
// (row r, row N-r) per row tile) across all tiles of one k_beta: 16 registers per row tile, so a wave takes 64 of the 128
// output rows.  Workgroups come in pairs that transform the same columns, one per row half: a workgroup keeps only its 64
// rows of the matrix image and has 96 KB of LDS for the OTF stream (below).  The block exponent is per tile (wave-uniform)
// here, because an accumulator register no longer belongs to the lane that folded its column.
// Partial sums leave the kernel once per (workgroup pair, k_beta) into `mpart[kb][slot][row][t][c]`; a second, tiny kernel
// adds the slots in a fixed order (deterministic, no atomics).
// Measured (config 3, 251 x 251 x 4096 planes): 0.59 ms against 0.42 + 0.42 ms for the pass and the separate reduction
// (0.49 ms with the OTF replaced by a constant; 0.90 ms with the OTF read straight from memory, lane = row).
__device__ __forceinline__ float wave_max(float m) {
#define H2_DPPMAX(ctrl_) m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), ctrl_, 0xF, 0xF, false)))
    H2_DPPMAX(0xB1);     // quad_perm [1,0,3,2]
    H2_DPPMAX(0x4E);     // quad_perm [2,3,0,1]
    H2_DPPMAX(0x141);    // row_half_mirror
    H2_DPPMAX(0x140);    // row_mirror: every lane of a row of 16 holds the row's maximum
#undef H2_DPPMAX
    unsigned u = __float_as_uint(m);
    auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    m = fmaxf(m, fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1])));
    u = __float_as_uint(m);
    auto s32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(m, fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1])));
}

__global__ __launch_bounds__(NTHREADS) void dft_h2_adjmix_kernel(DftH2Args g, DftH2AdjMix am, const uint4 *__restrict__ img, int kA, int NS) {
    static_assert(NWAVES == 8, "dft_h2_adjmix_kernel: 4 column streams x 2 row halves");
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    constexpr int KIND = 0;
    constexpr bool MIX = false;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups come in pairs that transform the same columns, one per half of the output rows (blockIdx b and b + 8: the
    // same XCD under round-robin placement, so the second read of a tile comes from that XCD's L2 -- speed only).  A
    // workgroup therefore keeps only its 64 rows of the matrix image (64 KB) and has 96 KB of LDS left for the OTF stream.
    const int cs = wave;                                       // column stream
    const int rh = ((int)blockIdx.x >> 3) & 1;                 // row half
    const int vb = ((int)blockIdx.x >> 4) * 8 + ((int)blockIdx.x & 7), VP = (int)gridDim.x / 2;   // pair index, number of pairs
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int TNW = 16, SRC_T = 32;
    const int lcol = l31 >> 1;
    const int tilesX = g.N / TNW;
    const unsigned ldb4 = (unsigned)(g.ldb * 4), ldh4 = (unsigned)(am.ldh * 4);
    const unsigned c4 = (unsigned)l31 * 4u;
    const int nk = g.KP / BK;
    const int kin = g.Kn / 2 + 1;
    const int kt0 = am.kt0;            // first k-step whose rows (k or Kn - k) can be non-zero
    // unused by this kernel, referenced by the shared macros
    const float4 *mtab = nullptr;
    const float4 tw = make_float4(0.f, 0.f, 0.f, 0.f);
    const float sgv = 0.f;
    (void)mtab; (void)tw; (void)sgv; (void)lcol;
    {   // this half's 64 rows of every (matrix, piece, k-step) block of the image
        uint4 *l4 = reinterpret_cast<uint4 *>(lds);
        for (int i = tid; i < IMG_HALFS / 16; i += NTHREADS) {
            const int blk = i >> 7, in = i & 127;              // 128 uint4 (64 rows x 16 halfs) per block of the half image
            l4[i] = img[blk * 256 + rh * 128 + in];
        }
    }
    // super-tiles of 8 tiles (128 complex columns); wave cs takes tile 8 s + cs
    const int s0 = (int)((long)NS * vb / VP), s1 = (int)((long)NS * (vb + 1) / VP);
    const int ntw = s1 - s0;
    __syncthreads();
    if (ntw <= 0) return;

    __amdgpu_buffer_rsrc_t R0;
    int hv = h;
    float xr[8], qr[8];
    float x0[8], x1[8];
    f16x8 c0h, c0l, c1h, c1l;
    f32x16 acc1[2], acc2[2];
    f32x2 M[2][2][4];                  // [row tile][row r / row N - r][template] (re, im)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) M[i][k][t] = f32x2{0.f, 0.f};

    // fragments of the matrix rows of this wave's half: B operand now (same registers as the A fragment of dft_h2_kernel)
#define H2_MFMA_SW(m_, kt_, acc_, bh_, bl_)                                                                     \
    {                                                                                                           \
        const unsigned short *ra = lds + ((m_) * 2 * KT + (kt_)) * (PIECE / 2) + l31 * RS + 8 * (h ^ ((l31 >> 3) & 1)); \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {                                                      \
            const unsigned short *p = ra + mt * 32 * RS;                                                        \
            const f16x8 ah = *reinterpret_cast<const f16x8 *>(p);                                               \
            const f16x8 al = *reinterpret_cast<const f16x8 *>(p + KT * (PIECE / 2));                            \
            f32x16 c_ = acc_[mt];                                                                               \
            c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl_, ah, c_, 0, 0, 0);                                  \
            c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh_, al, c_, 0, 0, 0);                                  \
            c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh_, ah, c_, 0, 0, 0);                                  \
            acc_[mt] = c_;                                                                                      \
        }                                                                                                       \
    }
    // exponent of the tile's largest folded magnitude (wave-uniform)
#define H2_MAXEXP_W(p_)                                                                                         \
    {                                                                                                           \
        float m_ = 0.f;                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) m_ = fmaxf(m_, fmaxf(fabsf(x0[j]), fabsf(x1[j])));        \
        m_ = wave_max(m_);                                                                                      \
        p_ = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_frexp_expf(m_));                                   \
    }
    // partial sums of one k_beta leave the wave: the two lane halves (different wavelengths of the same rows) are added,
    // lanes 0-31 store [row][t][c]
#define H2_FLUSH(kb_)                                                                                           \
    {                                                                                                           \
        const long kbs_ = am.kbstart ? (long)am.kbstart[kb_] : (long)(kb_) * (tilesX / 8);   /* first super-tile of kb in the launch's numbering */ \
        int bf = vb;                                                                                            \
        while (bf > 0 && (long)NS * bf / VP > kbs_) --bf;                         /* first pair that holds tiles of kb */ \
        const int slot = (vb - bf) * 8 + cs;                                                                    \
        float *mp = am.mpart + ((long)(kb_) * am.nslot + slot) * (256 * 8);                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
            _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                     \
                float v[8];                                                                                     \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                 \
                    _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                             \
                        const unsigned u = __float_as_uint(M[i][k][t][c]);                                      \
                        const auto sw_ = __builtin_amdgcn_permlane32_swap(u, u, false, false);                  \
                        v[2 * t + c] = __uint_as_float(sw_[0]) + __uint_as_float(sw_[1]);   /* lower half + upper half, in every lane */ \
                    }                                                                                           \
                    M[i][k][t] = f32x2{0.f, 0.f};                                                               \
                }                                                                                               \
                const int r = (2 * rh + i) * 32 + l31;                                                          \
                const int row = k ? g.Rn - r : r;                                                               \
                const bool ok_ = k ? (r >= 1 && r < g.rvalid && 2 * r != g.Rn) : (r < g.rvalid);                \
                if (h == 0 && slot < am.nslot && ok_) {                                                         \
                    *reinterpret_cast<float4 *>(mp + row * 8) = make_float4(v[0], v[1], v[2], v[3]);            \
                    *reinterpret_cast<float4 *>(mp + row * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);        \
                }                                                                                               \
            }                                                                                                   \
    }

    const int *avl = am.vlist;
#define H2_ART(vt_) (avl ? 8 * avl[(vt_) >> 3] + ((vt_) & 7) : (vt_))
    int vt = 8 * s0 + cs;
    const int vtend = vt + 8 * ntw;
    int tile = H2_ART(vt);
    int e = 0, en = 0;

    H2_LSETUP(tile);
    H2_LOAD(kt0);
    H2_FOLD(kt0);
    {
        int p;
        H2_MAXEXP_W(p);
        e = E_TARGET - p;
    }
    split8h(x0, e, c0h, c0l);
    split8h(x1, e, c1h, c1l);
    H2_LOAD(kt0 + 1);
    // ---- the OTF stream -----------------------------------------------------------------------------------------
    // The OTF rows of a tile reach the lanes through staging buffers in LDS: three 4 KB slots per wave.  A lane owns ROWS
    // here, so read straight from memory a wave instruction would touch 32 rows x 32 B (measured: 2.3 TB/s).  Instead one
    // block = 32 rows x 128 B (the tile's 16 wavelengths, re / im) is moved by four LDS-DMA instructions that each read 8
    // whole 128-byte rows, into an XOR-swizzled [row][8 chunks of 16 B] image (chunk c of row k at position c ^ (k >> 1 & 7):
    // conflict-free ds_read_b128 for the lane = row pattern), and read back as four 16-byte chunks per lane.  A tile has four
    // blocks per wave (row tiles 0 / 1, rows r / N - r).  Blocks 0, 1, 2 are requested in k-steps 1, 2, 3 of the tile's own
    // k loop, in front of that step's data loads: when the loop ends they have landed (vmcnt is one in-order counter and
    // every fold waits for younger loads).  Block 3 follows into slot 0 as soon as block 0 has been read.
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const unsigned hb_lds = (unsigned)(size_t)lds + (unsigned)IMG_HALFS + (unsigned)wave * 12288u;     // behind the 64 KB half image
    // DMA lane map: instruction i moves LDS rows 8 i .. 8 i + 7; lane -> (row 8 i + (lane >> 3), position lane & 7), which
    // holds chunk (lane & 7) ^ ((row >> 1) & 7) of that row; LDS row k = matrix row R0 + k (ascending) or R0 + 31 - k.
    // (row >> 1) & 7 = (4 i + (lane >> 4)) & 7: two lane offsets per direction (i even / odd), the rest is in the scalar base
    const unsigned hc0 = (unsigned)((lane & 7) ^ ((lane >> 4) & 3)), hc1 = hc0 ^ 4u;
    const unsigned hoa0 = (unsigned)(lane >> 3) * ldh4 + hc0 * 16u, hoa1 = (unsigned)(lane >> 3) * ldh4 + hc1 * 16u;
    const unsigned hod0 = (unsigned)(7 - (lane >> 3)) * ldh4 + hc0 * 16u, hod1 = (unsigned)(7 - (lane >> 3)) * ldh4 + hc1 * 16u;
    const unsigned hrd = hb_lds + (unsigned)l31 * 128u, hsw = (unsigned)((l31 >> 1) & 7);
    // request block blk_ (2 * row tile + (0: rows r, 1: rows N - r)) of the tile (kb_, first column n0_) into slot slot_
#define H2_HDMA(blk_, slot_, kb_, n0_)                                                                          \
    {                                                                                                           \
        const int mt = 2 * rh + ((blk_) >> 1);                                                                  \
        const bool dsc = ((blk_) & 1) != 0;                                                                     \
        const char *hp = reinterpret_cast<const char *>(am.hsrc + (long)(kb_) * am.sH + (n0_) * 2);             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
            /* ascending: rows mt*32 + 8 i + (lane >> 3); descending: rows Rn - mt*32 - 8 i - (lane >> 3) = base + 7 - (lane >> 3) */ \
            const long r0 = dsc ? (long)(g.Rn - mt * 32 - 8 * i - 7) : (long)(mt * 32 + 8 * i);                 \
            unsigned off = dsc ? ((i & 1) ? hod1 : hod0) : ((i & 1) ? hoa1 : hoa0);                             \
            /* row Rn itself (the "mirror" of row 0, never used) may lie behind the array: its lanes read row Rn - 1 */ \
            if (dsc && mt == 0 && i == 0 && (lane >> 3) == 0) off -= ldh4;                                      \
            const char *bp = hp + r0 * (long)ldh4;                                                              \
            const unsigned la = hb_lds + (unsigned)(slot_) * 4096u + (unsigned)i * 1024u;                       \
            if (!(H2_EXP & 16))                                                                                 \
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(bp), "s"(la) : "memory"); \
        }                                                                                                       \
    }
    // consume the block in slot slot_: mir_ 0 rows r / 1 rows N - r of row tile i_
#define H2_HBLOCK(i_, mir_, slot_)                                                                              \
    {                                                                                                           \
        f32x4_ hq4[4];                                                                                          \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
            const unsigned ad = hrd + (unsigned)(slot_) * 4096u + (((unsigned)(2 * q + h)) ^ hsw) * 16u;        \
            if (H2_EXP & 16) hq4[q] = f32x4_{1.f, 0.5f, 0.25f, 2.f};                                            \
            else asm volatile("ds_read_b128 %0, %1" : "=v"(hq4[q]) : "v"(ad) : "memory");                       \
        }                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hq4[0]), "+v"(hq4[1]), "+v"(hq4[2]), "+v"(hq4[3])::"memory"); \
        const float c0 = f * ((mir_) ? g.e[2] : g.e[0]), c1 = f * ((mir_) ? g.e[3] : g.e[1]);                   \
        const float d0 = f * ((mir_) ? g.e_alt[2] : g.e_alt[0]), d1 = f * ((mir_) ? g.e_alt[3] : g.e_alt[1]);   \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
            float ta[4], tb[4];      /* template weights of this lane's two wavelengths of the quad: columns 4 q + 2 h, + 1 */ \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                     \
                const float lo0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tv), (4 * q) * 4 + t));     \
                const float lo1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tv), (4 * q + 1) * 4 + t)); \
                const float hi0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tv), (4 * q + 2) * 4 + t)); \
                const float hi1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tv), (4 * q + 3) * 4 + t)); \
                ta[t] = h ? hi0 : lo0;                                                                          \
                tb[t] = h ? hi1 : lo1;                                                                          \
            }                                                                                                   \
            const float p0 = acc1[i_][4 * q], p1 = acc1[i_][4 * q + 1], p2 = acc1[i_][4 * q + 2], p3 = acc1[i_][4 * q + 3];   \
            const float s0v = acc2[i_][4 * q], s1v = acc2[i_][4 * q + 1], s2v = acc2[i_][4 * q + 2], s3v = acc2[i_][4 * q + 3]; \
            const float yr0 = c0 * p0 + c1 * s0v, yi0 = d0 * p1 + d1 * s1v, yr1 = c0 * p2 + c1 * s2v, yi1 = d0 * p3 + d1 * s3v; \
            const f32x4_ hh = hq4[q];                                                                           \
            const f32x2 z0 = {hh[0] * yr0 + hh[1] * yi0, hh[0] * yi0 - hh[1] * yr0};                            \
            const f32x2 z1 = {hh[2] * yr1 + hh[3] * yi1, hh[2] * yi1 - hh[3] * yr1};                            \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) M[i_][mir_][t] += ta[t] * z0 + tb[t] * z1;            \
        }                                                                                                       \
    }
    const bool piped = nk - kt0 >= 5;  // k-steps kt0 + 1 .. kt0 + 3 of the loop below exist
    while (true) {
        asm volatile("" : "+v"(hv));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] = acc2[i][r] = 0.f;
        const bool more = vt + 8 < vtend;
        const int next = more ? H2_ART(vt + 8) : tile;
        const int kb = tile / tilesX;
        const long n0 = (long)(tile % tilesX) * TNW;
        for (int kt = kt0; kt + 1 < nk; ++kt) {
            H2_MFMA_SW(0, kt, acc1, c0h, c0l);
            H2_FOLD(kt + 1);
            int p;
            H2_MAXEXP_W(p);
            en = (p + e > E_LIMIT) ? E_TARGET - p : e;
            const int d = en - e;
            if (piped) {               // wave-uniform; the slots were emptied by the previous tile's epilogue
                if (kt == kt0 + 1) { H2_HDMA(0, 0, kb, n0); }
                else if (kt == kt0 + 2) { H2_HDMA(1, 1, kb, n0); }
                else if (kt == kt0 + 3) { H2_HDMA(2, 2, kb, n0); }
            }
            if (kt + 2 < nk) {
                H2_LOAD(kt + 2);
            } else if (more) {
                H2_LSETUP(next);
                H2_LOAD(kt0);
            }
            split8h(x0, en, c0h, c0l);
            H2_MFMA_SW(1, kt, acc2, c1h, c1l);
            split8h(x1, en, c1h, c1l);
            if (d != 0) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc1[mt][r] = __builtin_amdgcn_ldexpf(acc1[mt][r], d);
                        acc2[mt][r] = __builtin_amdgcn_ldexpf(acc2[mt][r], d);
                    }
            }
            e = en;
        }
        H2_MFMA_SW(0, nk - 1, acc1, c0h, c0l);
        if (more) {
            H2_FOLD(kt0);
            int p;
            H2_MAXEXP_W(p);
            en = E_TARGET - p;
            H2_LOAD(kt0 + 1);
            split8h(x0, en, c0h, c0l);
        }
        H2_MFMA_SW(1, nk - 1, acc2, c1h, c1l);
        if (more) split8h(x1, en, c1h, c1l);
        {
            // epilogue: z = conj(H) Y for this lane's rows, summed over the tile's wavelengths with the template weights.
            // lane L of `tv` holds tpl[t = L & 3][column n0 + (L >> 2)]; read back lane by lane (scalar)
            const float tv = ((lane & 3) < am.T) ? am.tpl[(long)(lane & 3) * am.LPt + n0 + (lane >> 2)] : 0.f;
            const float f = __builtin_amdgcn_ldexpf(1.f, -e - kA);
            if (!piped) {
                H2_HDMA(0, 0, kb, n0); H2_HDMA(1, 1, kb, n0); H2_HDMA(2, 2, kb, n0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            H2_HBLOCK(0, 0, 0);
            H2_HDMA(3, 0, kb, n0);
            H2_HBLOCK(0, 1, 1);
            H2_HBLOCK(1, 0, 2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            H2_HBLOCK(1, 1, 0);
        }
        if (!more || next / tilesX != kb) H2_FLUSH(kb);
        if (!more) break;
        e = en;
        tile = next;
        vt += 8;
    }
#undef H2_ART
#undef H2_HDMA
#undef H2_HBLOCK
#undef H2_MFMA_SW
#undef H2_MAXEXP_W
#undef H2_FLUSH
}

// madj[t][c][ka * KBP + kb] = sum over the slots that hold partial sums of kb (fixed order); the padding of the spectrum
// planes (kb >= hb, ka >= Na) is written as zero, as the separate reduction leaves it
__global__ __launch_bounds__(256) void dft_h2_adjmix_reduce_kernel(const float *__restrict__ mpart, float *__restrict__ madj, int nslot,
                                                                   int T, int Na, int KAP, int hb, long KBP, long PL, int NS, int G, int spk,
                                                                   float out_self, float out_pair, int Nb, const float *__restrict__ prior_src,
                                                                   float prior_mu, const int *__restrict__ kbstart) {
    const int kb = blockIdx.x;
    const float osc = (kb == 0 || 2 * kb == Nb) ? out_self : out_pair;
    const float pcb = prior_src ? 2.f - 2.f * cospif(2.f * (float)kb / (float)Nb) : 0.f;
    if (kb >= hb) {
        for (int i = blockIdx.y * 256 + threadIdx.x; i < KAP * 8; i += 256 * gridDim.y) {
            const int row = i >> 3, tc = i & 7, t = tc >> 1, c = tc & 1;
            if (t < T) madj[((long)t * 2 + c) * PL + (long)row * KBP + kb] = 0.f;
        }
        return;
    }
    int bf = 0, bl = G - 1;      // pairs bf .. bl of the pass hold tiles of kb: range of pair b = [NS b / G, NS (b + 1) / G)
    const long lo = kbstart ? kbstart[kb] : (long)kb * spk, hi = kbstart ? kbstart[kb + 1] : lo + spk;
    if (hi > lo) {
        int b = (int)(((long)lo * G) / NS);
        while (b > 0 && (long)NS * b / G > lo) --b;
        while ((long)NS * (b + 1) / G <= lo) ++b;
        bf = b;
        while (b + 1 < G && (long)NS * (b + 1) / G < hi) ++b;
        bl = b;
    }
    const int ns = hi <= lo ? 0 : ((bl - bf + 1) * 8 < nslot ? (bl - bf + 1) * 8 : nslot);      // a k_beta without tiles: zero
    for (int i = blockIdx.y * 256 + threadIdx.x; i < KAP * 8; i += 256 * gridDim.y) {      // gridDim.y blocks share a k_beta: the sums are short chains of loads
        const int row = i >> 3, tc = i & 7, t = tc >> 1, c = tc & 1;
        if (t >= T) continue;
        float s = 0.f;
        const long o = ((long)t * 2 + c) * PL + (long)row * KBP + kb;
        if (row < Na) {
            for (int sl = 0; sl < ns; ++sl) s += mpart[((long)kb * nslot + sl) * (256 * 8) + row * 8 + tc];
            s *= osc;
            if (prior_src) s += prior_mu * (pcb + 2.f - 2.f * cospif(2.f * (float)row / (float)Na)) * prior_src[o];
        }
        madj[o] = s;
    }
}

#undef H2_MIXTAB
#undef H2_FSETUP
#undef H2_LSETUP
#undef H2_RSRC
#undef H2_BLOAD
#undef H2_BLOAD2
#undef H2_LOAD
#undef H2_FOLD
#undef H2_FOLD_BODY
#undef H2_MAXEXP
#undef H2_MFMA
#undef H2_RT

}  // namespace

int dft_h2_build_image(const float *A0, const float *A1, int MP, int KP, int lda, unsigned short *img) {
    std::memset(img, 0, DFT_H2_IMAGE_HALFS * sizeof(unsigned short));
    float amax = 0.f;
    const float *A[2] = {A0, A1};
    for (int m = 0; m < 2; ++m)
        for (int r = 0; r < MP; ++r)
            for (int k = 0; k < KP; ++k) amax = std::fmax(amax, std::fabs(A[m][(size_t)r * lda + k]));
    int pa = 0;
    if (amax > 0.f) std::frexp(amax, &pa);                       // amax < 2^pa
    const int kA = 14 - pa;                                       // largest piece below 2^14
    for (int m = 0; m < 2; ++m)
        for (int r = 0; r < MP; ++r)
            for (int k = 0; k < KP; ++k) {
                const float xs = std::ldexp(A[m][(size_t)r * lda + k], kA);
                const _Float16 hh = (_Float16)xs;
                const _Float16 ll = (_Float16)(xs - (float)hh);
                const int kt = k / BK, c = (k % BK) / 8, j = k % 8;
                const size_t pos = (size_t)kt * PIECE + (size_t)r * RS + 8 * (c ^ ((r >> 3) & 1)) + j;
                unsigned short uh, ul;
                std::memcpy(&uh, &hh, 2);
                std::memcpy(&ul, &ll, 2);
                img[((size_t)(m * 2 + 0) * KT) * PIECE + pos] = uh;
                img[((size_t)(m * 2 + 1) * KT) * PIECE + pos] = ul;
            }
    return kA;
}

int launch_dft_h2(hipStream_t stream, const DftH2Args &g, const unsigned short *img, int kA) {
    if (g.kind < 0 || g.kind > 2 || g.KP % BK || g.KP < 2 * BK || g.KP > KT * BK || g.N % 128 || g.batch < 1 || !img || !g.src || !g.dst)
        return (int)hipErrorInvalidValue;
    if (g.kind != 2 && g.KP > g.Kn) return (int)hipErrorInvalidValue;
    if (g.rvalid < 1 || g.rvalid > 128) return (int)hipErrorInvalidValue;
    // 32-bit scalar row offsets (source rows 0 .. Kn, destination rows 0 .. max(Rn, 128) + 4) and 32-bit lane offsets
    const double rows_src = (double)(g.Kn > g.KP ? g.Kn : g.KP) + 9.0, rows_dst = (double)(g.Rn > 128 ? g.Rn : 128) + 9.0;
    if (rows_src * (double)g.ldb * 4.0 + 1024.0 >= 4294967296.0 || rows_dst * (double)g.ldc * 4.0 + 1024.0 >= 4294967296.0) return (int)hipErrorInvalidValue;
    const int mix_rows = g.Kn > g.KP ? g.Kn : g.KP;
    if (g.mhat && (g.kind != 0 || g.LP % 128 || g.T < 1 || g.T > 4 || mix_rows > MIX_ROWS || !g.tpl)) return (int)hipErrorInvalidValue;
    static int cus_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (!cus_of[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return (int)hipErrorInvalidDevice;
        cus_of[dev] = cus;
    }
    const int tnw = g.kind == 0 ? 16 : 32;
    long NT = (long)(g.N / tnw) * g.batch;
    long NS = (NT + NWAVES - 1) / NWAVES;
    if (NS >= 2147483647L / NWAVES) return (int)hipErrorInvalidValue;
    if ((g.ktab || g.rtab) && (g.tabLP < 128 || g.tabLP % 128)) return (int)hipErrorInvalidValue;
    if (g.vlist) {             // whole super-tiles only
        if (NT % NWAVES || g.nvalid < 1 || g.nvalid > NS) return (int)hipErrorInvalidValue;
        NS = g.nvalid;
        NT = NS * NWAVES;
    }
    dim3 grid((unsigned)(NS < cus_of[dev] ? NS : cus_of[dev]));
    const uint4 *im = reinterpret_cast<const uint4 *>(img);
    static unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    if (g.mhat) {
        if (int e = ensure_dynamic_lds(dft_h2_kernel<0, true>, LDS_IMG + LDS_MIX, d3)) return e;
        hipLaunchKernelGGL((dft_h2_kernel<0, true>), grid, dim3(NTHREADS), LDS_IMG + LDS_MIX, stream, g, im, kA, (int)NS, NT);
    } else if (g.kind == 0) {
        if (int e = ensure_dynamic_lds(dft_h2_kernel<0, false>, LDS_IMG, d0)) return e;
        hipLaunchKernelGGL((dft_h2_kernel<0, false>), grid, dim3(NTHREADS), LDS_IMG, stream, g, im, kA, (int)NS, NT);
    } else if (g.kind == 1) {
        if (int e = ensure_dynamic_lds(dft_h2_kernel<1, false>, LDS_IMG, d1)) return e;
        hipLaunchKernelGGL((dft_h2_kernel<1, false>), grid, dim3(NTHREADS), LDS_IMG, stream, g, im, kA, (int)NS, NT);
    } else {
        if (int e = ensure_dynamic_lds(dft_h2_kernel<2, false>, LDS_IMG, d2)) return e;
        hipLaunchKernelGGL((dft_h2_kernel<2, false>), grid, dim3(NTHREADS), LDS_IMG, stream, g, im, kA, (int)NS, NT);
    }
    return (int)hipGetLastError();
}

// workgroups of the fused adjoint pass and partial-sum slots per k_beta for `tiles_per_kb` = LP / 16 tiles per k_beta
static void adjmix_geometry(int dev_cus, long tilesX, int hb, long nvalid, long &NS, int &G, int &nslot) {
    NS = nvalid > 0 ? nvalid : tilesX / 8 * hb;                // super-tiles of 8 tiles (all, or the listed ones)
    long pairs = dev_cus / 16 * 8;                             // workgroup pairs (b, b + 8): the grid is a multiple of 16
    if (pairs < 8) pairs = 8;
    while (pairs > 8 && pairs > NS) pairs -= 8;
    G = (int)(2 * pairs);
    const long spk = tilesX / 8;
    const long per = NS / pairs > 0 ? NS / pairs : 1;          // smallest range of a pair (super-tiles)
    nslot = (int)(8 * ((spk + per - 1) / per + 2));
}

size_t dft_h2_adjmix_part_floats(long LP, int hb, long nvalid) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return 0;
    long NS; int G, nslot;
    adjmix_geometry(cus, LP / 16, hb, nvalid, NS, G, nslot);
    return (size_t)hb * nslot * 256 * 8;
}

int launch_dft_h2_adjmix(hipStream_t stream, const DftH2Args &g, const DftH2AdjMix &am0, float *madj, long PL, long KBP, const unsigned short *img, int kA) {
    if (am0.kt0 < 0 || (am0.kt0 > 0 && g.KP / BK - am0.kt0 < 2)) return (int)hipErrorInvalidValue;
    if (g.kind != 0 || g.mhat || g.KP % BK || g.KP < 2 * BK || g.KP > KT * BK || g.KP > g.Kn || g.N % 128 || g.batch < 1 || !img || !g.src ||
        !am0.hsrc || !am0.tpl || !am0.mpart || !madj || am0.T < 1 || am0.T > 4 || g.Rn < 32 * 3 + 31 || g.Rn > 255 || g.rvalid < 1 || g.rvalid > 128)
        return (int)hipErrorInvalidValue;
    const double rows_src = (double)g.Kn + 9.0;
    if (rows_src * (double)g.ldb * 4.0 + 1024.0 >= 4294967296.0 || 260.0 * (double)am0.ldh * 4.0 + 1024.0 >= 4294967296.0) return (int)hipErrorInvalidValue;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return (int)hipErrorInvalidDevice;
    long NS; int G, nslot;
    if ((am0.vlist != nullptr) != (am0.kbstart != nullptr) || (am0.vlist && (am0.nvalid < 1 || am0.nvalid > (long)(g.N / 128) * g.batch)))
        return (int)hipErrorInvalidValue;
    adjmix_geometry(cus, g.N / 16, g.batch, am0.vlist ? am0.nvalid : 0, NS, G, nslot);
    if (NS >= 2147483647L / 8 || g.N % 128) return (int)hipErrorInvalidValue;
    DftH2AdjMix am = am0;
    am.nslot = nslot;
    static unsigned long long d4 = 0;
    const size_t ldsb = LDS_IMG / 2 + (size_t)NWAVES * 12288;  // half image + three 4 KB OTF staging slots per wave = 160 KB
    if (int e = ensure_dynamic_lds(dft_h2_adjmix_kernel, ldsb, d4)) return e;
    hipLaunchKernelGGL(dft_h2_adjmix_kernel, dim3((unsigned)G), dim3(NTHREADS), ldsb, stream, g, am, reinterpret_cast<const uint4 *>(img), kA, (int)NS);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(dft_h2_adjmix_reduce_kernel, dim3((unsigned)KBP, 8), dim3(256), 0, stream, am.mpart, madj, nslot, am.T, g.Rn, (int)(PL / KBP),
                       g.batch, KBP, PL, (int)NS, G / 2, (int)(g.N / 16 / 8), am.out_self, am.out_pair, am.Nb, am.prior_src, am.prior_mu, am.kbstart);
    return (int)hipGetLastError();
}

bool dft_h2_supported(int Na, int Nb, long NAP, long KBP, long LP) {
    const double pitch = 2.0 * (double)(NAP > KBP ? NAP : KBP) * (double)LP * 4.0;   // largest row pitch of any pass (bytes, interleaved)
    const int n = Na > Nb ? Na : Nb, m = Na < Nb ? Na : Nb;
    return n / 2 + 1 <= 128 && m / 2 + 1 > BK && n <= MIX_ROWS && LP % 128 == 0 && ((double)n + 9.0) * pitch + 1024.0 < 4294967296.0;
}
