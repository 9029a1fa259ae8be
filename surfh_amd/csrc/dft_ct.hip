// DFT pass of length N = R * M: R sub-transforms of length M on the matrix cores (the k loop of dft_h2.hip, one wave per
// residue class), combined through LDS by twiddles and an R-point butterfly (see dft_ct.h).
#include "dft_ct.h"
#include "lds_attr.h"
#include <cmath>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef CT_EXP
#define CT_EXP 0        // tools/exp: 2 no stores (and no combine: dead code), 4 no MFMAs, 16 no group synchronisation (wrong results), 32 no loads,
                        // 64 workgroup barrier instead of the group counters, 256 every store into one cache-resident region, 512 the combine without
                        // its stores, 1024 stores without the twiddle / butterfly arithmetic, 2048 a third / half of the row pairs not stored
#endif

namespace {

constexpr int BK = 16;
constexpr int MAXMT = 3;                               // row tiles of 32 of the folded sub-transform (M / 2 + 1 <= 96)
constexpr int E_TARGET = 10, E_LIMIT = 15;             // per-column block exponent of the fp16 pieces (dft_h2.hip)
constexpr int XFLOATS = 1024;                          // one wave's exchange slot: 32 rows x 32 lanes
constexpr size_t LDS_LIMIT = 160 * 1024;

template <int R>
struct Cfg {
    static constexpr int NG = (R == 2) ? 4 : 2;        // groups of R waves per workgroup
    static constexpr int NW = NG * R;
    static constexpr int NTH = 64 * NW;
    static constexpr int UPS = 128 / (16 * NG);        // units (NG tiles of 16 columns) per super-tile of 128 columns
};

// cos / sin of 2 pi q / R, q = 0 .. R-1
template <int R> __device__ __forceinline__ constexpr float wr_cos(int q) {
    return R == 2 ? (q == 0 ? 1.f : -1.f) : R == 3 ? (q == 0 ? 1.f : -0.5f) : (q == 0 ? 1.f : q == 2 ? -1.f : 0.f);
}
template <int R> __device__ __forceinline__ constexpr float wr_sin(int q) {
    return R == 2 ? 0.f : R == 3 ? (q == 0 ? 0.f : q == 1 ? 0.86602540378443865f : -0.86602540378443865f) : (q == 1 ? 1.f : q == 3 ? -1.f : 0.f);
}

#define MFMA3(acc_, ah_, al_, bh_, bl_)                                             \
    {                                                                               \
        f32x16 c_ = acc_;                                                           \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_, bh_, c_, 0, 0, 0);         \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bl_, c_, 0, 0, 0);         \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bh_, c_, 0, 0, 0);         \
        acc_ = c_;                                                                  \
    }

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

// workgroup barrier that waits for LDS traffic only: global loads of the next tile and stores of the previous phase stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The R waves of a group meet at a counter in LDS instead of the workgroup barrier, so the groups of a workgroup run
// decoupled: one group's loads and MFMAs overlap the other's stores (0.67 -> 0.62 ms per 2 GB pass at N = 501).  Every wave
// adds 1 per phase after its LDS writes (the LDS operations of a wave are performed in order) and waits until the count
// reaches R x phases.
__device__ __forceinline__ void group_arrive_wait(unsigned *cnt, unsigned target, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (true) {
        const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(v - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

template <int R, int LD, int EP>
__global__ __launch_bounds__(Cfg<R>::NTH) void dft_ct_kernel(DftCtArgs g, const uint4 *__restrict__ img, const float *__restrict__ twg,
                                                            const int *__restrict__ vl, const int *__restrict__ ktab, const int *__restrict__ rtab,
                                                            int MT, int KT, int kA, int NU) {
    constexpr int NG = Cfg<R>::NG, NW = Cfg<R>::NW, NTH = Cfg<R>::NTH, UPS = Cfg<R>::UPS;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / R, n1 = wave % R;                   // group (column tile of the unit), residue class
    const int l31 = lane & 31, h = lane >> 5;
    const int var = l31 & 1;                                   // the component this lane owns
    const int lcol = l31 >> 1;
    const float sig = var ? 1.f : -1.f;
    const int M = g.M, N = R * M, Mh = M / 2 + 1, Nh = N / 2;
    const int PIECE = 32 * MT * 16;                            // halfs of one (matrix, piece, k-step) block
    const int IMGH = 4 * KT * PIECE;
    constexpr int SRC_T = (LD == DFT_CT_HPACK) ? 64 : 32, DST_T = (EP == DFT_CT_HSEP) ? 64 : 32;   // floats of a tile's row segment
    const int tilesX = g.ncols / 16;
    const long stepR = (long)R * g.ldb;                        // floats between consecutive rows of a sub-sequence
    const unsigned step4 = (unsigned)(stepR * 4), ldb4 = (unsigned)(g.ldb * 4);
    const unsigned c4 = (unsigned)l31 * (LD == DFT_CT_HPACK ? 8u : 4u);
    float *const xbuf = reinterpret_cast<float *>(lds + IMGH);
    float *const twl = xbuf + 2 * NW * XFLOATS;                // [M][R - 1] (cos, sin)
    unsigned *const gsync = reinterpret_cast<unsigned *>(twl + (((M * (R - 1) * 2) + 3) & ~3));      // one arrival counter per group
    float4 *const mixbuf = reinterpret_cast<float4 *>(gsync + 4);
    float *const awk = reinterpret_cast<float *>(mixbuf);      // PRODADD: weight of row k of the full transform, [N]

    {   // constants: global -> LDS, once
        uint4 *l4 = reinterpret_cast<uint4 *>(lds);
        for (int i = tid; i < IMGH / 8; i += NTH) l4[i] = img[i];
        for (int i = tid; i < M * (R - 1) * 2; i += NTH) twl[i] = twg[i];
        if (LD == DFT_CT_PRODADD)
            for (int i = tid; i < N; i += NTH) awk[i] = g.add_w * (2.f - 2.f * cospif(2.f * (float)i / (float)N));
        if (tid < 4) gsync[tid] = 0u;
    }
    // this workgroup's contiguous range of units (NG adjacent tiles of 16 columns, one per group of waves)
    const int u0 = (int)((long)NU * blockIdx.x / gridDim.x), u1 = (int)((long)NU * (blockIdx.x + 1) / gridDim.x);
    const int ntw = u1 - u0;
#define CT_TILE(vu_) ((vl ? vl[(vu_) / UPS] * UPS + (vu_) % UPS : (vu_)) * NG + grp)

    // fused spectral mix: column (k, kb) of mhat for all N rows k as [k][template pair](re, im, re, im), two tables in turn
    const float4 *mtab = mixbuf;
    float4 tw4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int mix_kb = -1, mix_sel = 0;
#define CT_MIXTAB(kb_)                                                                                          \
    {                                                                                                           \
        float4 *mt_ = mixbuf + mix_sel * (N * 2);                                                               \
        mix_sel ^= 1;                                                                                           \
        const float msc_ = ((kb_) == 0 || 2 * (kb_) == g.mix_Nb) ? g.mhat_self : g.mhat_pair;                   \
        for (int e_ = tid; e_ < N * 2; e_ += NTH) {                                                             \
            const int k = e_ >> 1, tp = e_ & 1;                                                                 \
            float v[4];                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
                const int t = 2 * tp + (i >> 1), c = i & 1;                                                     \
                v[i] = (t < g.T) ? g.mhat[((long)t * 2 + c) * g.PL + (long)k * g.KBP + (kb_)] : 0.f;            \
            }                                                                                                   \
            mt_[e_] = make_float4(v[0] * msc_, v[1] * msc_, v[2] * msc_, v[3] * msc_);                          \
        }                                                                                                       \
        mtab = mt_;                                                                                             \
        mix_kb = (kb_);                                                                                         \
    }
    // per-tile state of the mix: template weights of this lane's wavelength, table of the tile's kb.  A change of kb is
    // reached by all waves of the workgroup in the same unit; the table written now was last read two changes ago
#define CT_FSETUP(t_)                                                                                           \
    if (LD == DFT_CT_MIX) {                                                                                     \
        const int fn0 = ((t_) % tilesX) * 16;                                                                   \
        const int kb = g.batch > 1 ? (t_) / tilesX : fn0 / g.LP;                                                \
        const int l = (g.batch > 1 ? fn0 : fn0 % g.LP) + lcol;                                                  \
        float t4[4];                                                                                            \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) t4[t] = (t < g.T) ? g.tpl[(long)t * g.LP + l] : 0.f;      \
        tw4 = make_float4(t4[0], t4[1], t4[2], t4[3]);                                                          \
        if (kb != mix_kb) {                                                                                     \
            CT_MIXTAB(kb);                                                                                      \
            lds_barrier();                                                                                      \
        }                                                                                                       \
    }                                                                                                           \
    if (LD == DFT_CT_PRODADD) {                                                                                 \
        const int kb = g.batch > 1 ? (t_) / tilesX : (((t_) % tilesX) * 16) / g.LP;                             \
        awb = g.add_w * (2.f - 2.f * cospif(2.f * (float)kb / (float)g.add_Nb));                                \
    }

    // Addressing of the k loop: buffer loads with one descriptor per k-step and direction (scalar 64-bit base: no 4 GB
    // limit on rows x pitch), the row inside the step as a scalar byte offset, the lane part in one VGPR.
    const float *tp = g.src;           // first row of this wave's sub-sequence in the current tile (HPACK: row 0 of the tile)
    const float *tpp = g.prod;         // PROD: the same row of the second operand
    const long stepP = (long)R * g.ldp;
    const unsigned stepP4 = (unsigned)(stepP * 4), c8 = (unsigned)(l31 >> 1) * 8u;      // both lanes of a column read its (re, im)
    const float psig = g.prod_sign * sig;
    constexpr bool HASPROD = (LD == DFT_CT_PROD || LD == DFT_CT_PRODADD);
    const float *tpd = g.add;          // PRODADD: the same row of the third operand
    float awb = 0.f;                   // ... its weight along the other axis for the tile's k_beta
#define CT_RSRC(ptr_) __builtin_amdgcn_make_buffer_rsrc((void *)(ptr_), 0, 0xFFFFFFFF, 0x00020000)
#define CT_BLOAD(r_, v_, s_) ((CT_EXP & 32) ? __uint_as_float((v_) + (s_)) : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_, (int)(v_), (int)(s_), 0)))
#define CT_BLOAD2(r_, v_, s_) ((CT_EXP & 32) ? f32x2{__uint_as_float((v_) + (s_)), 1.f} : __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_, (int)(v_), (int)(s_), 0)))
#define CT_LSETUP(t_)                                                                                           \
    {                                                                                                           \
        const int tx = (t_) % tilesX;                                                                           \
        const long bz_ = (t_) / tilesX;                                                                         \
        tp = g.src + bz_ * g.sB + (long)tx * SRC_T + (LD == DFT_CT_HPACK ? 0L : (long)n1 * g.ldb);              \
        if (HASPROD) tpp = g.prod + bz_ * g.sP + (long)tx * SRC_T + (long)n1 * g.ldp;                           \
        if (LD == DFT_CT_PRODADD) tpd = g.add + bz_ * g.sB + (long)tx * SRC_T + (long)n1 * g.ldb;               \
    }
    // HPACK: which k-steps hold only interior elements (every j has its mirror, every primary row R j + n1 <= N / 2 for all n1)
#define CT_HFAST(kt_) ((kt_) >= 1 && R * (16 * (kt_) + 16) - 1 <= Nh && 16 * (kt_) + 16 <= Mh - 1)
    // raw loads of this lane's 8 elements j = 16 kt + 8 h + jj of k-step kt_ and of their mirror elements M - j
#define CT_LOAD(kt_)                                                                                            \
    {                                                                                                           \
        if (LD != DFT_CT_HPACK) {                                                                               \
            const __amdgpu_buffer_rsrc_t rp_ = CT_RSRC(tp + (long)(16 * (kt_)) * stepR);                        \
            const __amdgpu_buffer_rsrc_t rq_ = CT_RSRC(tp + (long)(M - 16 * (kt_) - 15) * stepR);               \
            const unsigned vk = (unsigned)(8 * hv) * step4 + c4, vq = (unsigned)(8 * (1 - hv)) * step4 + c4;    \
            _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                  \
                xr[jj] = CT_BLOAD(rp_, vk, (unsigned)jj * step4);                                               \
                /* element 0 has no mirror (row M of the sub-sequence may not exist): both halves read row M - 8 */ \
                qr[jj] = CT_BLOAD(rq_, (jj == 0 && (kt_) == 0) ? c4 : vq, (unsigned)(7 - jj) * step4);          \
            }                                                                                                   \
            if (LD == DFT_CT_PRODADD) {                                                                         \
                const __amdgpu_buffer_rsrc_t dp_ = CT_RSRC(tpd + (long)(16 * (kt_)) * stepR);                   \
                const __amdgpu_buffer_rsrc_t dq_ = CT_RSRC(tpd + (long)(M - 16 * (kt_) - 15) * stepR);          \
                _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                              \
                    xi[jj] = CT_BLOAD(dp_, vk, (unsigned)jj * step4);                                           \
                    qi[jj] = CT_BLOAD(dq_, (jj == 0 && (kt_) == 0) ? c4 : vq, (unsigned)(7 - jj) * step4);      \
                }                                                                                               \
            }                                                                                                   \
            if (HASPROD) {                                                                                      \
                const __amdgpu_buffer_rsrc_t hp_ = CT_RSRC(tpp + (long)(16 * (kt_)) * stepP);                   \
                const __amdgpu_buffer_rsrc_t hq_ = CT_RSRC(tpp + (long)(M - 16 * (kt_) - 15) * stepP);          \
                const unsigned wk = (unsigned)(8 * hv) * stepP4 + c8, wq = (unsigned)(8 * (1 - hv)) * stepP4 + c8; \
                _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                              \
                    hx[jj] = CT_BLOAD2(hp_, wk, (unsigned)jj * stepP4);                                         \
                    hm[jj] = CT_BLOAD2(hq_, (jj == 0 && (kt_) == 0) ? c8 : wq, (unsigned)(7 - jj) * stepP4);    \
                }                                                                                               \
            }                                                                                                   \
        } else if (CT_HFAST(kt_)) {                                                                             \
            /* primary rows R j + n1 read directly, mirror elements Z[N - R j + n1] = conj of row R j - n1 */        \
            const __amdgpu_buffer_rsrc_t rp_ = CT_RSRC(tp + (long)(16 * (kt_) * R + n1) * g.ldb);               \
            const __amdgpu_buffer_rsrc_t rq_ = CT_RSRC(tp + (long)(16 * (kt_) * R - n1) * g.ldb);               \
            const unsigned vk = (unsigned)(8 * hv) * step4 + c4;                                                \
            _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                  \
                const f32x2 a2 = CT_BLOAD2(rp_, vk, (unsigned)jj * step4);                                      \
                const f32x2 b2 = CT_BLOAD2(rq_, vk, (unsigned)jj * step4);                                      \
                xr[jj] = a2[0]; xi[jj] = a2[1];                                                                 \
                qr[jj] = b2[0]; qi[jj] = b2[1];                                                                 \
            }                                                                                                   \
        } else {                                                                                                \
            const __amdgpu_buffer_rsrc_t r0_ = CT_RSRC(tp);                                                     \
            _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                  \
                const int j = 16 * (kt_) + 8 * hv + jj, m = R * j + n1;                                         \
                const bool okp = j < Mh, okq = okp && j >= 1 && 2 * j != M;                                     \
                int rowp = m > Nh ? N - m : m;                                                                  \
                rowp = okp ? rowp : 0;                                                                          \
                const int rowq = okq ? R * j - n1 : 0;                                                          \
                const f32x2 a2 = CT_BLOAD2(r0_, (unsigned)rowp * ldb4 + c4, 0);                                 \
                const f32x2 b2 = CT_BLOAD2(r0_, (unsigned)rowq * ldb4 + c4, 0);                                 \
                xr[jj] = a2[0]; xi[jj] = a2[1];                                                                 \
                qr[jj] = b2[0]; qi[jj] = b2[1];                                                                 \
            }                                                                                                   \
        }                                                                                                       \
    }
    // fold (and mix / unpack) the raw values into the two data streams of k-step kt_
#define CT_FOLD_PM(kt_, PV_)                                                                                     \
        _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                      \
            const int k = (kt_) * BK + 8 * hv + jj;                                                             \
            const bool pv = PV_;                                                                                \
            float a = xr[jj], b = qr[jj];                                                                       \
            if (HASPROD) {              /* own component of x * h (or x * conj h): x_own h_re +- x_other h_im */     \
                a = a * hx[jj][0] + psig * (pair_swap(a) * hx[jj][1]);                                          \
                b = b * hm[jj][0] + psig * (pair_swap(b) * hm[jj][1]);                                          \
            }                                                                                                   \
            if (LD == DFT_CT_PRODADD) { /* + (weight of the row + weight of the tile's k_beta) * third operand */  \
                const int mp = R * k + n1, mq = pv ? N - R * k + n1 : mp;                                       \
                a += (awk[mp] + awb) * xi[jj];                                                                  \
                b += (awk[mq] + awb) * qi[jj];                                                                  \
            }                                                                                                   \
            if (LD == DFT_CT_MIX) {     /* own component of (re + i im) * s */                                   \
                const float ap = pair_swap(a), bp = pair_swap(b);                                               \
                const int mp = R * k + n1, mq = pv ? N - R * k + n1 : mp;     /* rows of the full transform */   \
                const float4 m01 = mtab[2 * mp], m23 = mtab[2 * mp + 1], n01 = mtab[2 * mq], n23 = mtab[2 * mq + 1]; \
                const f32x2 sv = tw4.x * f32x2{m01.x, m01.y} + tw4.y * f32x2{m01.z, m01.w} + tw4.z * f32x2{m23.x, m23.y} + tw4.w * f32x2{m23.z, m23.w}; \
                const f32x2 uv = tw4.x * f32x2{n01.x, n01.y} + tw4.y * f32x2{n01.z, n01.w} + tw4.z * f32x2{n23.x, n23.y} + tw4.w * f32x2{n23.z, n23.w}; \
                a = a * sv[0] + sig * (ap * sv[1]);                                                             \
                b = b * uv[0] + sig * (bp * uv[1]);                                                             \
            }                                                                                                   \
            const float ev = a + (pv ? b : 0.f), od = pv ? a - b : 0.f;                                         \
            x0[jj] = ev;                                                                                        \
            x1[jj] = pair_swap(od);                                                                             \
        }
#define CT_FOLD(kt_)                                                                                            \
    {                                                                                                           \
        if (LD != DFT_CT_HPACK) {                                                                               \
            if ((kt_) >= 1 && 16 * (kt_) + 16 <= Mh - 1) {                                                      \
                CT_FOLD_PM(kt_, true)                                                                           \
            } else {                                                                                            \
                CT_FOLD_PM(kt_, (k >= 1) && (k < Mh) && (2 * k != M))                                           \
            }                                                                                                   \
        } else if (CT_HFAST(kt_)) {                                                                             \
            _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                  \
                /* Z = A + i B: re = Ar - Bi, im = Ai + Br; mirror from the conjugates: re = Ar + Bi, im = -Ai + Br */ \
                const float a = xr[jj] + sig * pair_swap(xi[jj]);                                               \
                const float b = qr[jj] - sig * pair_swap(qi[jj]);                                               \
                x0[jj] = a + b;                                                                                 \
                x1[jj] = pair_swap(a - b);                                                                      \
            }                                                                                                   \
        } else {                                                                                                \
            _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                  \
                const int j = 16 * (kt_) + 8 * hv + jj, m = R * j + n1;                                         \
                const bool okp = j < Mh, okq = okp && j >= 1 && 2 * j != M;                                     \
                const bool cj = m > Nh;                                                                         \
                const int rowp = cj ? N - m : m;                                                                \
                const bool zi = rowp == 0 || 2 * rowp == N;      /* imaginary parts of the self-conjugate rows are not part of the spectrum */ \
                const float ti = zi ? 0.f : pair_swap(xi[jj]), tq = pair_swap(qi[jj]);                          \
                const float a = okp ? xr[jj] + (cj ? -sig : sig) * ti : 0.f;                                    \
                const float b = okq ? qr[jj] - sig * tq : 0.f;                                                  \
                x0[jj] = a + b;                                                                                 \
                x1[jj] = pair_swap(okq ? a - b : 0.f);                                                          \
            }                                                                                                   \
        }                                                                                                       \
    }
    // exponent p of this column's largest folded magnitude (|x| < 2^p), over both streams and both k halves
#define CT_MAXEXP(p_)                                                                                           \
    {                                                                                                           \
        float m_ = 0.f;                                                                                         \
        _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) m_ = fmaxf(m_, fmaxf(fabsf(x0[jj]), fabsf(x1[jj])));   \
        const unsigned mu_ = __float_as_uint(m_);                                                               \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(mu_, mu_, false, false);                              \
        m_ = fmaxf(m_, fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1])));                                \
        p_ = __builtin_amdgcn_frexp_expf(m_);                                                                   \
    }
#define CT_MFMA(m_, kt_, acc_, bh_, bl_)                                                                        \
    {                                                                                                           \
        const unsigned short *ra = lds + ((m_) * 2 * KT + (kt_)) * PIECE + l31 * 16 + 8 * (h ^ ((l31 >> 3) & 1)); \
        _Pragma("unroll") for (int mt = 0; mt < MAXMT; ++mt) {                                                  \
            if (mt < MT) {          /* wave-uniform */                                                          \
                const unsigned short *p = ra + mt * 32 * 16;                                                    \
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(p);                                           \
                const f16x8 al = *reinterpret_cast<const f16x8 *>(p + KT * PIECE);                              \
                if (!(CT_EXP & 4)) MFMA3(acc_[mt], ah, al, bh_, bl_)                                            \
                else acc_[mt][0] += (float)bh_[mt] + (float)bl_[mt] + (float)ah[0] + (float)al[0];              \
            }                                                                                                   \
        }                                                                                                       \
    }

    int hv = h;
    float xr[8], qr[8], xi[8], qi[8];  // raw values: elements j and mirror elements (HPACK: both floats of the lane's 8 bytes)
    f32x2 hx[8], hm[8];                // PROD: (re, im) of the second operand at the elements / mirror elements
    (void)hx; (void)hm;
    float x0[8], x1[8];
    f16x8 c0h, c0l, c1h, c1l;          // fragments of the current k-step
    f32x16 acc1[MAXMT], acc2[MAXMT];
    (void)xi; (void)qi;

    if (LD == DFT_CT_MIX && ntw > 0) {
        const int t0_ = CT_TILE(u0);
        const int kb0 = g.batch > 1 ? t0_ / tilesX : ((t0_ % tilesX) * 16) / g.LP;
        CT_MIXTAB(kb0);
    }
    lds_barrier();                     // publishes the image, the twiddles and the first mix table
    if (ntw <= 0) return;              // all waves of a workgroup leave together

    int vu = u0;
    int tile = CT_TILE(vu);
    int e = 0, en = 0;                 // block exponents of the current / next k-step's column
    int phase = 0;                     // running count of exchange phases: the buffer in use is phase & 1

    CT_LSETUP(tile);
    CT_LOAD(0);
    CT_FSETUP(tile);
    CT_FOLD(0);
    {
        int p;
        CT_MAXEXP(p);
        e = E_TARGET - p;
    }
    split8h(x0, e, c0h, c0l);
    split8h(x1, e, c1h, c1l);
    CT_LOAD(1);
    while (true) {
        asm volatile("" : "+v"(hv));   // keeps the per-lane fold selectors from being hoisted out of the tile loop
#pragma unroll
        for (int i = 0; i < MAXMT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] = acc2[i][r] = 0.f;
        const bool more = vu + 1 < u1;
        const int next = more ? CT_TILE(vu + 1) : tile;
        const int chunk_ = g.tabLP ? (int)(((long)(tile % tilesX) * 16 % g.tabLP) >> g.tabShift) : 0;
        const int nkt = ktab ? ktab[chunk_] : KT;
        const int rlim = rtab ? rtab[chunk_] : N;          // output rows k with min(k, N - k) beyond it are not stored
        for (int kt = 0; kt + 1 < nkt; ++kt) {
            CT_MFMA(0, kt, acc1, c0h, c0l);
            CT_FOLD(kt + 1);
            int p;
            CT_MAXEXP(p);
            const bool need = p + e > E_LIMIT;
            en = need ? E_TARGET - p : e;
            const int d = en - e;
            if (kt + 2 < nkt) {
                CT_LOAD(kt + 2);
            } else if (more) {
                CT_LSETUP(next);
                CT_LOAD(0);
            }
            split8h(x0, en, c0h, c0l);
            CT_MFMA(1, kt, acc2, c1h, c1l);
            split8h(x1, en, c1h, c1l);
            if (__builtin_amdgcn_ballot_w64(d != 0) != 0ull) {
#pragma unroll
                for (int mt = 0; mt < MAXMT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc1[mt][r] = __builtin_amdgcn_ldexpf(acc1[mt][r], d);
                        acc2[mt][r] = __builtin_amdgcn_ldexpf(acc2[mt][r], d);
                    }
            }
            e = en;
        }
        CT_MFMA(0, nkt - 1, acc1, c0h, c0l);
        if (more) {
            CT_FSETUP(next);
            CT_FOLD(0);
            int p;
            CT_MAXEXP(p);
            en = E_TARGET - p;
            CT_LOAD(1);
            split8h(x0, en, c0h, c0l);
        }
        CT_MFMA(1, nkt - 1, acc2, c1h, c1l);
        if (more) split8h(x1, en, c1h, c1l);
        float exp_sink = 0.f;
        {
            // ---- epilogue: the R sub-transforms of a group meet in LDS, 16 rows r (and their mirrors M - r) per phase --------
            const int tx = tile % tilesX;
            const long bz = tile / tilesX;
            const float f = __builtin_amdgcn_ldexpf(g.scale, -e - kA);
            // Y_c[r] = C E_c + q S O_other,  Y_c[M - r] = C E_c - q S O_other,  q = sgn * (c ? 1 : -1)
            const float fq = f * g.sgn * sig;
            float *const dtile = (CT_EXP & 256) ? g.dst + (long)(tx & 7) * DST_T : g.dst + bz * g.sC + (long)tx * DST_T;
            const float sgs = g.sgn * sig;
            const int nph = (Mh + 15) / 16;
            // row pitch and lane half as values the compiler cannot see through: the per-row addresses below are loop invariants
            // it would otherwise keep in registers across the k loop (measured: 130-240 spilled VGPRs)
            long ldc_o = (CT_EXP & 256) ? 64 : g.ldc;         // 256 (tools/exp): every store into one small cache-resident region
            asm volatile("" : "+s"(ldc_o));
            const int he = hv;
            const unsigned ldc4 = (unsigned)ldc_o * 4u;
            const unsigned vlo = (unsigned)he * ldc4 + (unsigned)l31 * 4u, vmi = (unsigned)(1 - he) * ldc4 + (unsigned)l31 * 4u;
            // store descriptors span the two rows of a row pair; a masked lane (voffset) or a row pair beyond the row limit
            // (soffset) lands outside and is dropped by the range check: no exec-mask juggling around the stores
            const int wrange = (int)(ldc4 + 256u);
            constexpr unsigned VOOB = 0x7FFFFF00u;
            const long kstride = (long)M * ldc_o;
            // The phases run as a loop (one copy of the combine in the instruction stream); only the hand-over of the accumulators,
            // whose registers need static indices, is unrolled per phase.
#define CT_XWRITE(mt_, q2_)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                             \
        const int slot = (i & 3) + 8 * (i >> 2);       /* + 4 h (in xw) */                                      \
        const float a1 = acc1[mt_][8 * (q2_) + i], a2 = acc2[mt_][8 * (q2_) + i];                               \
        xw[slot * 32] = f * a1 + fq * a2;                                                                       \
        xw[(16 + slot) * 32] = f * a1 - fq * a2;                                                                \
    }
#pragma clang loop unroll(disable)
            for (int ph = 0; ph < nph; ++ph) {
                {
                    const int buf = phase & 1;
                    ++phase;
                    float *const xw = xbuf + (buf * NW + wave) * XFLOATS + l31 + 128 * he;
                    switch (ph) {
                        case 0: CT_XWRITE(0, 0) break;
                        case 1: CT_XWRITE(0, 1) break;
                        case 2: CT_XWRITE(1, 0) break;
                        case 3: CT_XWRITE(1, 1) break;
                        case 4: CT_XWRITE(2, 0) break;
                        default: CT_XWRITE(2, 1) break;
                    }
                    if (CT_EXP & 64) lds_barrier();
                    else if (!(CT_EXP & 16)) group_arrive_wait(gsync + grp, (unsigned)(R * phase), lane);
                    const float *const xg = xbuf + (buf * NW + grp * R) * XFLOATS + l31 + 32 * he;
                    // this wave's share of the phase's 8 row pairs: every LDS read of the phase is requested first (row pairs beyond the
                    // share read pair 7: in range, unused), so that the dependent chains of the pairs overlap
                    constexpr int NX = (8 + R - 1) / R;
                    f32x2 yall[NX][R], wpall[NX][R - 1], wmall[NX][R - 1];
#pragma unroll
                    for (int xi_ = 0; xi_ < NX; ++xi_) {
                        const int xx = n1 + R * xi_ < 8 ? n1 + R * xi_ : 7;
                        const int rr = 16 * ph + 2 * xx + he;
                        const bool okr_ = rr < Mh, okm_ = okr_ && rr >= 1 && 2 * rr != M;
                        const float *const twp = twl + (okr_ ? rr : 0) * ((R - 1) * 2), *const twm = twl + (okm_ ? M - rr : 0) * ((R - 1) * 2);
#pragma unroll
                        for (int n = 0; n < R; ++n) yall[xi_][n] = f32x2{xg[n * XFLOATS + (2 * xx) * 32], xg[n * XFLOATS + (16 + 2 * xx) * 32]};
#pragma unroll
                        for (int n = 1; n < R; ++n) {
                            wpall[xi_][n - 1] = *reinterpret_cast<const f32x2 *>(twp + (n - 1) * 2);
                            wmall[xi_][n - 1] = *reinterpret_cast<const f32x2 *>(twm + (n - 1) * 2);
                        }
                    }
#pragma unroll
                    for (int xi_ = 0; xi_ < NX; ++xi_) {
                        const int x = n1 + R * xi_;
                        const int r0 = 16 * ph + 2 * x;
                        if (x < 8 && r0 < Mh) {                    // wave-uniform
                            const int r = r0 + he;
                            const bool okr = r < Mh, okm = okr && r >= 1 && 2 * r != M;
                            const int rm = okm ? M - r : 0;
                            // (row r, mirror row M - r) travel as the two halves of one register pair: every operation below is one
                            // packed instruction for both (v_pk_mul / v_pk_fma / v_pk_add)
                            f32x2 y[R];
#pragma unroll
                            for (int n = 0; n < R; ++n) y[n] = yall[xi_][n];
                            // twiddles w_N^{n k2} (table: inverse sign), k2 = r and M - r
#pragma unroll
                            for (int n = 1; n < R; ++n) {
                                const f32x2 wp = wpall[xi_][n - 1], wm = wmall[xi_][n - 1];
                                const f32x2 wc = {wp[0], wm[0]}, ws = {sgs * wp[1], sgs * wm[1]};
                                const f32x2 ysw = {pair_swap(y[n][0]), pair_swap(y[n][1])};
                                y[n] = wc * y[n] + ws * ysw;
                            }
                            f32x2 ys[R];
#pragma unroll
                            for (int n = 1; n < R; ++n) ys[n] = f32x2{pair_swap(y[n][0]), pair_swap(y[n][1])};
                            f32x2 X[R];                            // (X[r + M k1], X[(M - r) + M k1])
#pragma unroll
                            for (int k1 = 0; k1 < R; ++k1) {
                                f32x2 acc = y[0];
#pragma unroll
                                for (int n = 1; n < R; ++n) {
                                    const float cr = wr_cos<R>((n * k1) % R), ci = wr_sin<R>((n * k1) % R);
                                    if (cr == 1.f) acc += y[n];
                                    else if (cr == -1.f) acc -= y[n];
                                    else if (cr != 0.f) acc += cr * y[n];
                                    if (ci != 0.f) acc += (sgs * ci) * ys[n];
                                }
                                X[k1] = acc;
                            }
                            float Xp[R], Xm[R];
#pragma unroll
                            for (int k1 = 0; k1 < R; ++k1) { Xp[k1] = X[k1][0]; Xm[k1] = X[k1][1]; }
                            if (CT_EXP & 1024) {        // tools/exp: no twiddles / butterflies, the LDS values go straight out
#pragma unroll
                                for (int k1 = 0; k1 < R; ++k1) { Xp[k1] = yall[xi_][k1][0]; Xm[k1] = yall[xi_][k1][1]; }
                            }
                            if ((CT_EXP & 2048) && (xi_ & 1)) {     // tools/exp: every other row pair of the wave's share not stored
#pragma unroll
                                for (int k1 = 0; k1 < R; ++k1) exp_sink += Xp[k1] * Xm[k1];
                                continue;
                            }
                            if (CT_EXP & 512) {         // tools/exp: the combine without its stores
#pragma unroll
                                for (int k1 = 0; k1 < R; ++k1) exp_sink += Xp[k1] * Xm[k1];
                                continue;
                            }
                            if (EP == DFT_CT_STORE) {
                                // rows (r0 + h) + M k1 and (M - r0 - 1 + (1 - h)) + M k1: a scalar running pointer per row pair, the half
                                // in the lane offset.  Interior pairs (both rows and both mirrors exist: all but the pair of row 0, the
                                // last one and, for even M, the one holding M / 2) store unmasked through global stores on a scalar base;
                                // the others through a buffer descriptor whose range drops the masked lanes.  A row pair is skipped
                                // (scalar branch) when both of its rows lie beyond the row limit.
                                const float *pp_ = dtile + (long)r0 * ldc_o, *pm_ = dtile + (long)(M - r0 - 1) * ldc_o;
                                const bool inner = r0 >= 1 && r0 + 1 < Mh && 2 * r0 != M && 2 * (r0 + 1) != M;
                                const unsigned vp_ = okr ? vlo : VOOB, vm_ = okm ? vmi : VOOB;
#pragma unroll
                                for (int k1 = 0; k1 < R; ++k1) {
                                    if (!(CT_EXP & 2)) {
                                        const int kp0 = r0 + M * k1, km0 = M - r0 - 1 + M * k1;        // first rows of the two pairs
                                        const int fp_ = kp0 < N - kp0 - 1 ? kp0 : N - kp0 - 1, fm_ = km0 < N - km0 - 1 ? km0 : N - km0 - 1;
                                        if (inner) {
                                            if (fp_ <= rlim) *reinterpret_cast<float *>(reinterpret_cast<char *>(const_cast<float *>(pp_)) + vlo) = Xp[k1];
                                            if (fm_ <= rlim) *reinterpret_cast<float *>(reinterpret_cast<char *>(const_cast<float *>(pm_)) + vmi) = Xm[k1];
                                        } else {
                                            const __amdgpu_buffer_rsrc_t wp_ = __builtin_amdgcn_make_buffer_rsrc((void *)pp_, 0, wrange, 0x00020000);
                                            const __amdgpu_buffer_rsrc_t wm_ = __builtin_amdgcn_make_buffer_rsrc((void *)pm_, 0, wrange, 0x00020000);
                                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Xp[k1]), wp_, (int)vp_, fp_ <= rlim ? 0 : 0x40000000, 0);
                                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Xm[k1]), wm_, (int)vm_, fm_ <= rlim ? 0 : 0x40000000, 0);
                                        }
                                    }
                                    pp_ += kstride;
                                    pm_ += kstride;
                                }
                            } else {
                                // the two Hermitian spectra of the packed pair: partner of k = r + M k1 is N - k = (M - r) + M (R - 1 - k1)
                                // (r = 0: M ((R - k1) % R); 2 r = M: r + M (R - 1 - k1)).  Row min(k, N - k) gets (A, B), conjugated if k > N - k;
                                // where both orders occur (r = 0, 2 r = M) only k <= N - k stores.
#pragma unroll
                                for (int k1 = 0; k1 < R; ++k1) {
                                    const int kp = r + M * k1, kq = N - kp;
                                    const float xq = (r == 0) ? Xp[(R - k1) % R] : (2 * r == M) ? Xp[R - 1 - k1] : Xm[R - 1 - k1];
                                    const float s = Xp[k1] + xq, d = Xp[k1] - xq;
                                    // lane c = 0 holds the real parts: A_re = s, B_im = -d;  lane c = 1 the imaginary parts: A_im = d, B_re = s
                                    const float ds = pair_swap(d);
                                    const bool cjg = kp > kq;
                                    const float v1 = ((var != 0) != cjg) ? -ds : ds;
                                    const bool st = okr && (okm || kp <= kq);
                                    const int row = cjg ? kq : kp;
                                    if (st && row <= rlim && !(CT_EXP & 2)) {
                                        const f32x2 v2 = {s, v1};
                                        *reinterpret_cast<f32x2 *>(dtile + (long)row * ldc_o + 2 * l31) = v2;
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        if ((CT_EXP & 512) && exp_sink == 12345.678f) g.dst[0] = exp_sink;
        if (!more) break;
        e = en;
        tile = next;
        ++vu;
    }
}

#undef CT_TILE
#undef CT_MIXTAB
#undef CT_FSETUP
#undef CT_RSRC
#undef CT_BLOAD
#undef CT_BLOAD2
#undef CT_LSETUP
#undef CT_HFAST
#undef CT_LOAD
#undef CT_FOLD_PM
#undef CT_FOLD
#undef CT_MAXEXP
#undef CT_MFMA

size_t ct_lds_bytes(int R, int M, int MT, int KT, bool mix, bool addtab = false) {
    const int NW = (R == 2 ? 4 : 2) * R;
    size_t b = (size_t)4 * KT * (32 * MT * 16) * 2;                        // image
    b += (size_t)2 * NW * XFLOATS * 4;                                    // exchange buffers
    b += (size_t)(((M * (R - 1) * 2) + 3) & ~3) * 4 + 16;                 // twiddles, group counters
    if (mix) b += (size_t)2 * (R * M) * 2 * sizeof(float4);               // two mix tables
    if (addtab) b += (size_t)(R * M) * sizeof(float);                     // PRODADD: row weights
    return b;
}

template <int R, int LD, int EP>
int launch_inst(hipStream_t stream, const DftCtArgs &g, const DftCtPlan &pl, int NU, int cus) {
    static unsigned long long done = 0;
    const size_t ldsb = ct_lds_bytes(R, pl.M, pl.MT, pl.KT, LD == DFT_CT_MIX, LD == DFT_CT_PRODADD);
    if (ldsb > LDS_LIMIT) return (int)hipErrorInvalidValue;
    if (int e = ensure_dynamic_lds(dft_ct_kernel<R, LD, EP>, ldsb, done)) return e;
    const dim3 grid((unsigned)(NU < cus ? NU : cus));
    hipLaunchKernelGGL((dft_ct_kernel<R, LD, EP>), grid, dim3(Cfg<R>::NTH), ldsb, stream, g, reinterpret_cast<const uint4 *>(pl.img), pl.tw,
                       g.vlist, g.ktab, g.rtab, pl.MT, pl.KT, pl.kA, NU);
    return (int)hipGetLastError();
}

template <int R>
int launch_r(hipStream_t stream, const DftCtArgs &g, const DftCtPlan &pl, int NU, int cus) {
    if (g.loader == DFT_CT_PLAIN && g.epi == DFT_CT_STORE) return launch_inst<R, DFT_CT_PLAIN, DFT_CT_STORE>(stream, g, pl, NU, cus);
    if (g.loader == DFT_CT_MIX && g.epi == DFT_CT_STORE) return launch_inst<R, DFT_CT_MIX, DFT_CT_STORE>(stream, g, pl, NU, cus);
    if (g.loader == DFT_CT_PROD && g.epi == DFT_CT_STORE) return launch_inst<R, DFT_CT_PROD, DFT_CT_STORE>(stream, g, pl, NU, cus);
    if (g.loader == DFT_CT_PRODADD && g.epi == DFT_CT_STORE) return launch_inst<R, DFT_CT_PRODADD, DFT_CT_STORE>(stream, g, pl, NU, cus);
    if (g.loader == DFT_CT_HPACK && g.epi == DFT_CT_STORE) return launch_inst<R, DFT_CT_HPACK, DFT_CT_STORE>(stream, g, pl, NU, cus);
    if (g.loader == DFT_CT_PLAIN && g.epi == DFT_CT_HSEP) return launch_inst<R, DFT_CT_PLAIN, DFT_CT_HSEP>(stream, g, pl, NU, cus);
    return (int)hipErrorInvalidValue;
}

}  // namespace

bool dft_ct_factor(int n, int *Rout, int *Mout) {
    for (int R : {4, 3, 2}) {
        if (n % R) continue;
        const int M = n / R, Mh = M / 2 + 1, KT = (Mh + 15) / 16, MT = (Mh + 31) / 32;
        if (M <= 32 || MT > MAXMT || KT < 2 || 16 * KT > M) continue;
        if (ct_lds_bytes(R, M, MT, KT, true) > LDS_LIMIT) continue;
        if (Rout) *Rout = R;
        if (Mout) *Mout = M;
        return true;
    }
    return false;
}

bool dft_ct_supported(int Na, int Nb) { return dft_ct_factor(Na, nullptr, nullptr) && dft_ct_factor(Nb, nullptr, nullptr); }

int dft_ct_plan_create(int n, DftCtPlan *out) {
    DftCtPlan p;
    if (!out || !dft_ct_factor(n, &p.R, &p.M)) return (int)hipErrorInvalidValue;
    p.N = n;
    const int M = p.M, Mh = M / 2 + 1;
    p.KT = (Mh + 15) / 16;
    p.MT = (Mh + 31) / 32;
    const int PIECE = 32 * p.MT * 16;
    std::vector<unsigned short> img((size_t)4 * p.KT * PIECE, 0);
    p.kA = 13;                                                   // |cos|, |sin| <= 1 < 2^1: pieces below 2^14
    for (int m = 0; m < 2; ++m)
        for (int r = 0; r < Mh; ++r)
            for (int k = 0; k < Mh; ++k) {
                const double th = 2.0 * M_PI * (double)(((long)r * k) % M) / (double)M;
                const float xs = std::ldexp((float)(m ? std::sin(th) : std::cos(th)), p.kA);
                const _Float16 hh = (_Float16)xs;
                const _Float16 ll = (_Float16)(xs - (float)hh);
                const int kt = k / BK, c = (k % BK) / 8, j = k % 8;
                const size_t pos = (size_t)kt * PIECE + (size_t)r * 16 + 8 * (c ^ ((r >> 3) & 1)) + j;
                unsigned short uh, ul;
                std::memcpy(&uh, &hh, 2);
                std::memcpy(&ul, &ll, 2);
                img[((size_t)(m * 2 + 0) * p.KT) * PIECE + pos] = uh;
                img[((size_t)(m * 2 + 1) * p.KT) * PIECE + pos] = ul;
            }
    std::vector<float> tw((size_t)M * (p.R - 1) * 2);
    for (int k2 = 0; k2 < M; ++k2)
        for (int n1 = 1; n1 < p.R; ++n1) {
            const double th = 2.0 * M_PI * (double)(((long)n1 * k2) % n) / (double)n;
            tw[((size_t)k2 * (p.R - 1) + n1 - 1) * 2] = (float)std::cos(th);
            tw[((size_t)k2 * (p.R - 1) + n1 - 1) * 2 + 1] = (float)std::sin(th);
        }
    if (hipMalloc((void **)&p.img, img.size() * 2) != hipSuccess) return (int)hipErrorOutOfMemory;
    if (hipMalloc((void **)&p.tw, tw.size() * 4) != hipSuccess) { hipFree(p.img); return (int)hipErrorOutOfMemory; }
    if (hipMemcpy(p.img, img.data(), img.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p.tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(p.img);
        hipFree(p.tw);
        return (int)hipErrorUnknown;
    }
    *out = p;
    return 0;
}

void dft_ct_plan_destroy(DftCtPlan *p) {
    if (!p) return;
    hipFree(p->img);
    hipFree(p->tw);
    *p = DftCtPlan();
}

int launch_dft_ct(hipStream_t stream, const DftCtArgs &g, const DftCtPlan &pl) {
    if (!pl.img || !pl.tw || g.R != pl.R || g.M != pl.M || !g.src || !g.dst || g.ncols < 64 || g.ncols % 64 || g.batch < 1)
        return (int)hipErrorInvalidValue;
    if (g.loader == DFT_CT_MIX && (!g.mhat || !g.tpl || g.T < 1 || g.T > 4 || g.LP % 128)) return (int)hipErrorInvalidValue;
    if ((g.loader == DFT_CT_PROD || g.loader == DFT_CT_PRODADD) &&
        (!g.prod || g.ldp <= 0 || 16.0 * (double)g.R * (double)g.ldp * 4.0 + 1024.0 >= 4294967296.0))
        return (int)hipErrorInvalidValue;
    if (g.loader == DFT_CT_PRODADD && (!g.add || g.add_Nb < 1 || (g.batch == 1 && (g.LP < 16 || g.LP % 16)))) return (int)hipErrorInvalidValue;
    if ((g.loader == DFT_CT_HPACK && g.sgn < 0.f) || (g.epi == DFT_CT_HSEP && g.sgn > 0.f)) return (int)hipErrorInvalidValue;
    // 32-bit offsets inside a k-step: 16 rows of a sub-sequence; HPACK's edge steps address rows 0 .. N / 2 from the tile's first row
    if (16.0 * (double)g.R * (double)g.ldb * 4.0 + 1024.0 >= 4294967296.0) return (int)hipErrorInvalidValue;
    if (g.loader == DFT_CT_HPACK && ((double)(pl.N / 2) + 1.0) * (double)g.ldb * 4.0 + 1024.0 >= 4294967296.0) return (int)hipErrorInvalidValue;
    if ((g.ktab || g.rtab) && (g.tabLP < (1 << g.tabShift) || g.tabLP % (1 << g.tabShift) || g.tabShift < 4 || g.tabShift > 7)) return (int)hipErrorInvalidValue;
    static int cus_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (!cus_of[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return (int)hipErrorInvalidDevice;
        cus_of[dev] = cus;
    }
    const int NG = g.R == 2 ? 4 : 2;
    long NU = (long)(g.ncols / (16 * NG)) * g.batch;
    if (g.vlist) {
        const long NS = (long)(g.ncols / 128) * g.batch;
        if (g.ncols % 128 || g.nvalid < 1 || g.nvalid > NS) return (int)hipErrorInvalidValue;
        NU = (long)g.nvalid * (128 / (16 * NG));
    }
    if (NU >= 2147483647L / 16) return (int)hipErrorInvalidValue;
    switch (g.R) {
        case 2: return launch_r<2>(stream, g, pl, (int)NU, cus_of[dev]);
        case 3: return launch_r<3>(stream, g, pl, (int)NU, cus_of[dev]);
        case 4: return launch_r<4>(stream, g, pl, (int)NU, cus_of[dev]);
    }
    return (int)hipErrorInvalidValue;
}
