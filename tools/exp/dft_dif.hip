// Experiment: the complex pass of length N = R * M as a decimation-in-FREQUENCY step with independent waves.
//
//   X[R k2 + k1] = sum_{j < M} w_M^{j k2} u_{k1}[j],     u_{k1}[j] = w_N^{j k1} sum_{n1 < R} w_R^{n1 k1} x[j + M n1]
//
// A wave job = (tile of 16 complex columns, residue k1): the wave loads ALL R rows j + M n1 of every element (the R jobs of a
// tile read the same rows: they run on neighbouring waves of one workgroup, the repeats come from L1 / L2), forms u in the
// loader (R-point butterfly and twiddle in VALU), runs the folded M-point transform of dft_ct.hip on it and stores its own M
// output rows R k2 + k1 straight from the accumulators: no exchange through LDS, no synchronisation between waves -- what
// dft_ct.hip pays for with its combine phase.  Same image / twiddle tables (DftCtPlan).  c2c only (see tools/exp/README.md).
#include "../../surfh_amd/csrc/dft_ct.h"
#include "../../surfh_amd/csrc/lds_attr.h"
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BK = 16, MAXMT = 3, NWAVES = 8, NTH = 64 * NWAVES;
constexpr int E_TARGET = 10, E_LIMIT = 15;

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
__device__ __forceinline__ float pair_swap(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
}

template <int R>
__global__ __launch_bounds__(NTH) void dft_dif_kernel(DftCtArgs g, const uint4 *__restrict__ img, const float *__restrict__ twg, int MT, int KT,
                                                      int kA, long NJ) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int var = l31 & 1;
    const float sig = var ? 1.f : -1.f;
    const float sgs = g.sgn * sig;
    const int M = g.M, Mh = M / 2 + 1;
    const int PIECE = 32 * MT * 16, IMGH = 4 * KT * PIECE;
    const int tilesX = g.ncols / 16;
    const unsigned ldb4 = (unsigned)(g.ldb * 4);
    const unsigned c4 = (unsigned)l31 * 4u;
    f32x2 *const twd = reinterpret_cast<f32x2 *>(lds + IMGH);        // [R][M] (cos, sin) of 2 pi k1 j / N
    {
        uint4 *l4 = reinterpret_cast<uint4 *>(lds);
        for (int i = tid; i < IMGH / 8; i += NTH) l4[i] = img[i];
        for (int i = tid; i < R * M; i += NTH) {
            const int k1 = i / M, j = i % M;
            twd[i] = k1 == 0 ? f32x2{1.f, 0.f} : f32x2{twg[(j * (R - 1) + (k1 - 1)) * 2], twg[(j * (R - 1) + (k1 - 1)) * 2 + 1]};
        }
    }
    // this workgroup's contiguous range of jobs (tile * R + k1); wave w takes j0 + w, j0 + w + 8, ...
    const long j0 = NJ * blockIdx.x / gridDim.x, j1 = NJ * (blockIdx.x + 1) / gridDim.x;
    __syncthreads();
    long job = j0 + wave;
    if (job >= j1) return;

#define DF_RSRC(ptr_) __builtin_amdgcn_make_buffer_rsrc((void *)(ptr_), 0, 0xFFFFFFFF, 0x00020000)
#define DF_BLOAD(r_, v_, s_) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_, (int)(v_), (int)(s_), 0))
    const float *tp = g.src;
    int k1 = 0;
    float cr[R], ci[R];                // w_R^{n k1}: cos, sin (unsigned)
    const f32x2 *twk = twd;
#define DF_JSETUP(job_)                                                                                         \
    {                                                                                                           \
        const long t_ = (job_) / R;                                                                             \
        k1 = (int)((job_) % R);                                                                                 \
        const int tx = (int)(t_ % tilesX);                                                                      \
        const long bz_ = t_ / tilesX;                                                                           \
        tp = g.src + bz_ * g.sB + (long)tx * 32;                                                                \
        _Pragma("unroll") for (int n = 1; n < R; ++n) {                                                         \
            const int q = (n * k1) % R;                                                                         \
            cr[n] = (R == 3) ? (q == 0 ? 1.f : -0.5f) : (R == 4) ? (q == 0 ? 1.f : q == 2 ? -1.f : 0.f) : (q == 0 ? 1.f : -1.f); \
            ci[n] = (R == 3) ? (q == 0 ? 0.f : q == 1 ? 0.86602540378443865f : -0.86602540378443865f)          \
                             : (R == 4) ? (q == 1 ? 1.f : q == 3 ? -1.f : 0.f) : 0.f;                           \
        }                                                                                                       \
        twk = twd + k1 * M;                                                                                     \
    }
    // raw loads of k-step kt_: elements j = 16 kt + 8 h + jj (rows j + M n) and their mirrors M - j (rows M - j + M n)
#define DF_LOAD(kt_)                                                                                            \
    {                                                                                                           \
        const unsigned vk = (unsigned)(8 * hv) * ldb4 + c4, vq = (unsigned)(8 * (1 - hv)) * ldb4 + c4;          \
        _Pragma("unroll") for (int n = 0; n < R; ++n) {                                                         \
            const __amdgpu_buffer_rsrc_t rp_ = DF_RSRC(tp + (long)(16 * (kt_) + M * n) * g.ldb);                \
            const __amdgpu_buffer_rsrc_t rq_ = DF_RSRC(tp + (long)(M - 16 * (kt_) - 15 + M * n) * g.ldb);       \
            _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                  \
                xr[n][jj] = DF_BLOAD(rp_, vk, (unsigned)jj * ldb4);                                             \
                /* element 0 has no mirror (row M + M n may lie behind the array): both halves read row M - 8 + M n */ \
                qr[n][jj] = DF_BLOAD(rq_, (jj == 0 && (kt_) == 0) ? c4 : vq, (unsigned)(7 - jj) * ldb4);        \
            }                                                                                                   \
        }                                                                                                       \
    }
    // butterfly + twiddle + fold into the two data streams of k-step kt_
#define DF_FOLD(kt_)                                                                                            \
    {                                                                                                           \
        _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                      \
            const int j = (kt_) * BK + 8 * hv + jj;                                                             \
            const bool pv = (j >= 1) && (j < Mh) && (2 * j != M);                                               \
            float a = xr[0][jj], b = qr[0][jj];                                                                 \
            _Pragma("unroll") for (int n = 1; n < R; ++n) {                                                     \
                const float xa = xr[n][jj], xb = qr[n][jj];                                                     \
                a += xa * cr[n] + (sgs * ci[n]) * pair_swap(xa);                                                \
                b += xb * cr[n] + (sgs * ci[n]) * pair_swap(xb);                                                \
            }                                                                                                   \
            const f32x2 wa = twk[j < M ? j : 0], wb = twk[pv ? M - j : 0];                                      \
            a = a * wa[0] + (sgs * wa[1]) * pair_swap(a);                                                       \
            b = b * wb[0] + (sgs * wb[1]) * pair_swap(b);                                                       \
            const float ev = a + (pv ? b : 0.f), od = pv ? a - b : 0.f;                                         \
            x0[jj] = ev;                                                                                        \
            x1[jj] = pair_swap(od);                                                                             \
        }                                                                                                       \
    }
#define DF_MAXEXP(p_)                                                                                           \
    {                                                                                                           \
        float m_ = 0.f;                                                                                         \
        _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) m_ = fmaxf(m_, fmaxf(fabsf(x0[jj]), fabsf(x1[jj])));   \
        const unsigned mu_ = __float_as_uint(m_);                                                               \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(mu_, mu_, false, false);                              \
        m_ = fmaxf(m_, fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1])));                                \
        p_ = __builtin_amdgcn_frexp_expf(m_);                                                                   \
    }
#define DF_MFMA(m_, kt_, acc_, bh_, bl_)                                                                        \
    {                                                                                                           \
        const unsigned short *ra = lds + ((m_) * 2 * KT + (kt_)) * PIECE + l31 * 16 + 8 * (h ^ ((l31 >> 3) & 1)); \
        _Pragma("unroll") for (int mt = 0; mt < MAXMT; ++mt) {                                                  \
            if (mt < MT) {                                                                                      \
                const unsigned short *p = ra + mt * 32 * 16;                                                    \
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(p);                                           \
                const f16x8 al = *reinterpret_cast<const f16x8 *>(p + KT * PIECE);                              \
                MFMA3(acc_[mt], ah, al, bh_, bl_)                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }

    int hv = h;
    float xr[R][8], qr[R][8];
    float x0[8], x1[8];
    f16x8 c0h, c0l, c1h, c1l;
    f32x16 acc1[MAXMT], acc2[MAXMT];
    int e = 0, en = 0;

    DF_JSETUP(job);
    DF_LOAD(0);
    DF_FOLD(0);
    {
        int p;
        DF_MAXEXP(p);
        e = E_TARGET - p;
    }
    split8h(x0, e, c0h, c0l);
    split8h(x1, e, c1h, c1l);
    DF_LOAD(1);
    while (true) {
        asm volatile("" : "+v"(hv));
#pragma unroll
        for (int i = 0; i < MAXMT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] = acc2[i][r] = 0.f;
        const bool more = job + NWAVES < j1;
        const long tile = job / R;
        const int k1c = k1;            // residue of the job whose accumulators are being filled
        for (int kt = 0; kt + 1 < KT; ++kt) {
            DF_MFMA(0, kt, acc1, c0h, c0l);
            DF_FOLD(kt + 1);
            int p;
            DF_MAXEXP(p);
            const bool need = p + e > E_LIMIT;
            en = need ? E_TARGET - p : e;
            const int d = en - e;
            if (kt + 2 < KT) {
                DF_LOAD(kt + 2);
            } else if (more) {
                DF_JSETUP(job + NWAVES);
                DF_LOAD(0);
            }
            split8h(x0, en, c0h, c0l);
            DF_MFMA(1, kt, acc2, c1h, c1l);
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
        DF_MFMA(0, KT - 1, acc1, c0h, c0l);
        if (more) {
            DF_FOLD(0);
            int p;
            DF_MAXEXP(p);
            en = E_TARGET - p;
            DF_LOAD(1);
            split8h(x0, en, c0h, c0l);
        }
        DF_MFMA(1, KT - 1, acc2, c1h, c1l);
        if (more) split8h(x1, en, c1h, c1l);
        {
            // epilogue: rows R k2 + k1 (and R (M - k2) + k1) straight from the accumulators; k2 = mt * 32 + (i & 3) + 8 (i >> 2) + 4 h
            const int tx = (int)(tile % tilesX);
            const long bz = tile / tilesX;
            const float f = __builtin_amdgcn_ldexpf(g.scale, -e - kA);
            const float fq = f * g.sgn * sig;
            long ldc_o = g.ldc;
            asm volatile("" : "+s"(ldc_o));
            char *const dt = reinterpret_cast<char *>(g.dst + bz * g.sC + (long)tx * 32 + (long)k1c * ldc_o);
            const int he = hv;
            const unsigned rstep = (unsigned)(R * ldc_o) * 4u;               // bytes between output rows of consecutive k2
            const unsigned vlo = (unsigned)(4 * he) * rstep + c4, vmi = (unsigned)(4 * (1 - he)) * rstep + c4;
#pragma unroll
            for (int mt = 0; mt < MAXMT; ++mt) {
                if (mt >= MT) break;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int kb = mt * 32 + (i & 3) + 8 * (i >> 2);         // k2 of lane half 0
                    const int k2 = kb + 4 * he;
                    const float a1 = acc1[mt][i], a2 = acc2[mt][i];
                    // row pointer of (kb, half 0); the mirror rows of halves (0, 1) are M - kb and M - kb - 4: base at the lower one
                    char *const pp = dt + (long)kb * R * ldc_o * 4;
                    char *const pm = dt + (long)(M - kb - 4) * R * ldc_o * 4;
                    if (k2 < Mh) *reinterpret_cast<float *>(pp + vlo) = f * a1 + fq * a2;
                    if (k2 >= 1 && k2 < Mh && 2 * k2 != M) *reinterpret_cast<float *>(pm + vmi) = f * a1 - fq * a2;
                }
            }
        }
        if (!more) break;
        e = en;
        job += NWAVES;
    }
}

}  // namespace

int launch_dft_dif(hipStream_t stream, const DftCtArgs &g, const DftCtPlan &pl) {
    if (g.loader != DFT_CT_PLAIN || g.epi != DFT_CT_STORE || g.R != pl.R || g.M != pl.M || g.ncols % 16) return (int)hipErrorInvalidValue;
    const size_t ldsb = (size_t)4 * pl.KT * (32 * pl.MT * 16) * 2 + (size_t)g.R * g.M * 8;
    const long NJ = (long)(g.ncols / 16) * g.batch * g.R;
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const dim3 grid((unsigned)(NJ / NWAVES < cus ? (NJ + NWAVES - 1) / NWAVES : cus));
#define DIF_GO(R_)                                                                                                                    \
    {                                                                                                                                 \
        hipFuncSetAttribute(reinterpret_cast<const void *>(dft_dif_kernel<R_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
        hipLaunchKernelGGL((dft_dif_kernel<R_>), grid, dim3(NTH), ldsb, stream, g, reinterpret_cast<const uint4 *>(pl.img), pl.tw, pl.MT, pl.KT,  \
                           pl.kA, NJ);                                                                                                \
    }
    if (g.R == 3) DIF_GO(3) else if (g.R == 4) DIF_GO(4) else if (g.R == 2) DIF_GO(2) else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}
