// Register-direct split-bf16 DFT pass (see dft_rx3.h).  One workgroup = 4 waves; each wave owns all 128 output
// rows of 32 columns (accumulators: 2 products x 4 row tiles x 16 = 128 registers), so every data element is
// loaded, folded and split by exactly one lane.  The matrix tiles (3 bf16 pieces x 2 matrices x 128 rows x 16 k)
// are double-buffered in LDS, filled by LDS-DMA into an XOR-swizzled image: each A fragment is one conflict-free
// ds_read_b128.  Workgroups are persistent (two per CU) and pipeline loads / MFMAs / stores across tile seams.
#include "dft_rx3.h"
#include "../../surfh_amd/csrc/lds_attr.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BK = 16, RS = 16;               // unpadded rows: the image is written by LDS-DMA (lane-linear)
constexpr int PIECE = 128 * RS;               // one bf16 piece of one 128-row matrix tile (4 KB)
constexpr int IMG = 3 * PIECE;
constexpr int BUF = 2 * IMG;                  // both matrices
constexpr size_t LDS_BYTES = (size_t)2 * BUF * sizeof(unsigned short);
constexpr int MIX_ROWS_MAX = 2048;            // rows of the per-workgroup spectral-mix table ([k][re/im] float4): 64 KB at most

__device__ __forceinline__ unsigned pack2(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// exact split of 8 values into three bf16x8 fragments (h, m, l)
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &fh, bf16x8 &fm, bf16x8 &fl) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned u = __float_as_uint(x[j]);
        h[j] = u & 0xFFFF0000u;
        const float r = x[j] - __uint_as_float(h[j]);
        m[j] = __float_as_uint(r) & 0xFFFF0000u;
        l[j] = __float_as_uint(r - __uint_as_float(m[j]));
    }
    fh = __builtin_bit_cast(bf16x8, make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7])));
    fm = __builtin_bit_cast(bf16x8, make_uint4(pack2(m[0], m[1]), pack2(m[2], m[3]), pack2(m[4], m[5]), pack2(m[6], m[7])));
    fl = __builtin_bit_cast(bf16x8, make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7])));
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
//      2: two streams, no fold (c2r)
template <int KIND, bool MIX>
__global__ __launch_bounds__(256, 2) void dft_rx3_kernel(DftRx3Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const bool packed = (KIND == 0) && g.packed;
    const int TN = packed ? 64 : 128;                          // columns of a tile
    const int var = packed ? (l31 >> 4) : 0;                   // packed: which variant this lane's column copy carries
    const int lcol = packed ? wave * 16 + (l31 & 15) : wave * 32 + l31;
    const int tilesX = g.N / TN, tilesY = g.MP / 128;
    const int ntile = tilesX * tilesY * g.batch * (packed ? 1 : g.nvar);     // units: (tile, variant), variant fastest
    const unsigned ldb4 = (unsigned)(g.ldb * 4), ldc4 = (unsigned)(g.ldc * 4), c4 = (unsigned)lcol * 4u;
    const long ldbB = g.ldb * 4, ldcB = g.ldc * 4;
    const int nk = g.KP / BK;
    const int kin = g.Kn / 2 + 1;
    const int arow = tid >> 1;                                 // = 32 * wave + (lane >> 1)
    const unsigned aoff = (unsigned)(arow * g.lda + 8 * ((lane & 1) ^ ((arow >> 3) & 1))) * 2u;
    float4 *mtab = reinterpret_cast<float4 *>(lds + 2 * BUF);   // [k][re/im] x 4 templates: this tile's mhat column

    // The workgroup is persistent: it walks tiles t = blockIdx.x, blockIdx.x + gridDim.x, ... and the software
    // pipeline runs across tile boundaries -- the first loads of the next tile are issued before the last
    // k-step's matrix work and the epilogue stores of the current tile, so that loads, stores and MFMAs overlap.
    // unit id = tile index * nvar + variant.  Large launches: a workgroup does all variants of its tiles back to back;
    // launches with fewer tiles than workgroup slots (`strided`): units are dealt one by one so every slot gets work
    int tile = (g.strided || packed) ? (int)blockIdx.x : (int)blockIdx.x * g.nvar;
    // per-tile state (uniform): uniform base + 32-bit lane offset addressing (the launcher checks that the rows fit
    // in 2^31 bytes): one scalar pair and one VGPR per address instead of 64-bit vector arithmetic
    const char *B0, *B1, *A0, *A1;
    int n0, m0;
    long bz;
    float fo0, fo1;                    // fold signs of the unit being loaded / folded
    float4 tw = make_float4(0.f, 0.f, 0.f, 0.f);
#define RX_SETUP(t_)                                                                                            \
    {                                                                                                           \
        const int alt_ = packed ? 0 : (t_) % g.nvar, tt_ = packed ? (t_) : (t_) / g.nvar;                       \
        const int tx = tt_ % tilesX, ty = (tt_ / tilesX) % tilesY;                                              \
        bz = tt_ / (tilesX * tilesY);                                                                           \
        fo0 = alt_ ? g.fold_alt[0] : g.fold[0];                                                                 \
        fo1 = alt_ ? g.fold_alt[1] : g.fold[1];                                                                 \
        n0 = tx * TN;                                                                                           \
        m0 = ty * 128;                                                                                          \
        B0 = reinterpret_cast<const char *>(g.src[0] + bz * g.sB + n0);                                         \
        B1 = reinterpret_cast<const char *>(g.src[1] + bz * g.sB + n0);                                         \
        A0 = reinterpret_cast<const char *>((alt_ ? g.A_alt[0] : g.A[0]) + (long)m0 * g.lda);                   \
        A1 = reinterpret_cast<const char *>((alt_ ? g.A_alt[1] : g.A[1]) + (long)m0 * g.lda);                   \
    }
    // fused spectral mix: the (k, kb) column of mhat for this tile's kb, all k, into LDS; template weights of this lane
#define RX_MIXTAB()                                                                                             \
    {                                                                                                           \
        const int kb = n0 / g.LP;                                                                               \
        const int l = (n0 % g.LP) + lcol;                                                                       \
        float t4[4];                                                                                            \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) t4[t] = (t < g.T) ? g.tpl[(long)t * g.LP + l] : 0.f;      \
        tw = make_float4(t4[0], t4[1], t4[2], t4[3]);                                                           \
        const int ne = (g.Kn > g.KP ? g.Kn : g.KP) * 2;                                                         \
        for (int e = tid; e < ne; e += 256) {                                                                   \
            const int k = e >> 1, c = e & 1;                                                                    \
            float v[4];                                                                                         \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                       \
                v[t] = (t < g.T && k < g.Kn) ? g.mhat[((long)t * 2 + c) * g.PL + (long)k * g.KBP + kb] : 0.f;   \
            mtab[e] = make_float4(v[0], v[1], v[2], v[3]);                                                      \
        }                                                                                                       \
    }

    f32x16 acc1[4], acc2[4];

    // raw loads of this lane's 8 k of tile kt_ (k = 16 kt + 8 h + j) and of their mirror rows; branch-free so that
    // all of them are in flight together
#define RX_BLOAD(kt_)                                                                                           \
    {                                                                                                           \
        /* row part of every address in 64-bit scalar pointers, lane part (k half and column) in one small VGPR: no  \
           limit on the array size.  Mirror row of k = 16 kt + 8 h + j is (Kn - 16 kt - 8 - j) + 8 (1 - h); it is read  \
           unconditionally (its weight is zero where there is no mirror), except k = 0 whose "mirror" Kn may not exist */ \
        const char *rk0 = B0 + (long)((kt_) * BK) * ldbB, *rk1 = B1 + (long)((kt_) * BK) * ldbB;                 \
        const char *rp0 = B0 + (long)(g.Kn - (kt_) * BK - 8) * ldbB, *rp1 = B1 + (long)(g.Kn - (kt_) * BK - 8) * ldbB; \
        const unsigned vk = (unsigned)(8 * hv) * ldb4 + c4, vp = (unsigned)(8 * (1 - hv)) * ldb4 + c4;           \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                         \
            xr[j] = *reinterpret_cast<const float *>(rk0 + j * ldbB + vk);                                      \
            if (KIND != 1) xi[j] = *reinterpret_cast<const float *>(rk1 + j * ldbB + vk);                       \
            if (KIND != 2) {                                                                                    \
                const unsigned q = (j == 0 && (kt_) == 0) ? c4 : vp;                                            \
                qr[j] = *reinterpret_cast<const float *>(rp0 - j * ldbB + q);                                   \
                if (KIND != 1) qi[j] = *reinterpret_cast<const float *>(rp1 - j * ldbB + q);                    \
            }                                                                                                   \
        }                                                                                                       \
    }
    // fold (and mix) the raw values into the two data streams of tile kt_
#define RX_BFOLD(kt_)                                                                                           \
    {                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                         \
            const int k = (kt_) * BK + 8 * hv + j;                                                               \
            float ar = xr[j], ai = (KIND == 1) ? xr[j] : xi[j];                                                 \
            if (MIX) {                                                                                          \
                const float4 mr = mtab[2 * k], mi = mtab[2 * k + 1];                                            \
                const float sr = tw.x * mr.x + tw.y * mr.y + tw.z * mr.z + tw.w * mr.w;                         \
                const float si = tw.x * mi.x + tw.y * mi.y + tw.z * mi.z + tw.w * mi.w;                         \
                const float hr = ar, hi = ai;                                                                   \
                ar = hr * sr - hi * si;                                                                         \
                ai = hr * si + hi * sr;                                                                         \
            }                                                                                                   \
            if (KIND == 2) {                                                                                    \
                x0[j] = ar;                                                                                     \
                x1[j] = ai;                                                                                     \
            } else {                                                                                            \
                const bool pv = (k >= 1) && (k < kin) && (2 * k != g.Kn);                                       \
                float br = qr[j], bi = (KIND == 1) ? qr[j] : qi[j];                                             \
                if (MIX) {                                                                                      \
                    const int kp = pv ? g.Kn - k : k;                                                           \
                    const float4 mr = mtab[2 * kp], mi = mtab[2 * kp + 1];                                      \
                    const float sr = tw.x * mr.x + tw.y * mr.y + tw.z * mr.z + tw.w * mr.w;                     \
                    const float si = tw.x * mi.x + tw.y * mi.y + tw.z * mi.z + tw.w * mi.w;                     \
                    const float hr = br, hi = bi;                                                               \
                    br = hr * sr - hi * si;                                                                     \
                    bi = hr * si + hi * sr;                                                                     \
                }                                                                                               \
                const float g0 = var ? g.fold_alt[0] : fo0, g1 = var ? g.fold_alt[1] : fo1;                     \
                const float f0 = pv ? g0 : 0.f, f1 = pv ? g1 : 0.f;                                             \
                const float w0 = (!pv && g0 < 0.f) ? 0.f : 1.f, w1 = (!pv && g1 < 0.f) ? 0.f : 1.f;             \
                const float s0 = w0 * ar + f0 * br, s1v = w1 * ai + f1 * bi;                                    \
                x0[j] = var ? s1v : s0;       /* the second variant pairs its streams with the other matrix */   \
                x1[j] = var ? s0 : s1v;                                                                         \
            }                                                                                                   \
        }                                                                                                       \
    }
    // matrix tiles: global -> LDS by DMA (no registers).  One wave-instruction writes 64 x 16 B = 32 rows of one piece,
    // lane-linear; position 2*row + c holds the k-half c ^ ((row >> 3) & 1) of that row, which makes the fragment reads
    // (ds_read_b128, 16-lane groups) conflict-free without padding.  Wave w fills rows 32w..32w+31 of all six pieces.
#define RX_ALOAD(kt_, buf_)                                                                                     \
    {                                                                                                           \
        const unsigned ao = aoff + (unsigned)((kt_) * BK) * 2u;                                                 \
        unsigned short *lb = lds + (buf_) * BUF + wave * 512;                                                   \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                         \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A0 + 2 * q * g.planeA + ao), \
                                             (__attribute__((address_space(3))) void *)(lb + q * PIECE), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A1 + 2 * q * g.planeA + ao), \
                                             (__attribute__((address_space(3))) void *)(lb + IMG + q * PIECE), 16, 0, 0); \
        }                                                                                                       \
    }
#define RX_MFMA(buf_)                                                                                           \
    {                                                                                                           \
        const unsigned short *ra = lds + (buf_) * BUF + l31 * RS + 8 * (h ^ ((l31 >> 3) & 1));                  \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                      \
            const unsigned short *p = ra + mt * 32 * RS;                                                        \
            const bf16x8 a0h = *reinterpret_cast<const bf16x8 *>(p);                                            \
            const bf16x8 a0m = *reinterpret_cast<const bf16x8 *>(p + PIECE);                                    \
            const bf16x8 a0l = *reinterpret_cast<const bf16x8 *>(p + 2 * PIECE);                                \
            MFMA6(acc1[mt], a0h, a0m, a0l, b0h, b0m, b0l)                                                       \
            const bf16x8 a1h = *reinterpret_cast<const bf16x8 *>(p + IMG);                                      \
            const bf16x8 a1m = *reinterpret_cast<const bf16x8 *>(p + IMG + PIECE);                              \
            const bf16x8 a1l = *reinterpret_cast<const bf16x8 *>(p + IMG + 2 * PIECE);                          \
            MFMA6(acc2[mt], a1h, a1m, a1l, b1h, b1m, b1l)                                                       \
        }                                                                                                       \
    }

    int hv = h;
    float xr[8], xi[8], qr[8], qi[8];
    float x0[8], x1[8];
    bf16x8 b0h, b0m, b0l, b1h, b1m, b1l;

    RX_SETUP(tile);
    if (MIX) RX_MIXTAB();
    RX_ALOAD(0, 0);
    RX_BLOAD(0);
    __syncthreads();                   // drains the DMA (vmcnt(0)) and publishes the mix table
    RX_BFOLD(0);
    split8(x0, b0h, b0m, b0l);
    split8(x1, b1h, b1m, b1l);
    int buf = 0;
    while (true) {
        asm volatile("" : "+v"(hv));   // keeps the per-lane fold selectors from being hoisted out of the tile loop (32 registers)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] = acc2[i][r] = 0.f;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            RX_ALOAD(kt + 1, buf ^ 1); // buf ^ 1 was last read before the previous barrier
            RX_BLOAD(kt + 1);          // the loads fly while the matrix cores work on tile kt
            RX_MFMA(buf);
            RX_BFOLD(kt + 1);
            split8(x0, b0h, b0m, b0l);
            split8(x1, b1h, b1m, b1l);
            __syncthreads();
            buf ^= 1;
        }
        // last k-step: where the next tile's first loads are issued
        // epilogue addressing: uniform base + 32-bit lane offset; the asm barrier keeps the (loop-invariant) row
        // offsets from being hoisted out of the persistent loop into a few hundred registers
        const int ealt = packed ? 0 : tile % g.nvar;
        // packed: a second-variant lane holds (acc1, acc2) = (A[0] X_second, A[1] X_first), i.e. that variant's products swapped
        const float e00 = var ? g.e_alt[1] : (ealt ? g.e_alt[0] : g.e00), e01 = var ? g.e_alt[0] : (ealt ? g.e_alt[1] : g.e01);
        const float e10 = var ? g.e_alt[3] : (ealt ? g.e_alt[2] : g.e10), e11 = var ? g.e_alt[2] : (ealt ? g.e_alt[3] : g.e11);
        char *D0 = reinterpret_cast<char *>((ealt ? g.dst_alt : g.dst[0]) + bz * g.sC + n0);
        const unsigned dvar = var ? (unsigned)((g.dst_alt - g.dst[0]) * 4) : 0u;      // second variant's array, as a lane offset
        char *D1 = reinterpret_cast<char *>((g.dst[1] ? g.dst[1] : g.dst[0]) + bz * g.sC + n0);
        const int em0 = m0;
        // row (em0 + rr + 4 h) through a running scalar pointer + the lane's (4 h, column) offset; mirror row
        // Rn - row = (Rn - em0 - 4 - rr) + 4 (1 - h) likewise
        char *Dk0 = D0 + (long)em0 * ldcB, *Dk1 = D1 + (long)em0 * ldcB, *Dm0 = D0 + (long)(g.Rn - em0 - 4) * ldcB;
        const unsigned lo = (unsigned)(4 * h) * ldc4 + c4 + dvar, lm = (unsigned)(4 * (1 - h)) * ldc4 + c4 + dvar;
        const int next = (g.strided || packed) ? tile + (int)gridDim.x
                                   : (((tile % g.nvar) + 1 < g.nvar) ? tile + 1 : tile + 1 + ((int)gridDim.x - 1) * g.nvar);
        const bool more = next < ntile;
        if (more) {
            RX_SETUP(next);
            RX_ALOAD(0, buf ^ 1);
            RX_BLOAD(0);
        }
        RX_MFMA(buf);
        {
            const long s1 = ldcB, s5 = 5 * ldcB;
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
        if (!more) break;
        if (MIX && (g.strided || packed || (next % g.nvar) == 0)) RX_MIXTAB();   // every wave finished reading the old table before the last barrier
        __syncthreads();
        RX_BFOLD(0);
        split8(x0, b0h, b0m, b0l);
        split8(x1, b1h, b1m, b1l);
        buf ^= 1;
        tile = next;
    }
#undef RX_SETUP
#undef RX_MIXTAB
#undef RX_BLOAD
#undef RX_BFOLD
#undef RX_ALOAD
#undef RX_MFMA
}

}  // namespace

int launch_dft_rx3(hipStream_t stream, const DftRx3Args &g) {
    if (g.MP % 128 || g.KP % BK || g.N % 128 || g.batch < 1) return (int)hipErrorInvalidValue;
    if (g.mode == 1 && !g.dst[1]) return (int)hipErrorInvalidValue;
    // the lane part of an address is 8 rows + a column inside the tile (32-bit); rows go through 64-bit scalar pointers
    if (8.0 * (double)g.ldb * 4.0 + 1024.0 >= 4294967296.0 || 4.0 * (double)g.ldc * 4.0 + 1024.0 >= 4294967296.0) return (int)hipErrorInvalidValue;
    const bool folded = g.fold[0] != 0.f || g.fold[1] != 0.f || (g.nvar == 2 && (g.fold_alt[0] != 0.f || g.fold_alt[1] != 0.f));
    const int kind = !folded ? 2 : (g.src[0] == g.src[1] ? 1 : 0);
    const int mix_rows = g.Kn > g.KP ? g.Kn : g.KP;
    const size_t mix_bytes = (size_t)mix_rows * 2 * sizeof(float4);
    if (g.mhat && (kind != 0 || g.LP % 128 || g.T < 1 || g.T > 4 || mix_rows > MIX_ROWS_MAX)) return (int)hipErrorInvalidValue;
    static int slots_of[64] = {0};     // two workgroups per CU (LDS and registers), persistent over the tiles
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (!slots_of[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return (int)hipErrorInvalidDevice;
        slots_of[dev] = 2 * cus;
    }
    const int slots = slots_of[dev];
    if (g.nvar != 1 && (g.nvar != 2 || g.mode != 0 || !g.dst_alt || !g.A_alt[0] || !g.A_alt[1])) return (int)hipErrorInvalidValue;
    const long ntile = (long)(g.N / 128) * (g.MP / 128) * g.batch;
    DftRx3Args a = g;
    const long dalt = (g.nvar == 2 && g.dst_alt) ? (long)(g.dst_alt - g.dst[0]) * 4 : -1;
    a.packed = (g.packed && g.nvar == 2 && kind == 0 && g.mode == 0 && g.N % 64 == 0 && dalt >= 0 && dalt < 2147483648L) ? 1 : 0;
    a.strided = (!a.packed && ntile < slots) ? 1 : 0;
    const long units = a.packed ? (long)(g.N / 64) * (g.MP / 128) * g.batch : (a.strided ? ntile * g.nvar : ntile);
    dim3 grid((unsigned)(units < slots ? units : slots));
    static unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    if (g.mhat) {
        if (int e = ensure_dynamic_lds(dft_rx3_kernel<0, true>, LDS_BYTES + (size_t)MIX_ROWS_MAX * 2 * sizeof(float4), d3)) return e;
        hipLaunchKernelGGL((dft_rx3_kernel<0, true>), grid, dim3(256), LDS_BYTES + mix_bytes, stream, a);
    } else if (kind == 0) {
        if (int e = ensure_dynamic_lds(dft_rx3_kernel<0, false>, LDS_BYTES, d0)) return e;
        hipLaunchKernelGGL((dft_rx3_kernel<0, false>), grid, dim3(256), LDS_BYTES, stream, a);
    } else if (kind == 1) {
        if (int e = ensure_dynamic_lds(dft_rx3_kernel<1, false>, LDS_BYTES, d1)) return e;
        hipLaunchKernelGGL((dft_rx3_kernel<1, false>), grid, dim3(256), LDS_BYTES, stream, a);
    } else {
        if (int e = ensure_dynamic_lds(dft_rx3_kernel<2, false>, LDS_BYTES, d2)) return e;
        hipLaunchKernelGGL((dft_rx3_kernel<2, false>), grid, dim3(256), LDS_BYTES, stream, a);
    }
    return (int)hipGetLastError();
}

// whether the kernel can run the 2-D transforms of an [NBP][NAP][LP] cube / [2][NAP][KBP][LP] spectrum (plan creation
// falls back to the fp32 folded kernels otherwise): eight row pitches must fit a 32-bit lane offset, the spectral-mix
// table must fit LDS
bool dft_rx3_supported(int Na, int Nb, long NAP, long KBP, long LP) {
    const double pitch = (double)(NAP > KBP ? NAP : KBP) * (double)LP * 4.0;      // largest row pitch of any pass (bytes)
    return 8.0 * pitch + 1024.0 < 4294967296.0 && (Na > Nb ? Na : Nb) <= MIX_ROWS_MAX;
}
