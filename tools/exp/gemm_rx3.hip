// Spectral-blur GEMM on the bf16 matrix cores, fp32-accurate by exact three-way splitting (arithmetic: gemm_bf16x3.hip),
// with a pre-split constant operand and a register-direct data operand.
//
//   Ct[n][m] = sum_k A[m][k] * B[n][k]          A: constant (spectral PSF W or W^T), handed over as three bf16 planes
//                                                B: data (gathered slit spectra / detector samples), fp32, K-contiguous
//
// One workgroup = 4 waves = a 128 (m) x 256 (n) tile, K step 32.  Each wave owns all 128 rows of 64 columns
// (2 column groups x 4 row tiles: 128 accumulator registers).
//  * B never touches LDS.  The B fragment of v_mfma_f32_32x32x16_bf16 is "8 consecutive k of one column per lane": a
//    lane reads 16 consecutive k of its own row n (64 contiguous bytes, four dwordx4) for the two MFMA k-steps of a
//    K step, and cuts them into bf16 pieces in registers.  The k order inside a K step is permuted (lane half h takes
//    k = 16h .. 16h+15) -- the A side reads its fragments with the same permutation, so the sum is unchanged.
//    Every B element is loaded and split by exactly one lane of the whole grid column.
//  * A tiles (3 pieces x 128 rows x 32 k bf16 = 24 KB) go global -> LDS by DMA (global_load_lds_dwordx4, no staging
//    registers, no ds_write) into an XOR-swizzled, unpadded image: position 4*row + c holds the 8-k chunk
//    c ^ ((row >> 2) & 3) of that row, which makes every ds_read_b128 fragment read conflict-free.  Double-buffered,
//    one barrier per K step of 96 MFMAs per wave.
//  * The result is stored transposed (Ct[n][m], four consecutive m per lane = one 16-byte store): both spectral-blur
//    products of the model want exactly that layout (y^T[(p,s,a)][l'] and Xs^T[(p,s,a)][k]).
//  * blockIdx.x -> (K slab, tile) with the slab index fastest: workgroups of one XCD (blockIdx.x mod 8) share a K
//    slab, so the slab of both operands stays in that XCD's L2 while its tiles walk it in step.
#include "gemm_rx3.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BM = 128, BN = 256, BK = 32;
constexpr int PIECE = BM * BK;                // one bf16 piece of the A tile (8 KB)
constexpr int BUF = 3 * PIECE;
constexpr size_t LDS_BYTES = (size_t)2 * BUF * sizeof(unsigned short);

__device__ __forceinline__ unsigned pack2(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// exact split of 8 values into three bf16x8 fragments (h, m, l)
__device__ __forceinline__ void split8(const float *x, bf16x8 &fh, bf16x8 &fm, bf16x8 &fl) {
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

__global__ __launch_bounds__(256, 2) void gemm_rx3_kernel(GemmRx3Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int tilesM = g.M / BM;
    const int sk = blockIdx.x % g.splitK, tile = blockIdx.x / g.splitK;
    const int m0 = (tile % tilesM) * BM, n0 = (tile / tilesM) * BN;
    const int Kper = g.K / g.splitK, kbeg = sk * Kper, nk = Kper / BK;
    const bool active = n0 + 64 * wave < g.N;           // ragged last column tile (N is a multiple of 128, not of 256)

    // B: uniform base + 32-bit lane offset
    const char *Bb = reinterpret_cast<const char *>(g.B + (long)n0 * g.ldb + kbeg);
    const unsigned bo0 = (unsigned)((64 * wave + l31) * g.ldb + 16 * h) * 4u;
    const unsigned bo1 = bo0 + (unsigned)(32 * g.ldb) * 4u;
    // A: DMA instruction i of this wave = piece (6w+i)>>3, rows 16*((6w+i)&7) .. +15; lane -> row (lane>>2), stored chunk
    // lane&3 holds chunk (lane&3) ^ ((row>>2)&3) = (lane&3) ^ ((lane>>4)&3)
    const unsigned ao = (unsigned)((lane >> 2) * g.lda + 8 * ((lane & 3) ^ ((lane >> 4) & 3))) * 2u;
    const char *Ab = reinterpret_cast<const char *>(g.A3 + (long)m0 * g.lda + kbeg);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define GX_ALOAD(kt_, buf_)                                                                                          \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                              \
            const int q = 6 * wave + i, pc = q >> 3, rb = q & 7;                                                     \
            const char *src = Ab + ((long)pc * g.planeA + (long)(16 * rb) * g.lda + (kt_) * BK) * 2;                 \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + ao),             \
                                             (__attribute__((address_space(3))) void *)(lds + (buf_) * BUF + pc * PIECE + rb * 512), \
                                             16, 0, 0);                                                              \
        }                                                                                                            \
    }
#define GX_BLOAD(kt_)                                                                                                \
    {                                                                                                                \
        const unsigned ko = (unsigned)((kt_) * BK) * 4u;                                                             \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                              \
            const float4 t0 = *reinterpret_cast<const float4 *>(Bb + (bo0 + ko + 16u * v));                          \
            const float4 t1 = *reinterpret_cast<const float4 *>(Bb + (bo1 + ko + 16u * v));                          \
            raw0[4 * v] = t0.x; raw0[4 * v + 1] = t0.y; raw0[4 * v + 2] = t0.z; raw0[4 * v + 3] = t0.w;              \
            raw1[4 * v] = t1.x; raw1[4 * v + 1] = t1.y; raw1[4 * v + 2] = t1.z; raw1[4 * v + 3] = t1.w;              \
        }                                                                                                            \
    }
#define GX_SPLIT()                                                                                                   \
    {                                                                                                                \
        split8(raw0, b00h, b00m, b00l);                                                                              \
        split8(raw0 + 8, b01h, b01m, b01l);                                                                          \
        split8(raw1, b10h, b10m, b10l);                                                                              \
        split8(raw1 + 8, b11h, b11m, b11l);                                                                          \
    }

    float raw0[16], raw1[16];
    bf16x8 b00h, b00m, b00l, b01h, b01m, b01l, b10h, b10m, b10l, b11h, b11m, b11l;   // [column group][k-step][piece]

    GX_ALOAD(0, 0);
    if (active) {
        GX_BLOAD(0);
        GX_SPLIT();
    }
    __syncthreads();                   // drains the DMA (vmcnt(0))
    int buf = 0;
    const int sw = (l31 >> 2) & 3;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            GX_ALOAD(kt + 1, buf ^ 1);     // buf ^ 1 was last read before the previous barrier
            if (active) GX_BLOAD(kt + 1);
        }
        if (active) {
            const unsigned short *ra = lds + buf * BUF + l31 * BK;
            // fragment reads run one (k-step, row tile) ahead of the MFMAs that consume them
            bf16x8 ah, am, al, nh, nm, nl;
            {
                const unsigned short *p = ra + 8 * ((2 * h) ^ sw);
                ah = *reinterpret_cast<const bf16x8 *>(p);
                am = *reinterpret_cast<const bf16x8 *>(p + PIECE);
                al = *reinterpret_cast<const bf16x8 *>(p + 2 * PIECE);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int s = i >> 2, mt = i & 3;
                if (i < 7) {
                    const int s2 = (i + 1) >> 2, mt2 = (i + 1) & 3;
                    const unsigned short *p = ra + mt2 * 32 * BK + 8 * ((2 * h + s2) ^ sw);
                    nh = *reinterpret_cast<const bf16x8 *>(p);
                    nm = *reinterpret_cast<const bf16x8 *>(p + PIECE);
                    nl = *reinterpret_cast<const bf16x8 *>(p + 2 * PIECE);
                }
                __builtin_amdgcn_sched_barrier(0);     // keep the reads ahead of this group's MFMAs
                if (s == 0) {
                    MFMA6(acc[mt][0], ah, am, al, b00h, b00m, b00l)
                    MFMA6(acc[mt][1], ah, am, al, b10h, b10m, b10l)
                } else {
                    MFMA6(acc[mt][0], ah, am, al, b01h, b01m, b01l)
                    MFMA6(acc[mt][1], ah, am, al, b11h, b11m, b11l)
                }
                __builtin_amdgcn_sched_barrier(0);
                ah = nh; am = nm; al = nl;
            }
            if (kt + 1 < nk) GX_SPLIT();
        }
        __syncthreads();
        buf ^= 1;
    }
#undef GX_ALOAD
#undef GX_BLOAD
#undef GX_SPLIT

    if (!active) return;
    float *Cb = g.Ct + (long)sk * g.sCsplit;
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) {
        float *crow = Cb + (long)(n0 + 64 * wave + 32 * cg + l31) * g.ldct + m0 + 4 * h;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4 *>(crow + 32 * mt + 8 * q) =
                    make_float4(acc[mt][cg][4 * q], acc[mt][cg][4 * q + 1], acc[mt][cg][4 * q + 2], acc[mt][cg][4 * q + 3]);
    }
}

}  // namespace

int launch_gemm_rx3(hipStream_t stream, const GemmRx3Args &g) {
    if (g.M % BM || g.N % 128 || g.splitK < 1 || g.K % (BK * g.splitK) || g.ldct % 4 || g.lda % 8 || g.ldb % 4)
        return (int)hipErrorInvalidValue;
    if ((double)(BN + 1) * (double)g.ldb * 4.0 >= 2147483648.0 || (double)16 * g.lda * 2.0 >= 2147483648.0) return (int)hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)gemm_rx3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long tiles = (long)(g.M / BM) * ((g.N + BN - 1) / BN);
    dim3 grid((unsigned)(tiles * g.splitK));
    hipLaunchKernelGGL(gemm_rx3_kernel, grid, dim3(256), LDS_BYTES, stream, g);
    return (int)hipGetLastError();
}
