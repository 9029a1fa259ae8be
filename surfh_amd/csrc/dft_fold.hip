// Symmetry-folded DFT pass on fp32 MFMA (see dft_fold.h).  Same 4-wave 128x128x16 tiling and
// v_mfma_f32_32x32x2_f32 operand maps as gemm_f32.hip; differences: the B tile loader folds a
// row with its mirror row on the fly, the K loop runs two phases into two accumulator sets, and
// the epilogue writes the (r, N-r) output pair from their sum / difference.
#include "dft_fold.h"
#include "lds_attr.h"

#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 128, BK = 16, ASTR = 20;

// ---- two products in ONE K loop (the real <-> half-spectrum passes) ------------------------------------------
// acc1 = A[0]*B1, acc2 = A[1]*B2 with (B1, B2) = (s[k] + s[Kn-k], s[k] - s[Kn-k]) of one source (the mirror row is
// read once for both), or (src[0][k], src[1][k]) of two sources.  PAIR / SPLIT epilogues of dft_fold.h.
constexpr int D2_BUF = 2 * BM * ASTR + 2 * BK * BN;     // floats per LDS buffer: A0, A1, B1, B2

__global__ __launch_bounds__(256, 2) void dft_dual_kernel(DftFoldArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int ar = tid >> 2, ac = (tid & 3) * 4;
    const int br = tid >> 5, bc = (tid & 31) * 4;
    const long b = blockIdx.z;
    const bool eo = g.fold[0] != 0.f;
    const float *S0 = g.src[0] + b * g.sB;
    const float *S1 = eo ? S0 : g.src[1] + b * g.sB;
    const int nk = g.KP / BK;
    const int kin = g.Kn / 2 + 1;

    f32x16 acc1[2][2], acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][j][r] = acc2[i][j][r] = 0.f;

    float4 a00, a01, a10, a11, b10, b11, b20, b21;
#define D2_ROW(k_, v1_, v2_)                                                                                 \
    {                                                                                                        \
        v1_ = *reinterpret_cast<const float4 *>(S0 + (long)(k_) * g.ldb + n0 + bc);                          \
        if (eo) {                                                                                            \
            v2_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                           \
            if (((k_) >= 1) && ((k_) < kin) && (2 * (k_) != g.Kn)) {                                         \
                const float4 q = *reinterpret_cast<const float4 *>(S0 + (long)(g.Kn - (k_)) * g.ldb + n0 + bc); \
                v2_ = make_float4(v1_.x - q.x, v1_.y - q.y, v1_.z - q.z, v1_.w - q.w);                       \
                v1_.x += q.x; v1_.y += q.y; v1_.z += q.z; v1_.w += q.w;                                      \
            }                                                                                                \
        } else {                                                                                             \
            v2_ = *reinterpret_cast<const float4 *>(S1 + (long)(k_) * g.ldb + n0 + bc);                      \
        }                                                                                                    \
    }
#define D2_GLOAD(kt_)                                                                                        \
    {                                                                                                        \
        const int k0 = (kt_) * BK;                                                                           \
        a00 = *reinterpret_cast<const float4 *>(g.A[0] + (long)(m0 + ar) * g.lda + k0 + ac);                 \
        a01 = *reinterpret_cast<const float4 *>(g.A[0] + (long)(m0 + ar + 64) * g.lda + k0 + ac);            \
        a10 = *reinterpret_cast<const float4 *>(g.A[1] + (long)(m0 + ar) * g.lda + k0 + ac);                 \
        a11 = *reinterpret_cast<const float4 *>(g.A[1] + (long)(m0 + ar + 64) * g.lda + k0 + ac);            \
        D2_ROW(k0 + br, b10, b20);                                                                           \
        D2_ROW(k0 + br + 8, b11, b21);                                                                       \
    }
#define D2_LSTORE(buf_)                                                                                      \
    {                                                                                                        \
        float *base = lds + (buf_) * D2_BUF;                                                                 \
        *reinterpret_cast<float4 *>(base + ar * ASTR + ac) = a00;                                            \
        *reinterpret_cast<float4 *>(base + (ar + 64) * ASTR + ac) = a01;                                     \
        *reinterpret_cast<float4 *>(base + BM * ASTR + ar * ASTR + ac) = a10;                                \
        *reinterpret_cast<float4 *>(base + BM * ASTR + (ar + 64) * ASTR + ac) = a11;                         \
        float *bb = base + 2 * BM * ASTR;                                                                    \
        *reinterpret_cast<float4 *>(bb + br * BN + bc) = b10;                                                \
        *reinterpret_cast<float4 *>(bb + (br + 8) * BN + bc) = b11;                                          \
        *reinterpret_cast<float4 *>(bb + BK * BN + br * BN + bc) = b20;                                      \
        *reinterpret_cast<float4 *>(bb + BK * BN + (br + 8) * BN + bc) = b21;                                \
    }
    D2_GLOAD(0);
    D2_LSTORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        D2_GLOAD((kt + 1 < nk) ? kt + 1 : kt);
        const float *A0 = lds + buf * D2_BUF, *A1 = A0 + BM * ASTR, *B1 = A0 + 2 * BM * ASTR, *B2 = B1 + BK * BN;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float4 x0[2], x1[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                x0[mt] = *reinterpret_cast<const float4 *>(A0 + (wm * 64 + mt * 32 + l31) * ASTR + 8 * q + 4 * h);
                x1[mt] = *reinterpret_cast<const float4 *>(A1 + (wm * 64 + mt * 32 + l31) * ASTR + 8 * q + 4 * h);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float v1[2], v2[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int off = (8 * q + 4 * h + m) * BN + wn * 64 + nt * 32 + l31;
                    v1[nt] = B1[off];
                    v2[nt] = B2[off];
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float c0 = (m == 0) ? x0[mt].x : (m == 1) ? x0[mt].y : (m == 2) ? x0[mt].z : x0[mt].w;
                    const float c1 = (m == 0) ? x1[mt].x : (m == 1) ? x1[mt].y : (m == 2) ? x1[mt].z : x1[mt].w;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, v1[nt], acc1[mt][nt], 0, 0, 0);
                        acc2[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, v2[nt], acc2[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
        D2_LSTORE(buf ^ 1);
        __syncthreads();
    }
#undef D2_ROW
#undef D2_GLOAD
#undef D2_LSTORE

    float *d0 = g.dst[0] + b * g.sC;
    float *d1 = g.dst[1] ? g.dst[1] + b * g.sC : nullptr;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = n0 + wn * 64 + nt * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= g.rvalid) continue;
                const float a1 = acc1[mt][nt][r], a2 = acc2[mt][nt][r];
                if (g.mode == 0) {
                    d0[(long)row * g.ldc + col] = g.e00 * a1 + g.e01 * a2;
                    if (row >= 1 && 2 * row != g.Rn) d0[(long)(g.Rn - row) * g.ldc + col] = g.e10 * a1 + g.e11 * a2;
                } else {
                    d0[(long)row * g.ldc + col] = g.e00 * a1;
                    d1[(long)row * g.ldc + col] = g.e11 * a2;
                }
            }
        }
}

// ---- four-product variant for the complex-to-complex passes ------------------------------------------
constexpr int BN4 = 64;
constexpr int A4_FLOATS = BM * ASTR;        // one A tile
constexpr int B4_FLOATS = BK * BN4;         // one folded B tile
constexpr int BUF4_FLOATS = 2 * A4_FLOATS + 4 * B4_FLOATS;

__global__ __launch_bounds__(256, 2) void dft_fold4_kernel(DftFold4Args g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN4;
    const int ar = tid >> 2, ac = (tid & 3) * 4;       // A tiles: 128 rows x 16 k
    const int br = tid >> 4, bc = (tid & 15) * 4;      // B tiles: 16 k x 64 cols
    const long b = blockIdx.z;
    const float *Xr = g.src_r + b * g.sB, *Xi = g.src_i + b * g.sB;
    const int nk = g.KP / BK;
    const int kin = g.Nn / 2 + 1;

    // fused spectral mix: the column tile lies inside one k_beta (LP is a multiple of the tile width)
    const bool mix = g.mhat != nullptr;
    const int kb = mix ? n0 / g.LP : 0;
    const int l0 = mix ? n0 % g.LP : 0;

    f32x16 P1[2], P2[2], P3[2], P4[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) P1[i][r] = P2[i][r] = P3[i][r] = P4[i][r] = 0.f;

    float4 rc0, rc1, rs0, rs1, bre, bro, bie, bio;

    // X = H * S with S = sum_t tpl[t][l] * mhat[t][k][kb] when the spectral mix is fused, else the plain row
#define F4_LOAD_ROW(k_, vr_, vi_)                                                                           \
    {                                                                                                       \
        vr_ = *reinterpret_cast<const float4 *>(Xr + (long)(k_) * g.ldb + n0 + bc);                         \
        vi_ = *reinterpret_cast<const float4 *>(Xi + (long)(k_) * g.ldb + n0 + bc);                         \
        if (mix) {                                                                                          \
            float4 sr = make_float4(0.f, 0.f, 0.f, 0.f), si = sr;                                           \
            for (int t = 0; t < g.T; ++t) {                                                                 \
                const float4 w = *reinterpret_cast<const float4 *>(g.tpl + (long)t * g.LP + l0 + bc);       \
                const float mr = g.mhat[((long)t * 2 + 0) * g.PL + (long)(k_) * g.KBP + kb];                \
                const float mi = g.mhat[((long)t * 2 + 1) * g.PL + (long)(k_) * g.KBP + kb];                \
                sr.x += w.x * mr; sr.y += w.y * mr; sr.z += w.z * mr; sr.w += w.w * mr;                     \
                si.x += w.x * mi; si.y += w.y * mi; si.z += w.z * mi; si.w += w.w * mi;                     \
            }                                                                                               \
            const float4 hr = vr_, hi = vi_;                                                                \
            vr_.x = hr.x * sr.x - hi.x * si.x; vi_.x = hr.x * si.x + hi.x * sr.x;                           \
            vr_.y = hr.y * sr.y - hi.y * si.y; vi_.y = hr.y * si.y + hi.y * sr.y;                           \
            vr_.z = hr.z * sr.z - hi.z * si.z; vi_.z = hr.z * si.z + hi.z * sr.z;                           \
            vr_.w = hr.w * sr.w - hi.w * si.w; vi_.w = hr.w * si.w + hi.w * sr.w;                           \
        }                                                                                                   \
    }
#define F4_GLOAD(kt_)                                                                                       \
    {                                                                                                       \
        const int k0 = (kt_) * BK;                                                                          \
        rc0 = *reinterpret_cast<const float4 *>(g.Cm + (long)(m0 + ar) * g.lda + k0 + ac);                  \
        rc1 = *reinterpret_cast<const float4 *>(g.Cm + (long)(m0 + ar + 64) * g.lda + k0 + ac);             \
        rs0 = *reinterpret_cast<const float4 *>(g.Sm + (long)(m0 + ar) * g.lda + k0 + ac);                  \
        rs1 = *reinterpret_cast<const float4 *>(g.Sm + (long)(m0 + ar + 64) * g.lda + k0 + ac);             \
        const int k = k0 + br;                                                                              \
        float4 xr, xi;                                                                                      \
        F4_LOAD_ROW(k, xr, xi);                                                                             \
        bre = xr; bie = xi;                                                                                 \
        /* k = 0 (and the Nyquist row of an even length) has no partner: the odd folds vanish */           \
        bro = make_float4(0.f, 0.f, 0.f, 0.f); bio = bro;                                                   \
        if ((k >= 1) && (k < kin) && (2 * k != g.Nn)) {                                                     \
            float4 qr, qi;                                                                                  \
            F4_LOAD_ROW(g.Nn - k, qr, qi);                                                                  \
            bre.x += qr.x; bre.y += qr.y; bre.z += qr.z; bre.w += qr.w;                                     \
            bro.x = xr.x - qr.x; bro.y = xr.y - qr.y; bro.z = xr.z - qr.z; bro.w = xr.w - qr.w;             \
            bie.x += qi.x; bie.y += qi.y; bie.z += qi.z; bie.w += qi.w;                                     \
            bio.x = xi.x - qi.x; bio.y = xi.y - qi.y; bio.z = xi.z - qi.z; bio.w = xi.w - qi.w;             \
        }                                                                                                   \
    }
#define F4_LSTORE(buf_)                                                                                     \
    {                                                                                                       \
        float *base = lds + (buf_) * BUF4_FLOATS;                                                           \
        *reinterpret_cast<float4 *>(base + ar * ASTR + ac) = rc0;                                           \
        *reinterpret_cast<float4 *>(base + (ar + 64) * ASTR + ac) = rc1;                                    \
        *reinterpret_cast<float4 *>(base + A4_FLOATS + ar * ASTR + ac) = rs0;                               \
        *reinterpret_cast<float4 *>(base + A4_FLOATS + (ar + 64) * ASTR + ac) = rs1;                        \
        float *bb = base + 2 * A4_FLOATS + br * BN4 + bc;                                                   \
        *reinterpret_cast<float4 *>(bb) = bre;                                                              \
        *reinterpret_cast<float4 *>(bb + B4_FLOATS) = bro;                                                  \
        *reinterpret_cast<float4 *>(bb + 2 * B4_FLOATS) = bie;                                              \
        *reinterpret_cast<float4 *>(bb + 3 * B4_FLOATS) = bio;                                              \
    }

    F4_GLOAD(0);
    F4_LSTORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        F4_GLOAD((kt + 1 < nk) ? kt + 1 : kt);
        const float *Ac = lds + buf * BUF4_FLOATS, *As_ = Ac + A4_FLOATS, *Bb = Ac + 2 * A4_FLOATS;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float4 c4[2], s4[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                c4[mt] = *reinterpret_cast<const float4 *>(Ac + (wm * 64 + mt * 32 + l31) * ASTR + 8 * q + 4 * h);
                s4[mt] = *reinterpret_cast<const float4 *>(As_ + (wm * 64 + mt * 32 + l31) * ASTR + 8 * q + 4 * h);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int off = (8 * q + 4 * h + m) * BN4 + wn * 32 + l31;
                const float vre = Bb[off], vro = Bb[B4_FLOATS + off], vie = Bb[2 * B4_FLOATS + off], vio = Bb[3 * B4_FLOATS + off];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float cv = (m == 0) ? c4[mt].x : (m == 1) ? c4[mt].y : (m == 2) ? c4[mt].z : c4[mt].w;
                    const float sv = (m == 0) ? s4[mt].x : (m == 1) ? s4[mt].y : (m == 2) ? s4[mt].z : s4[mt].w;
                    P1[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cv, vre, P1[mt], 0, 0, 0);
                    P2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, vio, P2[mt], 0, 0, 0);
                    P3[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, vro, P3[mt], 0, 0, 0);
                    P4[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cv, vie, P4[mt], 0, 0, 0);
                }
            }
        }
        F4_LSTORE(buf ^ 1);
        __syncthreads();
    }
#undef F4_LOAD_ROW
#undef F4_GLOAD
#undef F4_LSTORE

    float *dr = g.dst_r + b * g.sC, *di = g.dst_i + b * g.sC;
    const int col = n0 + wn * 32 + l31;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= g.rvalid) continue;
            const float p1 = P1[mt][r], p2 = g.sgn * P2[mt][r], p3 = g.sgn * P3[mt][r], p4 = P4[mt][r];
            dr[(long)row * g.ldc + col] = p1 - p2;
            di[(long)row * g.ldc + col] = p4 + p3;
            if (row >= 1 && 2 * row != g.Nn) {
                dr[(long)(g.Nn - row) * g.ldc + col] = p1 + p2;
                di[(long)(g.Nn - row) * g.ldc + col] = p4 - p3;
            }
        }
}

}  // namespace

int launch_dft_fold4(hipStream_t stream, const DftFold4Args &g) {
    if (g.MP % BM || g.KP % BK || g.N % BN4 || g.batch < 1) return (int)hipErrorInvalidValue;
    if (g.mhat && (g.LP % BN4 || g.T < 1)) return (int)hipErrorInvalidValue;
    const size_t lds_bytes = (size_t)2 * BUF4_FLOATS * sizeof(float);
    static unsigned long long attr_done = 0;
    if (int e = ensure_dynamic_lds(dft_fold4_kernel, lds_bytes, attr_done)) return e;
    dim3 grid(g.N / BN4, g.MP / BM, g.batch);
    hipLaunchKernelGGL(dft_fold4_kernel, grid, dim3(256), lds_bytes, stream, g);
    return (int)hipGetLastError();
}

int launch_dft_fold(hipStream_t stream, const DftFoldArgs &g) {
    if (g.MP % BM || g.KP % BK || g.N % BN || g.batch < 1) return (int)hipErrorInvalidValue;
    if (g.mode == 1 && !g.dst[1]) return (int)hipErrorInvalidValue;
    dim3 grid(g.N / BN, g.MP / BM, g.batch);
    // the single-K-loop kernel covers both real passes: even/odd fold of one source, or two plain sources
    const bool eo = g.src[0] == g.src[1] && g.fold[0] == 1.f && g.fold[1] == -1.f;
    const bool two = g.fold[0] == 0.f && g.fold[1] == 0.f;
    if (!eo && !two) return (int)hipErrorInvalidValue;
    const size_t lds_bytes = (size_t)2 * D2_BUF * sizeof(float);
    static unsigned long long attr_done = 0;
    if (int e = ensure_dynamic_lds(dft_dual_kernel, lds_bytes, attr_done)) return e;
    hipLaunchKernelGGL(dft_dual_kernel, grid, dim3(256), lds_bytes, stream, g);
    return (int)hipGetLastError();
}
