// Symmetry-folded DFT pass on fp32 MFMA (see dft_fold.h).  Same 4-wave 128x128x16 tiling and
// v_mfma_f32_32x32x2_f32 operand maps as gemm_f32.hip; differences: the B tile loader folds a
// row with its mirror row on the fly, the K loop runs two phases into two accumulator sets, and
// the epilogue writes the (r, N-r) output pair from their sum / difference.
#include "dft_fold.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 128, BK = 16, ASTR = 20;

struct Tile {
    int m0, n0, ar, ac, br, bc, wm, wn, l31, h;
};

__device__ __forceinline__ void run_phase(const DftFoldArgs &g, int ph, const float *__restrict__ Ap,
                                          const float *__restrict__ Bp, const Tile &t, float (*As)[BM * ASTR],
                                          float (*Bs)[BK * BN], f32x16 (&acc)[2][2]) {
    const int nk = g.KP / BK;
    const float f = g.fold[ph];
    const int kin = g.Kn / 2 + 1;
    float4 ra[2], rb[2];

    auto gload = [&](int kt, float4 (&xa)[2], float4 (&xb)[2]) {
        const int k0 = kt * BK;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            xa[u] = *reinterpret_cast<const float4 *>(Ap + (long)(t.m0 + t.ar + 64 * u) * g.lda + k0 + t.ac);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = k0 + t.br + 8 * u;
            float4 v = *reinterpret_cast<const float4 *>(Bp + (long)k * g.ldb + t.n0 + t.bc);
            const bool pair = (f != 0.f) && (k >= 1) && (k < kin) && (2 * k != g.Kn);
            if (pair) {
                const float4 q = *reinterpret_cast<const float4 *>(Bp + (long)(g.Kn - k) * g.ldb + t.n0 + t.bc);
                v.x += f * q.x; v.y += f * q.y; v.z += f * q.z; v.w += f * q.w;
            }
            xb[u] = v;
        }
    };

    gload(0, ra, rb);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        *reinterpret_cast<float4 *>(&As[0][(t.ar + 64 * u) * ASTR + t.ac]) = ra[u];
        *reinterpret_cast<float4 *>(&Bs[0][(t.br + 8 * u) * BN + t.bc]) = rb[u];
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        gload((kt + 1 < nk) ? kt + 1 : kt, ra, rb);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float4 a[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                a[mt] = *reinterpret_cast<const float4 *>(&As[buf][(t.wm * 64 + mt * 32 + t.l31) * ASTR + 8 * q + 4 * t.h]);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float bv[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bv[nt] = Bs[buf][(8 * q + 4 * t.h + m) * BN + t.wn * 64 + nt * 32 + t.l31];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float av = (m == 0) ? a[mt].x : (m == 1) ? a[mt].y : (m == 2) ? a[mt].z : a[mt].w;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<float4 *>(&As[buf ^ 1][(t.ar + 64 * u) * ASTR + t.ac]) = ra[u];
            *reinterpret_cast<float4 *>(&Bs[buf ^ 1][(t.br + 8 * u) * BN + t.bc]) = rb[u];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void dft_fold_kernel(DftFoldArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][BM * ASTR];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Tile t;
    t.wm = wave >> 1; t.wn = wave & 1; t.l31 = lane & 31; t.h = lane >> 5;
    t.m0 = blockIdx.y * BM; t.n0 = blockIdx.x * BN;
    t.ar = tid >> 2; t.ac = (tid & 3) * 4; t.br = tid >> 5; t.bc = (tid & 31) * 4;
    const long b = blockIdx.z;

    f32x16 acc1[2][2], acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][j][r] = acc2[i][j][r] = 0.f;

    run_phase(g, 0, g.A[0], g.src[0] + b * g.sB, t, As, Bs, acc1);
    run_phase(g, 1, g.A[1], g.src[1] + b * g.sB, t, As, Bs, acc2);

    float *d0 = g.dst[0] + b * g.sC;
    float *d1 = g.dst[1] ? g.dst[1] + b * g.sC : nullptr;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = t.n0 + t.wn * 64 + nt * 32 + t.l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = t.m0 + t.wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * t.h;
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

}  // namespace

int launch_dft_fold(hipStream_t stream, const DftFoldArgs &g) {
    if (g.MP % BM || g.KP % BK || g.N % BN || g.batch < 1) return (int)hipErrorInvalidValue;
    if (g.mode == 1 && !g.dst[1]) return (int)hipErrorInvalidValue;
    dim3 grid(g.N / BN, g.MP / BM, g.batch);
    hipLaunchKernelGGL(dft_fold_kernel, grid, dim3(256), 0, stream, g);
    return (int)hipGetLastError();
}
