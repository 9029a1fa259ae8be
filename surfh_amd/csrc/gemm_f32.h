// fp32-input MFMA GEMM used by every dense contraction on the path:
//   R / R^T  (wblur_subSampling / wblur_t, surfh/ToolsDir/jax_utils.py:72-91)
//   the 2-D DFTs of the C stage written as real matrix products (jax_utils.py:30-41)
// C[b] (MxN, row-major) = A[b] (MxK, row-major) * B[b] (KxN, row-major), all fp32,
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct GemmArgs {
    // A operand: element (m,k) = (k < ksplitA ? A0[m*lda + k] : A1[m*lda + k - ksplitA])
    const float *A0 = nullptr, *A1 = nullptr;
    int ksplitA = 1 << 30;
    long lda = 0, sA = 0;   // leading dimension, batch stride (elements)
    // B operand: element (k,n) = (k < ksplitB ? B0[k*ldb + n] : B1[(k-ksplitB)*ldb + n])
    const float *B0 = nullptr, *B1 = nullptr;
    int ksplitB = 1 << 30;
    long ldb = 0, sB = 0;
    float *C = nullptr;
    long ldc = 0, sC = 0;
    int M = 0, N = 0, K = 0;       // multiples of 64 (M,N) and 16 (K)
    int batch = 1;
    int splitK = 1;                // >1: K is cut in splitK slabs, slab s is written to C + s*sCsplit
    long sCsplit = 0;
    int accumulate = 0;            // 1: C += A*B (only with splitK==1)
    // two-piece fp16 kernel (gemm_cc16.hip): both operands as fp16 pieces, piece q of element (m, k) at A3[q*pA3 + m*lda + k]
    // (A / scale of its row) and B16[q*pB16 + n*ldb + k] (B / sB16); the scales of A (one per row) are derived on the device
    // from amax[batch][M] = max |A[m][:]| as bit patterns
    const unsigned short *A3 = nullptr, *B16 = nullptr;
    long pA3 = 0, pB16 = 0;
    float sB16 = 0.f;
    const unsigned *amax = nullptr;
    // alternative to amax: A's pieces carry one scale per (row, K segment) -- bscale[segment][M] with
    // segment(k) = (k / segLinP) * segChunks + (k % segLinP) / 1024 (what launch_spmm_rows_f16 writes)
    const float *bscale = nullptr;
    int segLinP = 0, segChunks = 0;
    // optional K-step lists of the two-piece fp16 kernel, one record of klistStride ints per 256-row tile of B:
    // [n_near, n_far, near steps ascending ..., far steps ascending ...], entry = K step | segment << 16 (segment 0 without
    // bscale).  Near steps keep three products per element, far steps only the leading one (build_klist, plan.hip)
    const int *klist = nullptr;
    int klistStride = 0;
    // optional tile shape of the two-piece fp16 kernel along N for a B whose rows come in columns of permLin consecutive rows (N = ncol * permLin): a tile then takes 256 / permP consecutive rows of each of permP
    // neighbouring columns instead of 256 consecutive rows (the adjoint spectral-blur GEMM: rows of the same wavelengths share
    // their near K steps).  permP in {1, 2, 4, 8}, permLin % (256 / permP) == 0; 0: plain tiles
    int permP = 0, permLin = 0;
};

// returns hipError_t as int; name is used by the profiler
int launch_gemm_f32(hipStream_t stream, const GemmArgs &g);
// the same product with float64 accumulation (plain vector arithmetic; surfh_config.verify)
int launch_gemm_f64acc(hipStream_t stream, const GemmArgs &g);

// dst2[q*plane + i] = fp16 piece q (h, l) of src[i] / scale (round to nearest); scale from gemm_f16x2_scale(max |src|)
int launch_split2h(hipStream_t stream, const float *src, unsigned short *dst2, long n, long plane, float scale);
float gemm_f16x2_scale(float amax);
// C[M][N] = A[M][K] * B[N][K]^T as two-piece fp16 products (gemm_cc16.hip): 256 x 256 tile, eight consumer waves, both
// operands as fp16 pieces delivered by LDS-DMA.  M multiple of 64, N of 128, K of 32 * splitK.
// A3 / pA3 = pieces of A split row by row with the scales of amax[M] (launch_split_rows2h); B16 / pB16 / sB16 as above.
int launch_gemm_nt_f16x2_cc(hipStream_t stream, const GemmArgs &g);
// several such products in ONE launch (n <= 4): their tiles fill the rounds of workgroups together (the adjoint spectral-blur
// GEMMs of the bands are 1.5-1.8 rounds each on their own)
constexpr int GEMM_GROUP_MAX = 4;
int launch_gemm_nt_f16x2_cc_group(hipStream_t stream, const GemmArgs *g, int n);
int launch_split_rows2h(hipStream_t stream, const float *src, const unsigned *rowmax, unsigned short *dst2, int rows, int ld, long plane);
