// Symmetry-folded 1-D DFT pass as an fp32 MFMA GEMM (gfx950).
//
// A length-N DFT along one axis of a wavelength-innermost array is a product with an N x N
// matrix.  cos is even and sin is odd under k -> N-k, so with the input rows folded into
// even/odd pairs and the output rows produced in (r, N-r) pairs only (N/2+1)^2 entries of
// the cos and of the sin matrix are needed: 4x fewer multiply-adds than the dense product
// (2x for the real <-> half-spectrum passes, which fold on one side only).
//
//   acc1 = A[0] * fold(src[0]),   acc2 = A[1] * fold(src[1])           (two K phases)
//   fold(s)[k] = s[k] + f * s[Kn-k]   (f = 0: plain rows; no partner for k = 0 and 2k = Kn)
//   PAIR : dst[0][r]    = e00*acc1 + e01*acc2,   dst[0][Rn-r] = e10*acc1 + e11*acc2
//   SPLIT: dst[0][r]    = e00*acc1,              dst[1][r]    = e11*acc2
//
// Rows are vectors over the contiguous (.., lambda) axis: N columns, leading dimension ldb / ldc.
#pragma once
#include <hip/hip_runtime.h>

struct DftFoldArgs {
    const float *A[2] = {nullptr, nullptr};   // [MP][KP] row-major, zero padded
    int lda = 0;
    const float *src[2] = {nullptr, nullptr};
    long ldb = 0, sB = 0;                      // row stride, batch stride
    float fold[2] = {0.f, 0.f};
    int Kn = 0;                                // input-side transform length
    float *dst[2] = {nullptr, nullptr};
    long ldc = 0, sC = 0;
    int mode = 0;                              // 0 PAIR, 1 SPLIT
    float e00 = 1.f, e01 = 0.f, e10 = 0.f, e11 = 1.f;
    int Rn = 0;                                // output-side transform length (PAIR)
    int rvalid = 0;                            // valid output rows (N/2+1)
    int MP = 0, KP = 0;                        // padded matrix dims (multiples of 128 / 16)
    int N = 0;                                 // columns (multiple of 128)
    int batch = 1;
};

int launch_dft_fold(hipStream_t stream, const DftFoldArgs &g);

// Complex-to-complex pass with all four folded products in one workgroup (the spectrum is read once):
//   P1 = Cm*fold+(Xr)  P2 = Sm*fold-(Xi)  P3 = Sm*fold-(Xr)  P4 = Cm*fold+(Xi)
//   dst_r[r] = P1 - sgn*P2   dst_r[Rn-r] = P1 + sgn*P2   dst_i[r] = P4 + sgn*P3   dst_i[Rn-r] = P4 - sgn*P3
// sgn = +1: inverse transform (e^{+i}), sgn = -1: forward transform (e^{-i}).
struct DftFold4Args {
    const float *Cm = nullptr, *Sm = nullptr;   // [MP][KP]
    int lda = 0;
    const float *src_r = nullptr, *src_i = nullptr;
    long ldb = 0, sB = 0;
    float *dst_r = nullptr, *dst_i = nullptr;
    long ldc = 0, sC = 0;
    float sgn = 1.f;
    int Nn = 0;          // transform length (same on both sides)
    int rvalid = 0;      // Nn/2 + 1
    int MP = 0, KP = 0, N = 0, batch = 1;
    // optional fused spectral mix on the source (forward model only): src = H * sum_t tpl[t][l] * mhat[t][k][kb]
    // with the column index n = kb * LP + l  (mhat == nullptr: plain source)
    const float *mhat = nullptr, *tpl = nullptr;
    int T = 0, LP = 0;
    long PL = 0, KBP = 0;
};
int launch_dft_fold4(hipStream_t stream, const DftFold4Args &g);
