// Symmetry-folded 1-D DFT pass on the bf16 matrix cores with exact 3-way operand splitting, register-direct data operand.
// Arithmetic: every fp32 operand is cut into three bf16 pieces by truncating successive remainders (24 mantissa bits = 3 x 8,
// so x = h + m + l exactly); six of the nine partial products are kept (hh, hm, mh, hl, lh, mm; the dropped ones are below
// 2^-23 relative, the size of the fp32 product rounding) and accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
//
//   acc1 = A[0] * B1,  acc2 = A[1] * B2        one K loop, two products
//   B_s[k] = X_s[k] + f_s * X_s[Kn-k]          f_s = +1 / -1 / 0 (0: plain rows; no mirror for k = 0 and 2k = Kn,
//                                              where an odd fold (f = -1) is zero)
//   X_s = src[s]                               or, with the spectral mix fused (forward model, complex pass):
//   X_0 + i X_1 = (src[0] + i src[1]) * sum_t tpl[t][l] * mhat[t][k][kb]        (column n = kb*LP + l)
//   PAIR : dst[0][r] = e00*acc1 + e01*acc2,  dst[0][Rn-r] = e10*acc1 + e11*acc2
//   SPLIT: dst[0][r] = e00*acc1,             dst[1][r]    = e11*acc2
//
// The MFMA B fragment of v_mfma_f32_32x32x16_bf16 is "8 consecutive k of one column per lane": a lane loads
// exactly those 8 values of its own column from global memory (consecutive lanes = consecutive columns, so
// every load instruction is two 128-byte segments), folds and splits them in registers and feeds the matrix
// core directly -- the data operand never touches LDS and needs no transposition.  Only the (tiny, pre-split)
// cos / sin matrix tiles go through LDS.
#pragma once
#include <hip/hip_runtime.h>

struct DftRx3Args {
    const unsigned short *A[2] = {nullptr, nullptr};   // each: three bf16 planes [MP][KP], plane stride planeA
    long planeA = 0;
    int lda = 0;
    const float *src[2] = {nullptr, nullptr};
    long ldb = 0, sB = 0;
    float fold[2] = {0.f, 0.f};
    int Kn = 0;
    float *dst[2] = {nullptr, nullptr};
    long ldc = 0, sC = 0;
    int mode = 0;                              // 0 PAIR, 1 SPLIT
    float e00 = 1.f, e01 = 0.f, e10 = 0.f, e11 = 1.f;
    int Rn = 0, rvalid = 0;
    int MP = 0, KP = 0, N = 0, batch = 1;
    // optional second variant on the same data (nvar = 2): each tile is processed twice back to back -- the two
    // output components of a complex pass -- so that the second read of the tile comes from the caches, not HBM
    int nvar = 1;
    int strided = 0;                           // set by the launcher
    // packed = 1 (with nvar = 2): both variants in ONE pass over 64-column tiles -- lanes 0-15 of a half-wave carry
    // the columns with the first variant's fold, lanes 16-31 the same columns with the second variant's, so the tile is
    // read once; needs 0 <= (dst_alt - dst[0]) * 4 < 2^31 (the launcher falls back to the two-pass form otherwise)
    int packed = 0;
    const unsigned short *A_alt[2] = {nullptr, nullptr};
    float fold_alt[2] = {0.f, 0.f};
    float *dst_alt = nullptr;                  // PAIR only
    float e_alt[4] = {1.f, 0.f, 0.f, 1.f};     // e00, e01, e10, e11
    const float *mhat = nullptr, *tpl = nullptr;   // optional fused spectral mix
    int T = 0, LP = 0;
    long PL = 0, KBP = 0;
};
int launch_dft_rx3(hipStream_t stream, const DftRx3Args &g);
bool dft_rx3_supported(int Na, int Nb, long NAP, long KBP, long LP);
