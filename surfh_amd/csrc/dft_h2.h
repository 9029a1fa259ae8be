// Symmetry-folded 1-D DFT pass on the fp16 matrix cores: two round-to-nearest fp16 pieces per operand, three
// products (arithmetic: gemm_cc16.hip), the cos / sin matrices RESIDENT in LDS for the whole launch, complex arrays
// INTERLEAVED (re, im adjacent).
//
//   acc1 = A0 * X0,  acc2 = A1 * X1            one K loop, two products per output column
//   kind 0 (complex -> complex, folded along the transformed axis, both components):
//        a lane owns one component c of one column: X0 = even part of component c, X1 = odd part of component 1 - c
//        (E[k] = x[k] + x[Kn-k], O[k] = x[k] - x[Kn-k]; no mirror for k = 0 and 2k = Kn, where O = 0)
//        out_c[r] = e0 acc1 + e1 acc2,  out_c[Rn-r] = e2 acc1 + e3 acc2      (e for c = 0, e_alt for c = 1)
//        optionally with the spectral mix fused into the loader (forward model):
//        x = src * sum_t tpl[t][l] * mhat[t][k][kb]            (column n = kb*LP + l)
//   kind 1 (real -> complex, folded):   X0 = E, X1 = O of the real source;  out[r] = (e0 acc1, e3 acc2)
//   kind 2 (complex -> real, plain rows): X0 = re, X1 = im;  out[r] = e0 acc1 + e1 acc2, out[Rn-r] = e2 acc1 + e3 acc2
//
// How the operands reach the matrix cores (against the round-1 split-bf16 kernel, tools/exp/dft_rx3.h):
//   * matrices: both 128 x 128 folded matrices of a pass, cut into (hi, lo) fp16 pieces at a common power-of-two
//     scale 2^kA, are one 128 KB image (`dft_h2_build_image`) that every workgroup copies into LDS once.  No matrix
//     traffic and NO BARRIER inside the k loop: the eight waves of a workgroup run decoupled, so one wave's loads,
//     fold / split arithmetic and stores overlap its neighbours' MFMAs (the split-bf16 kernel streams a 24 KB matrix
//     tile per k-step behind a workgroup barrier, which keeps all waves in the same phase and waits for the data
//     loads of the next k-step at every barrier).
//   * data: register-direct (a lane loads the 8 k of its own column: consecutive lanes = consecutive floats, every
//     wave instruction moves whole 128-byte segments; with interleaved complex arrays that holds for the 16-column
//     complex tiles too), folded, then cut into two fp16 pieces under a per-column running block exponent: the scale
//     of a column is set by the first k-step of a tile and lowered (accumulators rescaled by the exact power of two)
//     only if a later k-step would overflow; it never rises, so small late values keep an absolute error of 2^-35 of
//     the column's largest value.
// Needs 16 < N/2+1 <= 128 on the axis (the image must fit LDS) and row offsets below 4 GB; longer axes run dft_ct.h (one
// Cooley-Tukey step around the same loop), an axis neither covers puts the plan on dense fp32 products (planar arrays).
#pragma once
#include <hip/hip_runtime.h>

constexpr int DFT_H2_KT = 8;                                   // k-steps of 16 in the image (K padded to 128)
constexpr size_t DFT_H2_IMAGE_HALFS = (size_t)2 * 2 * DFT_H2_KT * 128 * 16;   // [matrix][piece][kt][row][16] = 128 KB

struct DftH2Args {
    int kind = 0;
    const float *src = nullptr;                // kind 0 / 2: interleaved complex; kind 1: real
    long ldb = 0, sB = 0;                      // row pitch / batch stride of src, in floats
    int Kn = 0;                                // transform length of the folded kinds (rows k and Kn - k)
    float *dst = nullptr;                      // kind 0 / 1: interleaved complex; kind 2: real
    long ldc = 0, sC = 0;                      // row pitch / batch stride of dst, in floats
    float e[4] = {1.f, 0.f, 0.f, 1.f};
    float e_alt[4] = {1.f, 0.f, 0.f, 1.f};
    int Rn = 0, rvalid = 0;                    // output rows r < rvalid (and their mirrors Rn - r where they exist)
    int KP = 0;                                // K padded to 16 (columns of the image in use)
    int N = 0, batch = 1;                      // columns per batch entry (complex columns for kind 0; % 128 == 0)
    const float *mhat = nullptr, *tpl = nullptr;   // optional fused spectral mix (kind 0)
    int T = 0, LP = 0;
    long PL = 0, KBP = 0;
    int mix_l0 = 0;                            // with batch > 1 (a wavelength chunk, batched over kb): first plane of the chunk
    // mhat given in Parseval-scaled form (the solver's vectors, see surfh_normal_spec_dev): column kb is multiplied by
    // mhat_self (kb = 0 or 2 kb = mix_Nb: the bin is its own conjugate) or mhat_pair (all others) when the table is built
    float mhat_self = 1.f, mhat_pair = 1.f;
    int mix_Nb = 0;
    // optional list of the super-tiles (8 wave tiles = 128 columns, numbered batch-major) to transform, ascending; the others
    // are left alone (the forward model's complex pass: the (k_beta, wavelength chunk) pairs outside the OTF's support)
    const int *vlist = nullptr;
    int nvalid = 0;
    // optional per-tile limits, by the tile's chunk of 128 columns, chunk = (first column of the tile % tabLP) / 128:
    // ktab[chunk] = k-steps of 16 to run (>= 2; the source rows beyond are zero or immaterial), rtab[chunk] = output rows to store
    const int *ktab = nullptr, *rtab = nullptr;
    int tabLP = 0;
};

// Fused tail of the adjoint: the kind-0 pass of rfft2 followed, inside the kernel, by
//   madj[t][c][ka][kb] = sum_l tpl[t][l] (conj(H[ka][kb][l]) Y[ka][kb][l])[c]
// (g: the kind-0 arguments with batch = k_beta, N = LP; g.dst / ldc / sC unused).  H interleaved complex like Y.
struct DftH2AdjMix {
    const float *hsrc = nullptr;               // sotf: row ka at pitch ldh floats, k_beta at stride sH floats, then [l][2]
    long ldh = 0, sH = 0;
    const float *tpl = nullptr;                // [T][LPt]
    int T = 0, LPt = 0;
    float *mpart = nullptr;                    // work buffer of dft_h2_adjmix_part_floats(LP, hb) floats
    int nslot = 0;                             // set by the launcher
    int kt0 = 0;                               // the source rows k and Kn - k are zero for k < 16 kt0: those k-steps are skipped
    // output in the solver's Parseval-scaled form, optionally with the quadratic prior added (surfh_normal_spec_dev):
    // madj[..][ka][kb] = out_self | out_pair (by kb, as above) * sum + prior_mu * (4 - 2 cos(2 pi ka / Na) - 2 cos(2 pi kb / Nb)) * prior_src[..]
    float out_self = 1.f, out_pair = 1.f;
    int Nb = 0;
    const float *prior_src = nullptr;
    float prior_mu = 0.f;
    // optional: only the super-tiles (8 tiles = 128 wavelengths of one k_beta; index k_beta * (LP / 128) + chunk) of `vlist`
    // (ascending, nvalid entries) are transformed -- the others lie outside the OTF's support and add nothing;
    // kbstart[kb] = position in vlist of the first super-tile of kb (hb + 1 entries)
    const int *vlist = nullptr, *kbstart = nullptr;
    int nvalid = 0;
};
size_t dft_h2_adjmix_part_floats(long LP, int hb, long nvalid = 0);
int launch_dft_h2_adjmix(hipStream_t stream, const DftH2Args &g, const DftH2AdjMix &am, float *madj, long PL, long KBP,
                         const unsigned short *img, int kA);

// host: builds the LDS image of the two row-major [MP][KP] fp32 matrices (MP <= 128, KP <= 128, KP % 16 == 0);
// returns the scale exponent kA (pieces hold A * 2^kA)
int dft_h2_build_image(const float *A0, const float *A1, int MP, int KP, int lda, unsigned short *img);
int launch_dft_h2(hipStream_t stream, const DftH2Args &g, const unsigned short *img, int kA);
bool dft_h2_supported(int Na, int Nb, long NAP, long KBP, long LP);
