// 1-D DFT pass of length N = R * M (R = 2, 3 or 4; M <= 190) on the fp16 matrix cores: one Cooley-Tukey step around the
// LDS-resident folded M-point transform of dft_h2.h.  For image sizes whose folded matrix does not fit LDS (N > 255: the
// reference driver's 501 = 3 * 167, the deconvolution path's 512 = 4 * 128; reference transforms:
// surfh/ToolsDir/jax_utils.py:30-41 dft / idft, ortho rfft2 / irfft2 on the last two axes).
//
//   decimation in time:  n = R n2 + n1,  k = k2 + M k1
//   X[k2 + M k1] = sum_{n1 < R} w_R^{n1 k1} w_N^{n1 k2} Y_{n1}[k2],     Y_{n1}[k2] = sum_{n2 < M} x[R n2 + n1] w_M^{n2 k2}
//
// A workgroup holds NG groups of R waves; the waves of a group transform the R residue classes of the same 16 complex
// columns.  Wave n1 runs the k loop of dft_h2_kernel unchanged on rows R n2 + n1 (register-direct loads of whole 128-byte
// row segments, symmetry fold n2 <-> M - n2, two fp16 pieces under a per-column running block exponent, the (cos | sin)
// image of the M-point transform resident in LDS, no barrier inside the loop).  The R sub-transforms then meet in LDS:
// 16 rows (and their 16 mirror rows M - r) at a time each wave writes its Y, one workgroup barrier (s_barrier with an LDS
// wait only: the next tile's loads and the previous phase's stores stay in flight), and every wave combines its share of
// the rows -- twiddle, R-point butterfly -- and stores R output rows per input row.  Two exchange buffers alternate, so a
// phase needs one barrier.
//
// All four passes of rfft2 / irfft2 run on this one complex kernel:
//   loader PLAIN  rows 0 .. N-1 of interleaved complex columns (or of a REAL array read as packed pairs a + i b of
//                 neighbouring columns: the r2c pass)
//   loader MIX    the same with the forward model's spectral mix formed on the fly (x = src * sum_t tpl[t][l] mhat[t][k][kb])
//   loader PROD   the same with an element-wise product formed on the fly: x = src * prod or src * conj(prod), `prod` an array of
//                 the same shape (the 2-D deconvolution path's OTF product, spectro_blind_rectangle.py:193-237: the product
//                 array is neither written nor read back)
//   loader PRODADD  PROD plus a third array weighted per (row, column block): x = src * conj(prod) + add_w (dk[k] + dkb[kb]) add,
//                 dk[k] = 2 - 2 cos(2 pi k / N), dkb likewise along the other axis (columns n = kb * LP + l): the quadratic prior of
//                 the plane-wise normal operator, mu_r |D|^2 d^, folded into the adjoint's inverse transform (fusion_CT.py:16-43)
//   loader HPACK  rows 0 .. N/2 of a Hermitian half spectrum, two neighbouring complex columns A, B read as Z = A + i B
//                 with Z[N - k] = conj(A[k]) + i conj(B[k]) (the c2r pass: the transform of Z is a + i b, both real)
//   epilogue PLAIN  N complex rows
//   epilogue HSEP   N/2 + 1 rows of the two Hermitian spectra A = (X[k] + conj X[N-k]) / 2, B = (X[k] - conj X[N-k]) / 2i
//                   of a packed pair (the r2c pass)
#pragma once
#include <hip/hip_runtime.h>

enum { DFT_CT_PLAIN = 0, DFT_CT_MIX = 1, DFT_CT_HPACK = 2, DFT_CT_PROD = 3, DFT_CT_PRODADD = 4 };
enum { DFT_CT_STORE = 0, DFT_CT_HSEP = 1 };

struct DftCtArgs {
    int R = 0, M = 0;                          // N = R * M
    int loader = DFT_CT_PLAIN, epi = DFT_CT_STORE;
    float sgn = -1.f;                          // -1: forward transform (e^{-i ...}), +1: inverse
    float scale = 1.f;                         // applied to the outputs (ortho: 1 / sqrt(N); HSEP: the 1/2 of the separation too)
    const float *src = nullptr;
    long ldb = 0, sB = 0;                      // row pitch / batch stride of src, in floats
    float *dst = nullptr;
    long ldc = 0, sC = 0;
    int ncols = 0;                             // complex (or packed) columns per batch entry, % 64 == 0 (% 128 with a list of super-tiles)
    int batch = 1;
    // MIX loader (column n = kb * LP + l, or batch entry = kb when batch > 1)
    const float *mhat = nullptr, *tpl = nullptr;
    int T = 0, LP = 0;
    long PL = 0, KBP = 0;
    float mhat_self = 1.f, mhat_pair = 1.f;    // Parseval-scaled spectra of the solver (dft_h2.h)
    int mix_Nb = 0;
    // PROD loader: second operand with the layout of src (own pitch / batch stride), +1: src * prod, -1: src * conj(prod)
    const float *prod = nullptr;
    long ldp = 0, sP = 0;
    float prod_sign = 1.f;
    // PRODADD: third operand (pitch / batch stride of src), its weight, the length of the other axis
    const float *add = nullptr;
    float add_w = 0.f;
    int add_Nb = 0;
    // optional list of super-tiles (128 columns, numbered batch-major) to transform, ascending
    const int *vlist = nullptr;
    int nvalid = 0;
    // optional limits per chunk of 2^tabShift columns (chunk = (first column of the tile % tabLP) >> tabShift; 7: 128 complex
    // columns, 6: 64 packed pairs = 128 wavelengths):
    //   ktab[chunk] = k-steps of 16 to run (>= 2): the elements of the sub-sequences beyond are zero or immaterial
    //   rtab[chunk] = largest output row k, counted as min(k, N - k), that anybody reads: the rows beyond are not stored
    const int *ktab = nullptr, *rtab = nullptr;
    int tabLP = 0, tabShift = 7;
};

// device-resident constants of one transform length: the LDS image of the folded M-point (cos | sin) matrices as two
// fp16 pieces, and the twiddles w_N^{n1 k2} (inverse sign) as [M][R - 1] (cos, sin)
struct DftCtPlan {
    int R = 0, M = 0, N = 0, MT = 0, KT = 0, kA = 0;
    unsigned short *img = nullptr;
    float *tw = nullptr;
};

// factorisation the kernel supports for length n (false: none): R in {4, 3, 2} with n % R == 0, 32 < M = n / R <= 190
bool dft_ct_factor(int n, int *R, int *M);
bool dft_ct_supported(int Na, int Nb);
int dft_ct_plan_create(int n, DftCtPlan *out);
void dft_ct_plan_destroy(DftCtPlan *p);
int launch_dft_ct(hipStream_t stream, const DftCtArgs &g, const DftCtPlan &pl);
