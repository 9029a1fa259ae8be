// Device kernels of the surfh hot path other than the dense GEMM (see gemm_f32.h).
#pragma once
// templates (abundance maps) the spectral-mix kernels hold in registers per frequency bin
constexpr int SURFH_MAX_TEMPLATES = 8;
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- spectral mix x OTF (T and C fused in the Fourier domain) -------------------------------
// forward : spec[k][l] = sotf[k][l] * sum_t tpl[t,l] * mhat[t][k]   (spectroModel.py:161,166; mixing.py:232-245)
// adjoint : madj[t][k] = sum_l tpl[t,l] * conj(sotf[k][l]) * spec[k][l] (spectroModel.py:178,181; mixing.py:263-266)
// Wavelength is the INNERMOST axis of every large array: spectra are [2 (re,im)][PL][LP] floats,
// PL = KAP*KBP frequency bins, LP = padded number of owned planes (zero padded); mhat is [T][2][PL].
// T == 0 (no LMM): mhat has the spectrum layout and the operation is element-wise.
// ilv = 1: the spectra (sotf, spec, and mhat when T == 0) are interleaved complex [PL][LP][2] instead (plans whose
// transforms run in dft_h2.hip)
int launch_specmix_fwd(hipStream_t s, const float *mhat, const float *sotf, const float *tpl, float *spec,
                       int T, long PL, int LP, int ilv = 0);
// options of the interleaved adjoint reduction (T > 0): `lim` = [2][LP / 128] per chunk of 128 wavelengths the largest k_beta and the
// largest folded k_alpha inside the OTF's support (bins beyond were not written to `spec` and are skipped);  Nb != 0: the output is
// the solver's Parseval-scaled half spectrum (bins that are their own conjugate x out_self, the others x out_pair) with
// prior_mu * |D|^2 * prior_src added (surfh_normal_spec_dev; same arithmetic as the fused tail's reduction, dft_h2.h)
// T == 0 (plane-wise product): prior_src == madj means that `madj` holds, on entry, the spectrum of the vector the normal operator is
// being applied to; the kernel then writes out_self * conj(H) Y + prior_mu * |D|^2 * that (the quadratic prior folded in).
struct SpecmixAdjOpt {
    const int *lim = nullptr;
    int Na = 0;
    long KBP = 0;
    int Nb = 0;
    float out_self = 1.f, out_pair = 1.f;
    const float *prior_src = nullptr;
    float prior_mu = 0.f;
};
int launch_specmix_adj(hipStream_t s, const float *spec, const float *sotf, const float *tpl, float *madj,
                       int T, long PL, int LP, bool f64 = false, int ilv = 0, const SpecmixAdjOpt *opt = nullptr);

// hth[(t,t')][k] = sum_l tpl[t,l] tpl[t',l] |sotf[k][l]|^2   (mixing.py:177-203), full T x T stored
int launch_wct_hessian(hipStream_t s, const float *sotf, const float *tpl, float *hth, int T, long PL, int LP, int ilv = 0);
// out[t][c][k] = sum_t' hth[t][t'][k] in[t'][c][k]            (mixing.py:102-126 with di = dj = 1)
int launch_wct_hess_apply(hipStream_t s, const float *hth, const float *in, float *out, int T, long PL);

// ---- sparse row gather, vectorised over wavelength ----------------------------------------------
// dst[dst_off[r] + l] (+)= sum_{e<cnt[r]} val[r*W+e] * src[col[r*W+e] + l]   for l in [0, nlam)
// One workgroup per (row, 1024-wavelength chunk): the table entries are workgroup-uniform (scalar
// loads), every tap is a contiguous coalesced read.  One table format serves S + box-sum + slit
// window + decimation (forward), its exact transpose, and the reference's interpolating
// gridding_t (adjoint_ref).
struct EllTable {
    int R = 0, W = 0;              // rows, max entries per row
    const int32_t *cnt = nullptr;  // [R]
    const int64_t *col = nullptr;  // [R][W]  source offset (floats) of wavelength 0
    const float *val = nullptr;    // [R][W]
    const int64_t *dst_off = nullptr;  // [R]
    // accumulate mode only, optional: bit j of rmw[r] set = the 1024-wavelength chunk j of row r may already hold another
    // table's contribution and is read-modify-written; clear = the destination is known to be zero, plain store
    const uint32_t *rmw = nullptr;
    // ... or exactly: wavelengths [rng[r].x, rng[r].y) of row r (relative to the window, multiples of 4) were written earlier in
    // this pass and are read-modify-written, all others are plain stores -- the destination then needs NO clearing (takes precedence)
    const int2 *rng = nullptr;
};
// Scatter table with its rows taken SCATTER_G at a time: neighbouring cube pixels receive from almost the same operand
// rows, so a group reads the union of its members' taps once and applies one weight per member (0 where a member does not
// use the tap).  dst < 0 marks a missing member.
constexpr int GATHER_G = 4, SCATTER_G = 8, GROUP_MAX = 8;      // rows per group: gather / scatter (measured: 4 -> 8 rows costs the
                                                                // gather 0.34 -> 0.45 ms and gains the scatter 0.32 -> 0.28 ms per step on config 3; scatter with 16: 0.31)
struct GroupTable {
    int NG = 0, W = 0, G = 0;          // groups, max union taps per group, rows per group (GATHER_G or SCATTER_G)
    const int32_t *cnt = nullptr;      // [NG]
    const int64_t *col = nullptr;      // [NG][W]
    const float *val = nullptr;        // [NG][W][G]
    const int64_t *dst = nullptr;      // [NG][G]
    const uint32_t *rmw = nullptr;     // [NG][G] read-modify-write chunk masks (as EllTable::rmw)
    const int2 *rng = nullptr;         // [NG][G] exact read-modify-write wavelength ranges (as EllTable::rng)
};
// float64-accumulating twin of launch_spmm_rows (every row, read-modify-write where `accumulate`)
int launch_spmm_rows_f64acc(hipStream_t s, const EllTable &t, const float *src, float *dst, int nlam, int accumulate);
int launch_spmm_group_scatter(hipStream_t s, const GroupTable &t, const float *src, float *dst, int nlam);
// the gather of launch_spmm_rows_f16 on a grouped table (members = rows neighbouring in cube-location order)
int launch_spmm_group_gather_f16(hipStream_t s, const GroupTable &t, const float *src, unsigned short *dst16, long plane, int nlam,
                                 float *bscale, int NP, long K, int LinP);

int launch_spmm_rows(hipStream_t s, const EllTable &t, const float *src, float *dst, int nlam, int accumulate);
// the gather writing its [NP][K] output as two fp16 pieces dst16[q*plane + ...] of value / block scale; one power-of-two
// scale per workgroup, i.e. per (row, segment) with segment = (column / LinP) * nchunk + (column % LinP) / 1024, nchunk =
// ceil(LinP / 1024): bscale[segment][NP] (entries of segments the table never writes must be 1)
int launch_spmm_rows_f16(hipStream_t s, const EllTable &t, const float *src, unsigned short *dst16, long plane, int nlam, float *bscale,
                         int NP, long K, int LinP);
int launch_dequant_f16x2(hipStream_t s, const unsigned short *src16, long plane, const float *bscale, float *dst, int NP, long K, int LinP,
                         int nchunk);
// half spectra [planes][KAP][KBP] <-> the spectral-domain solver's Parseval-scaled form; the quadratic prior on such vectors
int launch_spec_scale(hipStream_t s, const float *src, float *dst, int planes, long PL, long KBP, int Nb, float f_self, float f_pair);
int launch_spec_prior_add(hipStream_t s, const float *d, float *q, int planes, int Na, int Nb, long PL, long KBP, float mu_reg);
long ymat_from_y_waves(int PS, int Ldet, int aout);
// normal operator: sum of the forward GEMM's K slabs straight into the adjoint GEMM's operand -- the two fp16 pieces of
// ymat [NP][LdetP] (rows >= nrows and columns >= Ldet zero) with one power-of-two scale per row, rowmax[NP] = max |row| as
// bit patterns.  The same bits as y_from_cpart + ymat_from_y + launch_split_rows2h, without materialising y.
int launch_ymat16_from_cpart(hipStream_t s, const float *cpart, long slab, int nsplit, unsigned short *dst16, long plane, unsigned *rowmax,
                             int NP, int nrows, int Ldet, int LdetP);

// [L][Na][Nb] (wavelength-major, the reference's cube layout) <-> [NBP][NAP][LP] (wavelength innermost)
int launch_cube_to_lam_inner(hipStream_t s, const float *src, float *dst, int l0, int L, int na, int nb, int nap, int LP);
int launch_cube_from_lam_inner(hipStream_t s, const float *src, float *dst, int l0, int L, int na, int nb, int nap, int LP);

// ---- layout helpers -----------------------------------------------------------------------
int launch_pad_planes(hipStream_t s, const float *src, float *dst, int B, int na, int nb, int nap, int nbp);
int launch_unpad_planes(hipStream_t s, const float *src, float *dst, int B, int na, int nb, int nap, int nbp);
// y[(ps*Ldet + l)*aout + a] = sum_k cpart[k][(ps*aout + a)*LdetP + l]
int launch_y_from_cpart(hipStream_t s, const float *cpart, long slab, int nsplit, float *y, int PS, int Ldet,
                        int aout, int LdetP);
// ymat[(ps*aout + a)*LdetP + l] = y[(ps*Ldet + l)*aout + a]
int launch_ymat_from_y(hipStream_t s, const float *y, float *ymat, int PS, int Ldet, int aout, int LdetP, unsigned *pmax = nullptr,
                       unsigned *rowmax = nullptr, int NP = 0);
int launch_fill_zero(hipStream_t s, float *p, long n);

// ---- CG vector kernels (qmm.lcg loop body; fusion_CT.py:16-43 priors) --------------------------
// q += mu_reg * (Dr^T Dr + Dc^T Dc) d   on [T][na][nb], circular
int launch_prior_add(hipStream_t s, const float *d, float *q, int T, int na, int nb, float mu_reg);
// q += mu_reg * L^T L d, L = circular 3 x 3 Laplacian (the reference's joint prior, fusion_CT.py:45-62)
int launch_prior_joint_add(hipStream_t s, const float *d, float *q, int T, int na, int nb, float mu_reg);
int launch_scale(hipStream_t s, float *x, long n, float a);
// out[0] = sum a*b (fp64 accumulation); scratch holds >= 1024 doubles
int launch_dot(hipStream_t s, const float *a, const float *b, long n, double *scratch, double *out);
// step = rr / dq[0];  x += step d ; r -= step q ; out_rr = r.r
int launch_cg_step(hipStream_t s, float *x, float *r, const float *d, const float *q, long n, const double *rr,
                   const double *dq, double *scratch, double *out_rr);
// x += step d only (used when the residual is refreshed from scratch)
int launch_cg_xupdate(hipStream_t s, float *x, const float *d, long n, const double *rr, const double *dq);
// the CG iteration with device-resident scalars in three launches: per-block partial sums of a dot product (`parts`: room for
// dot_parts_stride() doubles), consumed by the next launch, which sums them itself (reduce_final_kernel's order)
int launch_dot_parts(hipStream_t s, const float *a, const float *b, long n, double *parts);
int launch_cg_step_parts(hipStream_t s, float *x, float *r, const float *d, const float *q, long n, const double *rr, const double *dq_parts,
                         double *dq_out, double *rr_parts);
int launch_cg_dir_parts(hipStream_t s, float *d, const float *r, long n, const double *rr_parts, const double *rr_old, double *rr_out);
int dot_parts_stride();
// d = r + (rr_new/rr_old) d
int launch_cg_dir(hipStream_t s, float *d, const float *r, long n, const double *rr_new, const double *rr_old);
// r = b - q
int launch_residual(hipStream_t s, float *r, const float *b, const float *q, long n);
// out = a + beta b
int launch_lincomb(hipStream_t s, float *out, const float *a, const float *b, long n, double beta);
// 3MG move: mv = s0 d + s1 m ; x += mv ; m = mv ; qm = s0 qd + s1 qm ; r -= qm (when update_r)
int launch_mmmg_update(hipStream_t s, float *x, float *r, const float *d, float *m, float *qm, const float *qd, long n, double s0,
                       double s1, int update_r);
// CG on independent planes ([nplanes][npix] arrays, per-plane scalars in double arrays of nplanes)
int launch_dot_planes(hipStream_t s, const float *a, const float *b, int nplanes, long npix, double *out);
int launch_cg_step_planes(hipStream_t s, float *x, float *r, const float *d, const float *q, int nplanes, long npix, const double *rr,
                          const double *dq, double *rrn, int update_r);
int launch_cg_dir_planes(hipStream_t s, float *d, const float *r, int nplanes, long npix, const double *rrn, double *rr);
// the plane-wise CG blocks on wavelength-innermost arrays [Nb rows][NAP][LP] (the cube layout; kernels.hip): per-wavelength
// scalars are [LP] doubles, `part` a work buffer of pn_part_doubles(LP) doubles
int launch_pn_prior_dot(hipStream_t s, const float *d, float *q, int Na, int Nb, int NAP, long LP, float mu_reg, double *part, double *out);
int launch_pn_dot(hipStream_t s, const float *x, const float *y, int Na, int Nb, int NAP, long LP, double *part, double *out);
int launch_pn_step(hipStream_t s, float *x, float *r, const float *d, const float *q, int Na, int Nb, int NAP, long LP, const double *rr,
                   const double *dq, double *part, double *rrn, int update_r);
int launch_pn_dir(hipStream_t s, float *d, const float *r, int Na, int Nb, int NAP, long LP, const double *rrn, const double *rr);
size_t pn_part_doubles(long LP);
// 3MG per plane: rr = r.r, mqm = m.Qm, d = r - (r.Qm / m.Qm) m ; then the 2x2 step in [d, m] with the carried images
int launch_mmmg_dir_planes(hipStream_t s, float *d, const float *r, const float *m, const float *qm, int nplanes, long npix,
                           double *rr, double *mqm);
int launch_mmmg_step_planes(hipStream_t s, float *x, float *r, const float *d, float *m, float *qm, const float *qd, int nplanes,
                            long npix, const double *mqm, int update_r);
// (hth + diag(mu reg)) z = in per frequency bin (reg < 0 marks padding bins); *flag |= 1 on a non-positive pivot
int launch_wct_solve(hipStream_t s, const float *hth, const float *reg, const double *mu, const float *in, float *out, int T,
                     long PL, int *flag);
// linear mixing model, plane-major arrays: cube[l][i] = sum_t tpl[t][l] maps[t][i];  maps[t][i] = sum_l tpl[t][l] cube[l][i]
int launch_lmm_maps2cube(hipStream_t s, const float *maps, const float *tpl, float *cube, int T, int L, long npix);
int launch_lmm_cube2maps(hipStream_t s, const float *cube, const float *tpl, float *maps, int T, int L, long npix);
