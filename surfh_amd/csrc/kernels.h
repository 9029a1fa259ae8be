// Device kernels of the surfh hot path other than the dense GEMM (see gemm_f32.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- spectral mix x OTF (T and C fused in the Fourier domain) -------------------------------
// forward : spec[l] = sotf[l] * sum_t tpl[t,l] * mhat[t]        (spectroModel.py:161,166; mixing.py:232-245)
// adjoint : madj[t] = sum_l tpl[t,l] * conj(sotf[l]) * spec[l]  (spectroModel.py:178,181; mixing.py:263-266)
// planes are stored split: [plane][2 (re,im)][PL] floats, PL = KAP*KBP (zero padded).
// T == 0 : no LMM, mhat has one plane per lambda.
int launch_specmix_fwd(hipStream_t s, const float *mhat, const float *sotf, const float *tpl, float *spec,
                       int T, int L, long PL);
int launch_specmix_adj(hipStream_t s, const float *spec, const float *sotf, const float *tpl, float *partial,
                       float *madj, int T, int L, long PL, int nchunk);

// ---- sparse (ELL) gather shared by all lambda planes -----------------------------------------
// dst[b*dstStride + dst_off[r]] (+)= sum_{e<cnt[r]} val[e*R+r] * src[b*srcStride + col[e*R+r]]
// One table serves: S + box-sum + slit window + decimation (forward), its exact transpose,
// and the reference's interpolating gridding_t (adjoint_ref).
struct EllTable {
    int R = 0, W = 0;              // rows, max entries per row
    const int32_t *cnt = nullptr;  // [R]
    const int32_t *col = nullptr;  // [W][R]
    const float *val = nullptr;    // [W][R]
    const int32_t *dst_off = nullptr;  // [R]
};
int launch_spmm_ell(hipStream_t s, const EllTable &t, const float *src, long srcStride, float *dst, long dstStride,
                    int nblk, int accumulate);

// ---- layout helpers -----------------------------------------------------------------------
int launch_pad_planes(hipStream_t s, const float *src, float *dst, int B, int na, int nb, int nap, int nbp);
int launch_unpad_planes(hipStream_t s, const float *src, float *dst, int B, int na, int nb, int nap, int nbp);
// y[(ps*Ldet + l)*aout + a] = sum_k cpart[k][l*NP + ps*aout + a]
int launch_y_from_cpart(hipStream_t s, const float *cpart, long slab, int nsplit, float *y, int PS, int Ldet,
                        int aout, int NP);
// ymat[l*NP + ps*aout + a] = y[(ps*Ldet + l)*aout + a]
int launch_ymat_from_y(hipStream_t s, const float *y, float *ymat, int PS, int Ldet, int aout, int NP);
int launch_fill_zero(hipStream_t s, float *p, long n);

// ---- CG vector kernels (qmm.lcg loop body; fusion_CT.py:16-43 priors) --------------------------
// q += mu_reg * (Dr^T Dr + Dc^T Dc) d   on [T][na][nb], circular
int launch_prior_add(hipStream_t s, const float *d, float *q, int T, int na, int nb, float mu_reg);
int launch_scale(hipStream_t s, float *x, long n, float a);
// out[0] = sum a*b (fp64 accumulation); scratch holds >= 1024 doubles
int launch_dot(hipStream_t s, const float *a, const float *b, long n, double *scratch, double *out);
// step = rr / dq[0];  x += step d ; r -= step q ; out_rr = r.r
int launch_cg_step(hipStream_t s, float *x, float *r, const float *d, const float *q, long n, const double *rr,
                   const double *dq, double *scratch, double *out_rr);
// x += step d only (used when the residual is refreshed from scratch)
int launch_cg_xupdate(hipStream_t s, float *x, const float *d, long n, const double *rr, const double *dq);
// d = r + (rr_new/rr_old) d
int launch_cg_dir(hipStream_t s, float *d, const float *r, long n, const double *rr_new, const double *rr_old);
// r = b - q
int launch_residual(hipStream_t s, float *r, const float *b, const float *q, long n);
