/* surfh_amd -- C ABI of the MI355X-native surfh hot path.
 *
 * One `surfh_plan` = one GPU + one HIP stream + the set of MRS channels that GPU
 * owns.  It is the drop-in replacement for the arithmetic behind
 *   spectroSigRLSCT.forward / .adjoint     (surfh/Models/spectroModel.py:158-185)
 *   Channel.forward / .adjoint             (surfh/Models/spectroModelChannel.py:215-264)
 *   Slicer.slicing / slicing_t             (surfh/Models/slicer.py:64-84)
 *   jax_utils.{lmm_*, dft, idft, dft_mult, wblur_subSampling, wblur_t}
 *                                          (surfh/ToolsDir/jax_utils.py:10-91)
 *   cythons_files.solve_2D_hypercube       (surfh/ToolsDir/cythons_files.pyx:163-193)
 *   NpDiff_r / NpDiff_c + qmm.lcg / qmm.mmmg loops (surfh/Simulation/fusion_CT.py:16-43,194-225)
 * All file:line citations are relative to the reference tree (sidiso/surfh @ 2025-02-04).
 *
 * Conventions
 *   - plain C, no torch types.  Host pointers unless the name ends in `_dev`.
 *   - every function returns 0 on success, non-zero on error; the message is
 *     available from surfh_last_error() (thread-local, valid until the next call).
 *   - the geometry tables are produced by the host side (surfh_amd/geometry.py),
 *     which restates instru.py / slicer.py; the library only consumes them.
 *   - arithmetic type: fp32 on device.  The dense stages evaluate every fp32 product as a few 16-bit
 *     matrix-core products of split operands, accumulated in fp32: the spectral blur as three products of a
 *     two-piece round-to-nearest fp16 split (22 mantissa bits) with per-row / per-segment operand scales, the DFT passes
 *     the same way under a per-column running block exponent (axis lengths neither transform kernel covers -- a prime
 *     above 255, fewer than 32 points: dense fp32-input MFMA products); fp32-input MFMA kernels behind SURFH_*
 *     environment switches; inner products of the
 *     solvers accumulate in fp64.  surfh_config.verify selects float64-accumulating kernels throughout.
 *   - environment switches read at plan creation (A/B paths, parity-tested where they engage): SURFH_DFT_H2=0 (axes
 *     <= 255 on the Cooley-Tukey kernel too), SURFH_DFT_CT=0, SURFH_DFT_DENSE=1, SURFH_NO_FUSED_MIX=1, SURFH_WBLUR_FP32=1,
 *     SURFH_WBLUR_FAR=0, SURFH_WBLUR_PERM=0, SURFH_GEMM_GROUPED=0, SURFH_ADJ_FUSED=0 (separate adjoint reduction kernel),
 *     SURFH_OTF_SUPPORT=0, SURFH_OTF_RANGES=0, SURFH_ALPHA_RANGE=0 (transform the whole cube), SURFH_GATHER_SORTED=0,
 *     SURFH_GATHER_GROUPED=0, SURFH_SCATTER_GROUPED=0, SURFH_SCATTER_RMW_ALL=1, SURFH_ADJ_CLEAR=1, SURFH_OVERLAP=1,
 *     SURFH_PLANES_NATIVE=0, SURFH_OTF_PROD=0 (plane-wise model: the OTF products as kernels of their own); surfh_config.exact switches the far class / the support lists off per plan; read per call:
 *     SURFH_SPECTRAL_CG=0 (solver vectors = maps); read once per process: SURFH_NORMAL_FUSED=0 (the normal operator
 *     goes through y).
 */
#ifndef SURFH_AMD_H
#define SURFH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct surfh_plan surfh_plan;

/* One MRS channel (what Channel.__init__ + Slicer derive; spectroModelChannel.py:27-108). */
typedef struct {
    int32_t wslice_start, wslice_stop; /* IFU.wslice on the cube axis (instru.py:649-658)            */
    int32_t n_pointings;               /* P                                                         */
    int32_t n_slit;                    /* S                                                         */
    int32_t n_lambda_out;              /* Ldet = len(instr.wavel_axis)                              */
    int32_t n_alpha_out;               /* ceil(npix_slit_alpha_width / srf)                         */
    int32_t srf;                       /* super-resolution factor (instru.py:67-84)                 */
    int32_t na, nb;                    /* local grid (len(local_alpha_axis), len(local_beta_axis))  */
    int32_t alpha0;                    /* first local alpha row of the slit window (slicer.py:118)  */
    int32_t n_alpha_slit;              /* length of the alpha window                                */
    int32_t n_beta_slit;               /* npix_slit_beta_width (slicer.py:45-48)                    */
    const int32_t *slit_beta0;         /* [S]    first local beta column of each slit               */
    const double *slit_weights;        /* [S][n_beta_slit] beta-edge weights (slicer.py:148-168)    */
    const int32_t *grid_i0;            /* [P][na*nb] lower alpha index (cythons_files.pyx:109-154)  */
    const int32_t *grid_i1;            /* [P][na*nb] lower beta index                               */
    const double *grid_y0;             /* [P][na*nb] normalised alpha distance                      */
    const double *grid_y1;             /* [P][na*nb] normalised beta distance                       */
    const double *wpsf;                /* [Ldet][Lin][n_beta_slit] spectral PSF (instru.py:499-572);
                                          NULL: no spectral blur, y[l][(p,s,a)] = sum over the slit's beta
                                          columns (MRSBlurred, spectro_blind_rectangle.py:193-209)           */
    /* reference-compatible back-interpolation tables for surfh_adjoint_ref (gridding_t,
       spectroModelChannel.py:180-199); may be NULL if adjoint_ref is never called.               */
    const int32_t *gt_i0;              /* [P][Na*Nb] lower local alpha index                        */
    const int32_t *gt_i1;              /* [P][Na*Nb] lower local beta index                         */
    const double *gt_y0;               /* [P][Na*Nb]                                                */
    const double *gt_y1;               /* [P][Na*Nb]                                                */
    const uint8_t *gt_inside;          /* [P][Na*Nb] 1 if the global pixel falls inside the local grid */
    /* alpha window summed into one detector sample: local rows alpha0 + a*srf + box_shift + [0, box_len) (circular).
       box_len = 0 means srf with shift 0, the operator's box sum (`_otf_sr * decalf`, spectroModelChannel.py:81-83,
       104-108).  The real-data projections use other windows: plain decimation (box_len 1,
       realData_cubeToSlice :303-309) and the box kernel without its re-centring shift (box_shift = -int((srf-1)/2),
       realData_sliceToCube :331-332).                                                             */
    int32_t box_len, box_shift;
} surfh_channel_desc;

typedef struct {
    int32_t n_alpha, n_beta;           /* cube spatial shape                                        */
    int32_t n_lambda;                  /* cube planes Lc                                            */
    int32_t n_templates;               /* T <= 8; 0 => no LMM (input is the cube itself)            */
    const double *templates;           /* [T][Lc] or NULL                                           */
    const double *sotf;                /* [Lc][n_alpha][n_beta/2+1] complex128 interleaved (re,im);
                                          NULL (only with n_templates = 0): no spatial blur, H = 1  */
    int32_t n_channels;
    const surfh_channel_desc *channels;
    int32_t device;                    /* HIP device ordinal                                        */
    void *stream;                      /* hipStream_t to run on, or NULL: the plan creates its own  */
    int32_t split_k_forward;           /* 0 = auto                                                  */
    int32_t verify;                    /* 1 = verification plan: every long sum (DFT products, spectral blur, spectral mix,
                                          gather / scatter rows) accumulated in float64 on plain vector kernels.  Same operator,
                                          same fp32 storage, ~100x slower: for the strict dot test
                                          (test/sandbox_dottest.py:16-27 with randn vectors), not for production. */
    int32_t exact;                     /* production plan without its two approximations (both bounded at plan creation, DESIGN.md
                                          section 4): bit 0 = every K step of the spectral-blur GEMMs keeps all three fp16
                                          products (no far class; same as SURFH_WBLUR_FAR=0), bit 1 = the transform passes visit
                                          the whole spectrum (no OTF-support lists; same as SURFH_OTF_SUPPORT=0).  0 = default. */
} surfh_config;

const char *surfh_last_error(void);
int surfh_version(void);

int surfh_plan_create(const surfh_config *cfg, surfh_plan **out);
int surfh_plan_destroy(surfh_plan *plan);

/* sizes: isize = T*Na*Nb (or Lc*Na*Nb without LMM), osize = sum_c P*S*Ldet*alpha_out */
int64_t surfh_isize(const surfh_plan *plan);
int64_t surfh_osize(const surfh_plan *plan);
void *surfh_stream(const surfh_plan *plan);

/* y = A x      (spectroSigRLSCT.forward, spectroModel.py:158-170)                   */
int surfh_forward(surfh_plan *plan, const float *maps, float *y);
/* x = A^T y    exact transpose of surfh_forward (what CG and the dot-test use)      */
int surfh_adjoint(surfh_plan *plan, const float *y, float *maps);
/* x = reference adjoint with the interpolating gridding_t (spectroModel.py:173-185) */
int surfh_adjoint_ref(surfh_plan *plan, const float *y, float *maps);
/* out = A^T A x                                                                      */
int surfh_fwadj(surfh_plan *plan, const float *x, float *out);

/* device-pointer, asynchronous variants (run on the plan's stream) */
int surfh_forward_dev(surfh_plan *plan, const float *maps_dev, float *y_dev);
int surfh_adjoint_dev(surfh_plan *plan, const float *y_dev, float *maps_dev);
int surfh_adjoint_ref_dev(surfh_plan *plan, const float *y_dev, float *maps_dev);
int surfh_fwadj_dev(surfh_plan *plan, const float *x_dev, float *out_dev);

/* ---- Fourier-domain fused W.C.T operator (Model_WCT, surfh/Models/mixing.py:131-272, di = dj = 1) ----
 * Uses only the plan's sotf / templates (a plan may be created with n_channels = 0 for this).
 * cube is [Lc][Na][Nb] (the reference's layout).                                               */
int surfh_wct_forward(surfh_plan *plan, const float *maps, float *cube);     /* mixing.py:232-245 */
int surfh_wct_adjoint(surfh_plan *plan, const float *cube, float *maps);     /* mixing.py:247-268 */
/* explicit normal operator through the per-frequency T x T Hessian sum_l tpl tpl' |H_l|^2
 * (mixing.py:102-126,177-212,270-272)                                                          */
int surfh_wct_fwadj(surfh_plan *plan, const float *x, float *out);
/* explicit inverse of the regularised normal operator: the minimiser of |y - H x|^2 + sum_t mu_reg[t] |D x_t|^2,
 * one T x T solve per frequency (QuadCriterion3.run_expsol, surfh/ToolsDir/fusion_mixing.py:309-438;
 * algorithms.py:156-184).  reg_freq = |D(f)|^2 on the half spectrum [Na][Nb/2+1] (fusion_mixing.py:364-395).
 * Fails when the matrix is singular at some frequency, where the reference's numpy.linalg.inv raises. */
int surfh_wct_expsol(surfh_plan *plan, const float *cube, const double *mu_reg, const double *reg_freq, float *maps);

/* ---- regularised least squares by linear CG (fusion_CT.py:118-238 + qmm.lcg) ----
 * minimises  mu |y - A x|^2 + mu_reg (|Dr x|^2 + |Dc x|^2).
 * grad_norm receives r.r (max_iter+1 doubles), nit the iterations done.
 * Where the plan offers the spectral-domain calls (surfh_spec_supported) the loop keeps its vectors as the maps' scaled half
 * spectra and its scalars on the device, and reads the trace -- qmm.lcg's stopping test sqrt(r.r) < size * tol -- every 8
 * iterations only: it may run up to 7 iterations past the one that met the tolerance (x and nit are those of the last
 * iteration run).  With a callback (surfh_cg_cb) the test is made after every iteration, as in qmm.lcg.             */
int surfh_cg(surfh_plan *plan, const float *y, double mu, double mu_reg, const float *x0,
             int32_t max_iter, double tol, int32_t refresh, float *x, double *grad_norm, int32_t *nit);
/* The same solver with the per-iteration callback of qmm.lcg (`callback=` at fusion_CT.py:194-225): after
 * iteration `it` (1-based) the callback receives the grad_norm trace so far (it+1 values) and the current
 * iterate copied to the host ([T,Na,Nb] floats, valid during the call).  A non-zero return stops the loop.
 * surfh_forward / surfh_adjoint on the same plan may be called from inside the callback (the criterion
 * trace of fusion_CT.py:163-175 does); the CG building blocks below may not.                          */
typedef int (*surfh_cg_callback)(void *user, int32_t it, const double *grad_norm, const float *x);
int surfh_cg_cb(surfh_plan *plan, const float *y, double mu, double mu_reg, const float *x0,
                int32_t max_iter, double tol, int32_t refresh, float *x, double *grad_norm, int32_t *nit,
                surfh_cg_callback callback, void *user);
/* 3MG, the reference's other solver choice (`method != 'lcg'` -> qmm.mmmg, fusion_CT.py:194-198; algorithms.py:69,106),
 * for the same quadratic criterion: every iteration minimises it exactly over span{-gradient, previous move}
 * (the quadratic majorant of a quadratic objective is the objective).  The 2x2 subspace system is solved in a basis
 * [d, move] with d Q-orthogonal to the previous move and the operator applied to d -- the same iterates as qmm's
 * [-gradient, move] form in exact arithmetic, but as accurate as CG in fp32 (see plan.hip).  One normal-operator
 * application per iteration; the gradient is carried by linearity and recomputed every `refresh` iterations.
 * grad_norm receives |gradient| of x0 and of every iterate (nit+1 doubles, capacity max_iter+1); stops when it falls
 * below size*tol.  callback as surfh_cg_cb. */
int surfh_mmmg(surfh_plan *plan, const float *y, double mu, double mu_reg, const float *x0,
               int32_t max_iter, double tol, int32_t refresh, float *x, double *grad_norm, int32_t *nit,
               surfh_cg_callback callback, void *user);

/* ---- linear mixing model on the device: the drivers' mapsToCube / cubeTomaps
 * (spectroModel.py:187-198, jax_utils.py:10-26).  templates [T][Lc] float64 as in surfh_config,
 * maps [T][Na][Nb], cube [Lc][Na][Nb]; works on any plan (only its device and stream are used).   */
int surfh_maps_to_cube(surfh_plan *plan, const double *templates, int32_t n_templates, int32_t n_lambda,
                       const float *maps, float *cube);
int surfh_cube_to_maps(surfh_plan *plan, const double *templates, int32_t n_templates, int32_t n_lambda,
                       const float *cube, float *maps);

/* The same solver for the plane-wise model (n_templates = 0: x is the cube [Lc][Na][Nb]): every plane is an independent
 * 2-D problem  mu |y_l - A_l x_l|^2 + mu_reg (|Dr x_l|^2 + |Dc x_l|^2)  with its own CG scalars -- the reference's 2-D
 * deconvolution (surfh/Simulation/criterion_2D.py:60-250, scripts/deconvolution_mrs_noRotation.py) batched over
 * wavelength.  grad_norm receives r_l.r_l as [max_iter+1][Lc]; the loop stops when every plane is below the tolerance. */
int surfh_cg_planes(surfh_plan *plan, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter,
                    double tol, int32_t refresh, float *x, double *grad_norm, int32_t *nit);
/* 3MG on the plane-wise model -- what `method = "qmm"` of the 2-D deconvolution driver selects
 * (scripts/deconvolution_mrs_noRotation.py:199-212 -> criterion_2D.py:190-193 -> qmm.mmmg).  Scheme of surfh_mmmg with
 * per-plane scalars; grad_norm receives |gradient_l| as [max_iter+1][Lc].                                            */
int surfh_mmmg_planes(surfh_plan *plan, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter,
                      double tol, int32_t refresh, float *x, double *grad_norm, int32_t *nit);

/* The plane-wise CG with the data and the iterate resident on the device and no host synchronisation inside the loop (drivers
 * that keep their cubes in HBM; bench.py --config 5).  begin: b = mu A^T y, r = b - Q x, d = r, with x_dev [Lc][Na][Nb] the start
 * and from then on the current iterate (the caller's buffer, updated in place by step);  step: `iters` more iterations of the loop
 * of surfh_cg_planes (residual recomputed every `refresh` iterations, counted from begin);  rr: r_l.r_l of the current iterate,
 * [Lc] doubles on the host (synchronises the plan's stream).  All pointers except rr_host are device pointers. */
int surfh_cg_planes_begin_dev(surfh_plan *plan, const float *y_dev, double mu, double mu_reg, float *x_dev);
int surfh_cg_planes_step_dev(surfh_plan *plan, int32_t iters, int32_t refresh);
int surfh_cg_planes_rr(surfh_plan *plan, double *rr_host);

/* the two plane-wise solvers with qmm's per-iteration callback (criterion_2D.py:163-225): grad_norm is the trace so far,
 * [it + 1][Lc] values, x the current iterate [Lc][Na][Nb] on the host; a non-zero return stops the loop */
int surfh_cg_planes_cb(surfh_plan *plan, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter,
                       double tol, int32_t refresh, float *x, double *grad_norm, int32_t *nit,
                       surfh_cg_callback callback, void *user);
int surfh_mmmg_planes_cb(surfh_plan *plan, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter,
                         double tol, int32_t refresh, float *x, double *grad_norm, int32_t *nit,
                         surfh_cg_callback callback, void *user);

/* CG building blocks on device vectors, for the multi-GPU driver (one plan per rank,
 * RCCL all-reduce of `q` between surfh_normal_dev and surfh_cg_step_dev).            */
int surfh_normal_dev(surfh_plan *plan, const float *d_dev, float *q_dev, double mu);          /* q  = mu A^T A d   */
int surfh_prior_add_dev(surfh_plan *plan, const float *d_dev, float *q_dev, double mu_reg);   /* q += mu_reg L d   */
/* The same building blocks with the solver's vectors in the Fourier domain of the maps (no transform of the maps, no padding and
 * no prior kernel inside the iteration: the forward model reads the spectra in the loader of its first transform pass, the
 * adjoint's last pass writes them).  A vector holds surfh_spec_size() floats: [T][2 (re, im)][KAP][KBP], padding zero, bin
 * (ka, kb) of the unitary half spectrum (rfft2 / sqrt(Na Nb)) multiplied by sqrt(2) unless the bin is its own conjugate
 * (kb = 0 or 2 kb = Nb) -- so plain dot products of such vectors equal the dot products of the maps, and the CG recurrences
 * (surfh_cg_iter_dev ...) run on them unchanged.  The quadratic prior here is the separated circular first differences
 * (surfh_set_prior 0), diagonal in this basis.  Available where the fused transform passes are (surfh_spec_supported).    */
int surfh_spec_supported(surfh_plan *plan);                                                    /* 1 / 0 */
int64_t surfh_spec_size(surfh_plan *plan);
int surfh_to_spec_dev(surfh_plan *plan, const float *x_dev, float *xt_dev);                    /* maps -> vector */
int surfh_from_spec_dev(surfh_plan *plan, const float *xt_dev, float *x_dev);                  /* vector -> maps */
int surfh_forward_spec_dev(surfh_plan *plan, const float *dt_dev, float *y_dev);               /* y = A maps(dt) */
/* qt = mu spectra(A^T y) (+ mu_reg L dt if dt_dev != NULL: only where qt is not summed over ranks afterwards) */
int surfh_adjoint_spec_dev(surfh_plan *plan, const float *y_dev, float *qt_dev, double mu, const float *dt_dev, double mu_reg);
/* qt = mu A^T A dt (+ mu_reg L dt if mu_reg != 0) */
int surfh_normal_spec_dev(surfh_plan *plan, const float *dt_dev, float *qt_dev, double mu, double mu_reg);
int surfh_prior_spec_add_dev(surfh_plan *plan, const float *dt_dev, float *qt_dev, double mu_reg);   /* qt += mu_reg L dt */
/* which quadratic regulariser L the solvers and surfh_prior_add_dev apply (QuadCriterion_MRS's `gradient`, fusion_CT.py:98-106,141-162):
 * 0 = "separated": Dr^T Dr + Dc^T Dc, circular first differences NpDiff_r / NpDiff_c (fusion_CT.py:16-43) -- the default;
 * 1 = "joint": D^T D with D the circular convolution by the 3 x 3 Laplacian (Difference_Operator_Joint, fusion_CT.py:45-62).  */
int surfh_set_prior(surfh_plan *plan, int32_t kind);
int surfh_dot_dev(surfh_plan *plan, const float *a_dev, const float *b_dev, int64_t n, double *out_host);
/* x += s d ; r -= s q ; returns r.r  (s = rr / d.q computed on device from rr_in)    */
int surfh_cg_step_dev(surfh_plan *plan, float *x_dev, float *r_dev, const float *d_dev,
                      const float *q_dev, int64_t n, double rr_in, double *rr_out_host);
/* d = r + beta d */
int surfh_cg_dir_dev(surfh_plan *plan, float *d_dev, const float *r_dev, int64_t n, double beta);
/* the two calls above fused (one host synchronisation per CG iteration instead of two):
 * x += s d, r -= s q with s = rr_in / d.q; *rr_out = r.r; d = r + (*rr_out / rr_in) d        */
int surfh_cg_iter_dev(surfh_plan *plan, float *x_dev, float *r_dev, float *d_dev, const float *q_dev, int64_t n,
                      double rr_in, double *rr_out);

/* The same recurrences with every scalar resident on the device -- NO host synchronisation: r.r of the current iterate lives in
 * the plan, each call appends the new r.r to a device-side trace.  The multi-GPU loop (surfh_amd/fusion.py) is then
 * normal operator -> RCCL all-reduce -> one of these calls, all asynchronous on the plan's stream; the host reads the trace
 * every few iterations for the stopping test (surfh_cg_trace synchronises).                                                 */
int surfh_cg_begin_dev(surfh_plan *plan, const float *r_dev, int64_t n);                      /* rr = r.r, trace = [rr]        */
int surfh_cg_iter_nosync_dev(surfh_plan *plan, float *x_dev, float *r_dev, float *d_dev, const float *q_dev, int64_t n);
/* residual refresh of qmm.lcg, in two halves around the caller's normal operator on x:
 *   x += (rr / d.q) d          then, with q = Q x:   r = b - q; rr' = r.r; d = r + (rr' / rr) d; rr = rr'                      */
int surfh_cg_xupdate_nosync_dev(surfh_plan *plan, float *x_dev, const float *d_dev, const float *q_dev, int64_t n);
int surfh_cg_refresh_nosync_dev(surfh_plan *plan, float *r_dev, const float *b_dev, const float *q_dev, float *d_dev, int64_t n);
int32_t surfh_cg_trace(surfh_plan *plan, double *out_host, int32_t capacity);                 /* -> number of entries, -1 on error */
/* r = b - q */
int surfh_residual_dev(surfh_plan *plan, float *r_dev, const float *b_dev, const float *q_dev, int64_t n);

/* ---- instrumentation ---- */
/* enable/disable per-kernel HIP-event timing on the plan's stream */
int surfh_profile_enable(surfh_plan *plan, int32_t on);
/* restrict the timing to stages whose name starts with `prefix` (NULL or "": all).  Every bracketed stage costs two event
 * packets on the stream; bracketing all ~45 stages of an iteration was measured to slow it by 4.6 %, so a benchmark times
 * only the kernel group it reports inside its timed region. */
int surfh_profile_filter(surfh_plan *plan, const char *prefix);
/* number of distinct kernel names timed since the last reset */
int32_t surfh_profile_count(surfh_plan *plan);
/* i-th entry: name, launches, total milliseconds */
int surfh_profile_get(surfh_plan *plan, int32_t i, const char **name, int64_t *launches, double *ms);
int surfh_profile_reset(surfh_plan *plan);

/* copy an internal buffer to the host for stage-level parity tests.
 * which: "blurred" [Lown][NaP][NbP], "xs:<c>" [Kp][Np], "gcube" [Lown][NaP][NbP], ...
 * returns the number of floats written (<= capacity) or a negative error.            */
int64_t surfh_debug_copy(surfh_plan *plan, const char *which, float *out, int64_t capacity);
int surfh_debug_dims(surfh_plan *plan, const char *which, int64_t dims[4]);

/* stand-alone fp32 MFMA GEMM self-test hook: C[M][N] = A[M][K] B[K][N] (host buffers) */
int surfh_gemm_selftest(int32_t device, int32_t M, int32_t N, int32_t K, int32_t split_k,
                        const float *A, const float *B, float *C);
/* (tile, K step) pairs the last two-piece fp16 self-test (SURFH_SELFTEST_F16X2=2: with K-step lists, as the spectral-blur
 * GEMMs of a plan) ran with all three products / with the leading product only */
int surfh_gemm_selftest_ksteps(int64_t near_far[2]);
/* host only (no GPU): the K-step classes the spectral-blur GEMMs of a plan would use for the constant operand B [n][ldb]
 * (k columns, k % 32 == 0): records[(tile) * (2 + k / 32)] = n_near, n_far, near steps ascending, far steps ascending
 * (entry = step | segment << 16).  perm_p / perm_lin: the adjoint's tile shape (0: tiles of 256 consecutive rows).
 * Returns the number of tiles, or a negative error.                                                                     */
int32_t surfh_klist_classify(const float *B, int32_t n, int32_t k, int64_t ldb, int32_t perm_p, int32_t perm_lin,
                             int32_t *records, int64_t capacity);

/* ---- masked linear mixing model (MixingST, surfh/Models/mixing.py:276-337; kernels c_fast_forward_TST,
 * c_fast_adjoint_TST, c_precompute_TST of surfh/ToolsDir/cythons_files.pyx:370-463) ----
 * voxels: [n_voxels][3] (lambda, i, j) as `fast_selection_arr`; S: [Lc][Na][Nb] float mask for fwadj's TST
 * (ones with zeros at `selection_arr`, mixing.py:320-321) or NULL.  templates are cast to float32 like the reference. */
typedef struct surfh_tst surfh_tst;
int surfh_tst_create(int32_t n_alpha, int32_t n_beta, int32_t n_lambda, int32_t n_templates, const double *templates,
                     const int32_t *voxels, int64_t n_voxels, const float *S, int32_t device, surfh_tst **out);
int surfh_tst_destroy(surfh_tst *t);
int surfh_tst_forward(surfh_tst *t, const float *maps, float *cube);     /* mixing.py:301-306 */
int surfh_tst_adjoint(surfh_tst *t, const float *cube, float *maps);     /* mixing.py:308-313 */
int surfh_tst_fwadj(surfh_tst *t, const float *maps, float *out);        /* mixing.py:316-317 */
const char *surfh_tst_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SURFH_AMD_H */
