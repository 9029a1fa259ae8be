// Host side of the C ABI (include/surfh_amd.h): plan construction, table building,
// the forward / adjoint pipelines and the CG loop.  All device work goes through
// gemm_f32.hip and kernels.hip on one HIP stream.
//
// Data layout: WAVELENGTH IS THE INNERMOST AXIS of every large device array (lambda is the batch
// dimension of every stage of the reference, so making it contiguous turns every kernel into
// coalesced streaming and every dense stage into one large GEMM):
//   spectra  sotf, spec      [2 (re,im)][KAP][KBP][LP]
//   cube     blurred / g     [NBP (beta)][NAP (alpha)][LP]
//   operand  Xs (per channel)[NP = (p,s,a)][n_beta_slit][LinP]      K index = (b', lambda)
// LP = owned planes padded to 128, all other dims padded to 64, padding is zero.
//
// Pipeline (per plan = per GPU), reference citations relative to /root/reference:
//   forward  (spectroModel.py:158-170, spectroModelChannel.py:215-231)
//     maps --pad--> rfft2 (2 small GEMMs) --> mhat[T][2][KAP][KBP]
//     spec[k][l] = sotf[k][l] * sum_t tpl[t,l] mhat[t][k]             (T and C fused, Fourier domain)
//     blurred    = irfft2(spec): two GEMMs with the DFT matrices as the A operand
//     per channel:  Xs[(p,s,a)][b'][l] = G * blurred                  (S + box-sum + L + decimation, row gather)
//                   y^T[(p,s,a)][l'] = Xs * W^T                       (R + beta-sum, one GEMM, split-K)
//   adjoint  (spectroModel.py:173-185, spectroModelChannel.py:234-264): the transposes, in reverse.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/surfh_amd.h"
#include "dft_h2.h"
#include "dft_ct.h"
#include "gemm_f32.h"
#include "kernels.h"

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define LAUNCH_OK(expr)                                                                           \
    do {                                                                                          \
        int e_ = (expr);                                                                          \
        if (e_ != 0) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString((hipError_t)e_), __FILE__, __LINE__); \
    } while (0)

inline int pad64(int n) { return (n + 63) / 64 * 64; }

// host-side sparse rows: (source index, weight) lists + one destination per row
struct HostEll {
    std::vector<std::vector<std::pair<int64_t, float>>> rows;
    std::vector<int64_t> dst;
};

struct DevEll {
    EllTable t;
    int32_t *cnt = nullptr;
    int64_t *col = nullptr, *dst = nullptr;
    float *val = nullptr;
    uint32_t *rmw = nullptr;            // scatter tables: chunks of a row that need read-modify-write (EllTable::rmw)
    int2 *rng = nullptr, *g_rng = nullptr;   // ... or the exact wavelength ranges (EllTable::rng, GroupTable::rng)
    std::vector<int64_t> host_dst;      // kept for the scatter tables until the plan is complete
    // the same table with its rows grouped SCATTER_G at a time (GroupTable)
    GroupTable g;
    int32_t *g_cnt = nullptr;
    int64_t *g_col = nullptr, *g_dst = nullptr;
    float *g_val = nullptr;
    uint32_t *g_rmw = nullptr;
};

struct Channel {
    int ws0 = 0, ws1 = 0, Lin = 0, P = 0, S = 0, Ldet = 0, aout = 0, srf = 0, na = 0, nb = 0, alpha0 = 0, nas = 0,
        nbs = 0;
    int ws0a = 0;      // window start relative to the plan's first plane, rounded down to a multiple of 4
    int LinA = 0;      // planes from ws0a to the window end
    int LinP = 0;      // LinA padded to 64
    int shift = 0;     // (ws0 - lo) - ws0a
    int nlam = 0;      // LinA rounded up to 4: wavelengths the gather kernels process
    int K = 0, NP = 0, LdetP = 0, splitK = 1;
    long yoff = 0, ysize = 0;
    float *W = nullptr, *Wt = nullptr, *Xs = nullptr, *Cpart = nullptr, *ymat = nullptr;
    unsigned short *W16 = nullptr, *Wt16 = nullptr; // ... or into their two fp16 pieces [2][rows][cols] of W / sW (gemm_cc16.hip)
    unsigned short *Xs16 = nullptr, *ymat16 = nullptr;   // the data operands as fp16 pieces (all-consumer kernel, gemm_cc16.hip)
    float *bscale = nullptr;                        // Xs16's scales, one per (row, K segment): [nbs * ceil(LinP/1024)][NP]
    float sW = 1.f;
    // K-step classes of the two GEMMs (gemm_cc16.hip, build_klist below): per 256-row tile of W16 / Wt16 the steps that keep
    // all three products and the steps kept as h*h only; ksteps = (near, far) of the forward, (near, far) of the adjoint
    int *klF = nullptr, *klA = nullptr;
    int klFs = 0, klAs = 0;
    int permA = 0;                      // adjoint GEMM: a tile takes 256 / permA wavelengths of each of permA neighbouring beta columns (0: 256 consecutive rows)
    long ksteps[4] = {0, 0, 0, 0};
    unsigned *amax = nullptr;                       // [2][NP] max |row| of the data operands: Xs (forward), ymat (adjoint)
    unsigned *pmax = nullptr;                       // per-wave maxima of the kernel that wrote the operand (reduced into amax)
    DevEll fwd, adjT, adjRef;
    HostEll adjT_host;                  // kept until the grouped scatter table is built (plan creation)
    bool has_ref = false;
    bool bsum = false;   // no spectral blur: y[l][(p,s,a)] = sum over the slit's beta columns (MRSBlurred)
};

struct ProfRec {
    const char *name;
    hipEvent_t a, b;
};

}  // namespace

struct surfh_plan {
    int dev = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // second stream: the spectral-blur GEMMs run here while the gather / scatter of the neighbouring channel runs on
    // `stream` (different units: matrix cores vs L2 bandwidth, and the GEMM leaves registers for them on every SIMD)
    hipStream_t stream2 = nullptr;
    bool overlap = false;
    std::vector<hipEvent_t> sync_ev;             // dependency events between the two streams (no timing)
    size_t sync_next = 0;
    int Na = 0, Nb = 0, Lc = 0, T = 0, NAP = 0, NBP = 0, KAP = 0, KBP = 0;
    long PL = 0, PLc = 0;
    int lo = 0, hi = 0, Lown = 0, LP = 0;
    // owned cube planes = union of the channels' windows, stored compactly: segment = (first plane, length, compact offset)
    struct Seg { int start, len, coff; };
    std::vector<Seg> segs;
    std::vector<int> planes;   // compact index -> cube plane
    int compact(int l) const {
        for (auto &g : segs) if (l >= g.start && l <= g.start + g.len) return g.coff + (l - g.start);
        return -1;
    }
    float *sotf = nullptr, *tpl = nullptr, *mhat = nullptr, *spec = nullptr, *ycol = nullptr, *cube = nullptr,
          *maps_pad = nullptr, *ycol_maps = nullptr;
    float *Fi = nullptr, *Gi = nullptr, *Gf = nullptr, *Ff = nullptr, *GiT = nullptr, *GfT = nullptr;
    // folded-DFT matrices [MPx][KPx]: cos/sin along alpha; weighted cos/sin for c2r; plain cos/sin for r2c
    float *Cma = nullptr, *Sma = nullptr, *Gc = nullptr, *Gs = nullptr, *Cf = nullptr, *Sf = nullptr;
    int MPa = 0, KPa = 0, MPb = 0, KPb = 0;
    int n_cu = 256;
    bool gather_sorted = true;                   // gather rows ordered by cube location (L2 reuse across pointings)
    bool gemm_grouped = true;                    // the adjoint's spectral-blur GEMMs of up to four channels as one launch (SURFH_GEMM_GROUPED=0: one each)
    bool scatter_grouped = true;                 // adjoint scatter with SCATTER_G neighbouring pixels per workgroup (GroupTable)
    bool otf_prod = true;                        // plane-wise model: OTF products inside the loader of the inverse transform (SURFH_OTF_PROD=0: own kernels)
    bool gather_grouped = true;                  // forward gather (fp16 output) likewise
    bool dense_dft = false, fuse_mix = true, wblur_fp32 = false;
    // surfh_config.verify: every long sum accumulated in float64 (dense DFT products, spectral blur, adjoint spectral mix,
    // gather / scatter rows) -- the strict dot test; storage stays fp32
    bool verify = false;
    int prior_kind = 0;                          // 0: separated first differences (NpDiff_r / NpDiff_c); 1: joint Laplacian (surfh_set_prior)
    // two-piece fp16 passes with LDS-resident matrices (dft_h2.h): the plan's complex arrays (sotf, spec, ycol, and
    // mhat when T == 0) are then INTERLEAVED [..][LP][2] instead of planar [2][..][LP]
    bool h2 = false;
    unsigned short *h2img = nullptr;             // three images: (Cma, Sma), (Gc, Gs), (Cf, Sf)
    // Cooley-Tukey passes (dft_ct.h) for lengths whose folded matrix does not fit LDS (N = R * M: 501, 512, ...); same
    // interleaved layout.  ilv = h2 || ct is the layout flag of the complex arrays.
    bool ct = false, ilv = false;
    DftCtPlan ctA, ctB;                          // transform lengths Na / Nb (ctB aliases ctA when they are equal)
    // which kernel transforms an axis: the choice is per axis (a 300 x 64 image runs dft_ct along alpha and dft_h2 along beta);
    // h2 = both axes on dft_h2 (fused adjoint tail, OTF-support lists), ct = at least one axis on dft_ct.  An axis neither covers
    // (a prime factor above 190, fewer than 32 points) puts the plan on the dense fp32 products with planar arrays.
    int ax_a = 0, ax_b = 0;                      // 0: none, 1: dft_h2, 2: dft_ct
    // cube columns alpha in [a_lo, a_hi) hold every pixel any channel's tables touch: the transform passes that are batched
    // over alpha skip the rest (forward: the cube outside is never read; adjoint: it is zero).  ycol_adj: the adjoint's
    // intermediate in its own buffer, whose columns outside the range stay zero from plan creation on.
    int a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;  // (b: the same for the cube rows beta)
    float *ycol_adj = nullptr;
    // spectral-domain solver calls (surfh_normal_spec_dev ...): the maps' half spectra in Parseval-scaled form go in and out of the
    // transform passes directly.  Set for the duration of one call.
    const float *spec_in = nullptr;              // forward: the mix loader reads this instead of mhat
    float *spec_out = nullptr;                   // adjoint: the fused tail writes this instead of mhat
    const float *spec_prior_src = nullptr;       // adjoint: + spec_prior_mu * |D|^2 * this (the quadratic prior, world = 1)
    float spec_mu = 1.f, spec_prior_mu = 0.f;
    float *adjmix_part = nullptr;                // fused adjoint tail (dft_h2_adjmix_kernel): partial sums per (k_beta, slot); null: off
    // Support of the OTF (otf_support below): the (k_beta, chunk of 128 wavelengths) pairs -- super-tiles of the two passes that
    // touch the OTF, index k_beta * (LP / 128) + chunk -- in which some |sotf| exceeds 2^-24 of its plane's largest magnitude.
    // The forward's complex pass and the fused adjoint tail visit only these; otf_kbstart[kb] = first list position of kb.
    // ycol_mix: the forward's intermediate in its own buffer, whose other tiles stay zero from plan creation on.
    int *otf_vlist = nullptr, *otf_kbstart = nullptr;
    int otf_nvalid = 0;
    // per chunk of 128 wavelengths: k-steps of the forward's complex pass (k_alpha inside the support), k-steps of its pass along
    // beta (k_beta inside: the rest of ycol_mix is zero), rows k_beta the adjoint's first pass has to store: [3][LP / 128]
    int *otf_tabs = nullptr;
    float *ycol_mix = nullptr;
    int h2kA[3] = {0, 0, 0};
    float *io_x = nullptr, *io_y = nullptr, *io_cube = nullptr, *hth = nullptr, *mhat2 = nullptr;
    // accumulator of the exact adjoint: cleared ONCE at plan creation.  Every scatter row knows which of its wavelengths an
    // earlier channel has already written in the same pass (read-modify-write) and stores the others, so nothing stale
    // survives a pass and no per-call clear is needed (nullptr: the tables could not express that -- clear `cube` every call)
    float *gcube = nullptr;
    std::vector<Channel> ch;
    long isize = 0, osize = 0;
    // CG
    float *cg_x = nullptr, *cg_r = nullptr, *cg_d = nullptr, *cg_q = nullptr, *cg_b = nullptr, *cg_y = nullptr, *cg_qm = nullptr, *cg_dd = nullptr;
    double *dscal = nullptr, *dscratch = nullptr;   // [8] device scalars, [1024] partial sums
    double *cg_hist = nullptr;                     // device-resident r.r trace of the no-host-sync CG blocks (CG_HIST_CAP entries)
    int cg_hist_n = 0;
    // plane-wise CG with device-resident data (surfh_cg_planes_begin_dev / _step_dev): per-plane scalars [3][Lc], the caller's iterate
    double *pl_sc = nullptr;
    float *pl_x = nullptr;
    double pl_mu = 1.0, pl_mu_reg = 0.0;
    int pl_it = 0;
    // ... with its vectors in the cube's wavelength-innermost layout [NBP][NAP][LP] (no layout transpose inside an iteration):
    // x, r, d, q, b; per-wavelength scalars [3][LP] + partial sums; set while forward_dev / adjoint_dev are called on such vectors
    float *pn_v[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double *pn_sc = nullptr, *pn_part = nullptr;
    bool pn_native = false, pn_active = false, pn_fold_prior = false;
    // profiling
    bool prof = false;
    std::string prof_filter;                     // non-empty: only stages whose name starts with it are bracketed by events
    std::vector<ProfRec> pending;
    std::vector<hipEvent_t> pool;
    std::map<std::string, std::pair<long, double>> acc;
    std::vector<std::string> acc_names;
};

namespace {

struct Prof {
    surfh_plan *p;
    ProfRec r;
    bool on;
    hipStream_t st;
    Prof(surfh_plan *pl, const char *name, hipStream_t stream = nullptr) : p(pl), on(pl->prof), st(stream ? stream : pl->stream) {
        if (on && !pl->prof_filter.empty() && strncmp(name, pl->prof_filter.c_str(), pl->prof_filter.size()) != 0) on = false;
        if (!on) return;
        r.name = name;
        for (hipEvent_t *e : {&r.a, &r.b}) {
            if (!p->pool.empty()) {
                *e = p->pool.back();
                p->pool.pop_back();
            } else {
                hipEventCreate(e);
            }
        }
        hipEventRecord(r.a, st);
    }
    ~Prof() {
        if (!on) return;
        hipEventRecord(r.b, st);
        p->pending.push_back(r);
    }
};

// make stream `to` wait for the work enqueued so far on stream `from`
int chain(surfh_plan *p, hipStream_t from, hipStream_t to) {
    if (from == to) return 0;
    if (p->sync_next == p->sync_ev.size()) {
        hipEvent_t e;
        HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->sync_ev.push_back(e);
    }
    hipEvent_t e = p->sync_ev[p->sync_next++];
    if (p->sync_next >= 64) p->sync_next = 0;       // ring: an event is re-recorded long after its waiters were enqueued
    HIP_OK(hipEventRecord(e, from));
    HIP_OK(hipStreamWaitEvent(to, e, 0));
    return 0;
}

void prof_collect(surfh_plan *p) {
    if (p->pending.empty()) return;
    hipStreamSynchronize(p->stream);
    if (p->stream2) hipStreamSynchronize(p->stream2);
    for (auto &r : p->pending) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        auto &e = p->acc[r.name];
        e.first += 1;
        e.second += ms;
        p->pool.push_back(r.a);
        p->pool.push_back(r.b);
    }
    p->pending.clear();
    p->acc_names.clear();
    for (auto &kv : p->acc) p->acc_names.push_back(kv.first);
}

template <typename Tp>
int dev_alloc(Tp **p, size_t n) {
    HIP_OK(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(Tp)));
    return 0;
}

template <typename Tp>
int dev_upload(Tp **p, const std::vector<Tp> &h) {
    if (dev_alloc(p, h.size())) return 1;
    if (!h.empty()) HIP_OK(hipMemcpy(*p, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice));
    return 0;
}


int upload_ell(const HostEll &h, DevEll *d) {
    const int R = (int)h.rows.size();
    int W = 1;
    for (auto &r : h.rows) W = std::max(W, (int)r.size());
    std::vector<int32_t> cnt(R);
    std::vector<int64_t> col((size_t)R * W, 0);
    std::vector<float> val((size_t)R * W, 0.f);
    for (int r = 0; r < R; ++r) {
        cnt[r] = (int32_t)h.rows[r].size();
        for (int e = 0; e < cnt[r]; ++e) {
            col[(size_t)r * W + e] = h.rows[r][e].first;
            val[(size_t)r * W + e] = h.rows[r][e].second;
        }
    }
    if (dev_upload(&d->cnt, cnt) || dev_upload(&d->col, col) || dev_upload(&d->val, val) || dev_upload(&d->dst, h.dst))
        return 1;
    d->t.R = R;
    d->t.W = W;
    d->t.cnt = d->cnt;
    d->t.col = d->col;
    d->t.val = d->val;
    d->t.dst_off = d->dst;
    return 0;
}

void free_ell(DevEll *d) {
    hipFree(d->g_cnt);
    hipFree(d->g_col);
    hipFree(d->g_dst);
    hipFree(d->g_val);
    hipFree(d->g_rmw);
    hipFree(d->rmw);
    hipFree(d->rng);
    hipFree(d->g_rng);
    hipFree(d->cnt);
    hipFree(d->col);
    hipFree(d->val);
    hipFree(d->dst);
}

// rows [r, r + n) of h taken as one group: union of their taps with one weight per member
int upload_groups(const HostEll &h, const std::vector<std::pair<size_t, size_t>> &runs, const std::vector<uint32_t> *mask, DevEll *d,
                  const int G, const std::vector<int2> *ranges = nullptr) {
    const size_t NG = runs.size();
    std::vector<std::vector<std::pair<int64_t, std::array<float, GROUP_MAX>>>> grows(NG);
    std::vector<int64_t> gdst(NG * G, -1);
    std::vector<uint32_t> grmw(NG * G, 0u);
    std::vector<int2> grng(ranges ? NG * G : 0, make_int2(0, 0));
    int W = 1;
    for (size_t gi = 0; gi < NG; ++gi) {
        const size_t r = runs[gi].first, n = runs[gi].second;
        std::map<int64_t, std::array<float, GROUP_MAX>> u;
        for (size_t m = 0; m < n; ++m) {
            for (auto &e : h.rows[r + m]) {
                auto it = u.find(e.first);
                if (it == u.end()) it = u.emplace(e.first, std::array<float, GROUP_MAX>{}).first;
                it->second[m] += e.second;
            }
            gdst[gi * G + m] = h.dst[r + m];
            if (mask) grmw[gi * G + m] = (*mask)[r + m];
            if (ranges) grng[gi * G + m] = (*ranges)[r + m];
        }
        grows[gi].assign(u.begin(), u.end());
        W = std::max(W, (int)grows[gi].size());
    }
    if (const char *es = getenv("SURFH_TABLE_STATS"); es && es[0] == '1') {      // diagnostics: taps per group against taps of its members
        size_t ut = 0, mt = 0;
        for (size_t gi = 0; gi < NG; ++gi) {
            ut += grows[gi].size();
            for (size_t m = 0; m < runs[gi].second; ++m) mt += h.rows[runs[gi].first + m].size();
        }
        fprintf(stderr, "[surfh tables] %zu groups of <= %d rows, widest %d taps, %.2f union taps per group, %.2f taps per member row\n", NG,
                G, W, (double)ut / std::max<size_t>(NG, 1), (double)mt / std::max<size_t>(h.rows.size(), 1));
    }
    std::vector<int32_t> gcnt(NG);
    std::vector<int64_t> gcol(NG * W, 0);
    std::vector<float> gval(NG * W * G, 0.f);
    for (size_t gi = 0; gi < NG; ++gi) {
        gcnt[gi] = (int32_t)grows[gi].size();
        for (size_t e = 0; e < grows[gi].size(); ++e) {
            gcol[gi * W + e] = grows[gi][e].first;
            for (int m = 0; m < G; ++m) gval[(gi * W + e) * G + m] = grows[gi][e].second[m];
        }
    }
    if (dev_upload(&d->g_cnt, gcnt) || dev_upload(&d->g_col, gcol) || dev_upload(&d->g_val, gval) || dev_upload(&d->g_dst, gdst) ||
        dev_upload(&d->g_rmw, grmw))
        return 1;
    d->g.NG = (int)NG; d->g.W = W; d->g.G = G; d->g.cnt = d->g_cnt; d->g.col = d->g_col; d->g.val = d->g_val; d->g.dst = d->g_dst; d->g.rmw = d->g_rmw;
    if (ranges) {
        if (dev_upload(&d->g_rng, grng)) return 1;
        d->g.rng = d->g_rng;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// table construction for one channel
// ---------------------------------------------------------------------------------------------
int build_channel(surfh_plan *p, const surfh_channel_desc &d, Channel *c) {
    c->ws0 = d.wslice_start;
    c->ws1 = d.wslice_stop;
    c->Lin = c->ws1 - c->ws0;
    c->P = d.n_pointings;
    c->S = d.n_slit;
    c->Ldet = d.n_lambda_out;
    c->aout = d.n_alpha_out;
    c->srf = d.srf;
    c->na = d.na;
    c->nb = d.nb;
    c->alpha0 = d.alpha0;
    c->nas = d.n_alpha_slit;
    c->nbs = d.n_beta_slit;
    if (c->Lin <= 0 || c->ws0 < 0 || c->ws1 > p->Lc) return fail("channel wslice (%d,%d) outside cube (Lc=%d)", c->ws0, c->ws1, p->Lc);
    if (c->P < 1 || c->S < 1 || c->Ldet < 1 || c->aout < 1 || c->srf < 1 || c->nbs < 1) return fail("bad channel dims");
    const int box = d.box_len > 0 ? d.box_len : c->srf, bsh = d.box_len > 0 ? d.box_shift : 0;
    if (box > c->na || bsh <= -c->na || bsh >= c->na) return fail("bad box window (len %d, shift %d)", box, bsh);
    if ((c->aout - 1) * c->srf >= c->nas) return fail("decimation (alpha_out-1)*srf=%d exceeds the slit alpha window %d", (c->aout - 1) * c->srf, c->nas);
    if (c->alpha0 < 0 || c->alpha0 + c->nas > c->na) return fail("slit alpha window outside the local grid");
    if (!d.slit_beta0 || !d.slit_weights || !d.grid_i0 || !d.grid_i1 || !d.grid_y0 || !d.grid_y1)
        return fail("channel table pointer is NULL");
    c->bsum = (d.wpsf == nullptr);
    if (c->bsum) c->Ldet = c->Lin;
    for (int s = 0; s < c->S; ++s)
        if (d.slit_beta0[s] < 0 || d.slit_beta0[s] + c->nbs > c->nb) return fail("slit %d beta window outside the local grid", s);
    // wavelength window inside the plan's planes, start aligned to 4 floats for 16-byte vector access
    const int cw0 = p->compact(c->ws0);
    if (cw0 < 0) return fail("channel window not inside the plan's planes");
    c->ws0a = (cw0 / 4) * 4;
    c->shift = cw0 - c->ws0a;
    c->LinA = (cw0 + c->Lin) - c->ws0a;
    c->LinP = pad64(c->LinA);
    if (!c->bsum && ((long)c->nbs * c->LinP) % 128) c->LinP += 64;    // K, NP, LdetP multiples of 128: tile grid of the GEMMs
    c->nlam = (c->LinA + 3) / 4 * 4;
    c->K = (c->bsum ? 1 : c->nbs) * c->LinP;
    c->NP = (c->P * c->S * c->aout + 127) / 128 * 128;
    c->LdetP = (c->Ldet + 127) / 128 * 128;
    c->ysize = (long)c->P * c->S * c->Ldet * c->aout;

    const long nloc = (long)c->na * c->nb;
    // bounds: the forward gather mirrors bounds_error=True (cython_2D_interpolation.py:472-478);
    // indices come clamped from find_indices, so only range-check them here.
    for (long i = 0; i < (long)c->P * nloc; ++i)
        if (d.grid_i0[i] < 0 || d.grid_i0[i] > p->Na - 2 || d.grid_i1[i] < 0 || d.grid_i1[i] > p->Nb - 2)
            return fail("bilinear index out of range at local pixel %ld", i);

    const int64_t LP = p->LP;
    auto pix_off = [&](int ia, int ib) -> int64_t { return ((int64_t)ib * p->NAP + ia) * LP + c->ws0a; };
    auto xs_off = [&](int pt, int s, int a, int b) -> int64_t {
        return ((int64_t)((pt * c->S + s) * c->aout + a)) * c->K + (c->bsum ? 0 : (int64_t)b * c->LinP);
    };

    // ---- forward rows (p, a, j-order over (s,b')): S + box-sum + slit window + decimation -------
    HostEll f;
    std::vector<std::vector<std::pair<int, int>>> colslit(c->nb);   // local column -> (slit, b')
    for (int s = 0; s < c->S; ++s)
        for (int b = 0; b < c->nbs; ++b) colslit[d.slit_beta0[s] + b].push_back({s, b});
    for (int pt = 0; pt < c->P; ++pt)
        for (int a = 0; a < c->aout; ++a)
            for (int j = 0; j < c->nb; ++j)
                for (auto &sb : colslit[j]) {
                    const int s = sb.first, b = sb.second;
                    const double ws = d.slit_weights[(long)s * c->nbs + b];
                    std::vector<std::pair<int64_t, float>> row;
                    std::map<int64_t, double> acc;
                    for (int r = 0; r < box; ++r) {
                        const int i = (c->alpha0 + a * c->srf + r + bsh + c->na) % c->na;
                        const long li = (long)pt * nloc + (long)i * c->nb + j;
                        const int i0 = d.grid_i0[li], i1 = d.grid_i1[li];
                        const double y0 = d.grid_y0[li], y1 = d.grid_y1[li];
                        const double w[4] = {(1. - y0) * (1. - y1), (1. - y0) * y1, y0 * (1. - y1), y0 * y1};
                        const int da[4] = {0, 0, 1, 1}, db[4] = {0, 1, 0, 1};
                        for (int k = 0; k < 4; ++k)
                            if (w[k] * ws != 0.0) acc[pix_off(i0 + da[k], i1 + db[k])] += w[k] * ws;
                    }
                    // consecutive samples of the box window share two of their four cube pixels: one tap per distinct
                    // pixel (about 16 instead of 28 reads per output element)
                    for (auto &e : acc) row.push_back({e.first, (float)e.second});
                    f.rows.push_back(std::move(row));
                    f.dst.push_back(xs_off(pt, s, a, b));
                }
    if (c->bsum) {   // rows that share a destination (the slit's beta columns) are merged into one
        std::map<int64_t, std::map<int64_t, double>> mg;
        for (size_t r = 0; r < f.rows.size(); ++r)
            for (auto &e : f.rows[r]) mg[f.dst[r]][e.first] += (double)e.second;
        HostEll f2;
        for (auto &row : mg) {
            std::vector<std::pair<int64_t, float>> v;
            for (auto &e : row.second) v.push_back({e.first, (float)e.second});
            f2.rows.push_back(std::move(v));
            f2.dst.push_back(row.first);
        }
        f = std::move(f2);
    }
    if (p->gather_sorted) {
        // Order the gather rows by the cube pixel of their first tap.  In (pointing, alpha, beta) order the four dither
        // pointings, which cover the same pixels shifted by a few columns, are a quarter of the table apart: every pixel
        // was fetched from HBM once per pointing (measured 0.525 GB per launch for a 0.15 GB window).  Sorted, the rows
        // that share taps run together on one XCD and hit its L2.
        std::vector<size_t> ord(f.rows.size());
        for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
        std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) {
            const int64_t ka = f.rows[a].empty() ? INT64_MAX : f.rows[a][0].first, kb = f.rows[b].empty() ? INT64_MAX : f.rows[b][0].first;
            return ka < kb;
        });
        HostEll g;
        g.rows.reserve(ord.size());
        g.dst.reserve(ord.size());
        for (size_t i : ord) {
            g.rows.push_back(std::move(f.rows[i]));
            g.dst.push_back(f.dst[i]);
        }
        f = std::move(g);
    }
    if (upload_ell(f, &c->fwd)) return 1;
    if (p->gather_grouped && p->gather_sorted && !c->bsum) {      // GATHER_G rows neighbouring in cube-location order per workgroup
        std::vector<std::pair<size_t, size_t>> runs;
        for (size_t r = 0; r < f.rows.size(); r += GATHER_G) runs.push_back({r, std::min<size_t>(GATHER_G, f.rows.size() - r)});
        if (upload_groups(f, runs, nullptr, &c->fwd, GATHER_G)) return 1;
    }

    // ---- exact transpose: rows = touched cube pixels ------------------------------------------
    {
        std::map<int64_t, std::map<int64_t, double>> tr;   // pixel offset -> (Xs offset -> weight)
        for (size_t r = 0; r < f.rows.size(); ++r)
            for (auto &e : f.rows[r]) tr[e.first][f.dst[r]] += (double)e.second;
        HostEll t;
        for (auto &px : tr) {
            std::vector<std::pair<int64_t, float>> row;
            for (auto &e : px.second) row.push_back({e.first, (float)e.second});
            t.rows.push_back(std::move(row));
            t.dst.push_back(px.first);
        }
        if (upload_ell(t, &c->adjT)) return 1;
        for (int64_t o : t.dst) {
            const int ia = (int)((o / LP) % p->NAP), ib = (int)((o / LP) / p->NAP);
            p->a_lo = std::min(p->a_lo, ia); p->a_hi = std::max(p->a_hi, ia + 1);
            p->b_lo = std::min(p->b_lo, ib); p->b_hi = std::max(p->b_hi, ib + 1);
        }
        c->adjT.host_dst = t.dst;
        c->adjT_host = std::move(t);
    }

    // ---- reference-compatible back-interpolation (gridding_t) ---------------------------------
    c->has_ref = d.gt_i0 && d.gt_i1 && d.gt_y0 && d.gt_y1 && d.gt_inside;
    if (c->has_ref) {
        // local row i' -> decimated rows a whose box window contains it
        std::vector<std::vector<int>> arow(c->na);
        for (int a = 0; a < c->aout; ++a)
            for (int r = 0; r < box; ++r) arow[(c->alpha0 + a * c->srf + r + bsh + c->na) % c->na].push_back(a);
        HostEll t;
        const long npix = (long)p->Na * p->Nb;
        for (int ib = 0; ib < p->Nb; ++ib)
            for (int ia = 0; ia < p->Na; ++ia) {
                std::map<int64_t, double> m;
                for (int pt = 0; pt < c->P; ++pt) {
                    const long gi = (long)pt * npix + (long)ia * p->Nb + ib;
                    if (!d.gt_inside[gi]) continue;
                    const int i0 = d.gt_i0[gi], i1 = d.gt_i1[gi];
                    if (i0 < 0 || i0 > c->na - 2 || i1 < 0 || i1 > c->nb - 2) return fail("gridding_t index out of range");
                    const double y0 = d.gt_y0[gi], y1 = d.gt_y1[gi];
                    const double w[4] = {(1. - y0) * (1. - y1), (1. - y0) * y1, y0 * (1. - y1), y0 * y1};
                    const int li[4] = {i0, i0, i0 + 1, i0 + 1}, lj[4] = {i1, i1 + 1, i1, i1 + 1};
                    for (int k = 0; k < 4; ++k)
                        if (w[k] != 0.0)
                        for (int a : arow[li[k]])
                            for (auto &sb : colslit[lj[k]])
                                m[xs_off(pt, sb.first, a, sb.second)] += w[k] * d.slit_weights[(long)sb.first * c->nbs + sb.second];
                }
                if (m.empty()) continue;
                std::vector<std::pair<int64_t, float>> row;
                for (auto &e : m) row.push_back({e.first, (float)e.second});
                t.rows.push_back(std::move(row));
                t.dst.push_back(pix_off(ia, ib));
            }
        if (upload_ell(t, &c->adjRef)) return 1;
        for (int64_t o : t.dst) {
            const int ia = (int)((o / LP) % p->NAP), ib = (int)((o / LP) / p->NAP);
            p->a_lo = std::min(p->a_lo, ia); p->a_hi = std::max(p->a_hi, ia + 1);
            p->b_lo = std::min(p->b_lo, ib); p->b_hi = std::max(p->b_hi, ib + 1);
        }
    }

    // ---- spectral PSF as GEMM operands: W[l'][b'*LinP + shift + l] = wpsf[l'][l][b'] -------------
    if (!c->bsum) {
        std::vector<float> W((size_t)c->LdetP * c->K, 0.f), Wt((size_t)c->K * c->LdetP, 0.f);
        for (int l = 0; l < c->Ldet; ++l)
            for (int lam = 0; lam < c->Lin; ++lam)
                for (int b = 0; b < c->nbs; ++b) {
                    const float v = (float)d.wpsf[((long)l * c->Lin + lam) * c->nbs + b];
                    const size_t k = (size_t)b * c->LinP + c->shift + lam;
                    W[(size_t)l * c->K + k] = v;
                    Wt[k * c->LdetP + l] = v;
                }
        float wmax = 0.f;
        for (float v : W) wmax = std::max(wmax, std::fabs(v));
        c->sW = gemm_f16x2_scale(wmax);
        if (dev_upload(&c->W, W) || dev_upload(&c->Wt, Wt)) return 1;
    }
    if (dev_alloc(&c->Xs, (size_t)c->NP * c->K)) return 1;
    HIP_OK(hipMemset(c->Xs, 0, (size_t)c->NP * c->K * sizeof(float)));
    if (!c->bsum) {
        if (dev_alloc(&c->ymat, (size_t)c->NP * c->LdetP)) return 1;
        HIP_OK(hipMemset(c->ymat, 0, (size_t)c->NP * c->LdetP * sizeof(float)));
    }
    return 0;
}

// K-step classes of the two-piece fp16 GEMM for its constant operand B [N][ldb] (host copy; K columns, tiles of 256 rows).
// A step of 32 columns may be computed from the leading fp16 pieces alone ("far": relative error of its terms <= 2^-10, random
// sign) when what it contributes is small: per tile the steps are taken in ascending order of their largest share of a row,
// and moved to the far class as long as, for EVERY row of the tile, the far steps together hold <= tol1 of the row's l1 norm
// and <= tol2 of its l2 norm.  The error this adds to an output is then <= 2^-10 tol1 of sum |B||x| in the worst case (every
// rounding error aligned) and ~ 3e-4 tol2 of |B row|_2 |x|_2 for rounding errors of random sign: with tol1 = 2^-8, tol2 = 2^-10
// 4e-6 and 3e-7 of the row's own scale.  The spectral response (a grating's sinc^2, instru.py psfs) falls off as 1 / x^2 from
// its diagonal: about two thirds of the steps of a tile qualify.  Record per tile: [n_near, n_far, near..., far...], entry =
// step | segment << 16 (gemm_f32.h).  Fewer than 8 far steps are not worth the second pass: all near.
int build_klist(const float *B, int N, int K, long ldb, int segLinP, int segChunks, double tol1, double tol2, std::vector<int> *out,
                int *stride, long *n_near, long *n_far, int permP = 0, int permLin = 0) {
    // tile columns as in the kernel: 256 consecutive rows, or 256 / permP rows of each of permP neighbouring columns of permLin rows
    const int Q = permP ? 256 / permP : 256, tilesL = permP ? permLin / Q : 0, ncol = permP ? N / permLin : 0;
    const int nb = K / 32, tilesN = permP ? (ncol + permP - 1) / permP * tilesL : (N + 255) / 256;
    auto brow = [&](int tn, int v) {
        if (!permP) { const int n = tn * 256 + v; return n < N ? n : -1; }
        const int c = (tn / tilesL) * permP + v / Q;
        return c < ncol ? c * permLin + (tn % tilesL) * Q + v % Q : -1;
    };
    *stride = 2 + nb;
    out->assign((size_t)tilesN * *stride, 0);
    *n_near = *n_far = 0;
    std::vector<double> l1((size_t)256 * nb), l2((size_t)256 * nb), L1(256), L2(256), c1(256), c2(256), imp(nb);
    std::vector<int> order(nb);
    std::vector<char> far(nb);
    for (int tn = 0; tn < tilesN; ++tn) {
        const int nr = 256;
        for (int r = 0; r < nr; ++r) {
            const int br = brow(tn, r);
            if (br < 0) {          // no such row: nothing to bound
                for (int b = 0; b < nb; ++b) l1[(size_t)r * nb + b] = l2[(size_t)r * nb + b] = 0.0;
                L1[r] = L2[r] = 0.0;
                continue;
            }
            const float *row = B + (long)br * ldb;
            double s1 = 0.0, s2 = 0.0;
            for (int b = 0; b < nb; ++b) {
                double a1 = 0.0, a2 = 0.0;
                for (int k = 0; k < 32; ++k) { const double v = row[b * 32 + k]; a1 += std::fabs(v); a2 += v * v; }
                l1[(size_t)r * nb + b] = a1; l2[(size_t)r * nb + b] = a2;
                s1 += a1; s2 += a2;
            }
            L1[r] = s1; L2[r] = s2;
        }
        for (int b = 0; b < nb; ++b) {
            double m = 0.0;
            for (int r = 0; r < nr; ++r)
                if (L1[r] > 0.0) m = std::max(m, std::max(l1[(size_t)r * nb + b] / (L1[r] * tol1), std::sqrt(l2[(size_t)r * nb + b] / L2[r]) / tol2));
            imp[b] = m;
            order[b] = b;
            far[b] = 0;
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return imp[a] < imp[b]; });
        std::fill(c1.begin(), c1.end(), 0.0);
        std::fill(c2.begin(), c2.end(), 0.0);
        int nf = 0;
        for (int i = 0; i < nb; ++i) {
            const int b = order[i];
            bool ok = true;
            for (int r = 0; r < nr && ok; ++r)
                ok = c1[r] + l1[(size_t)r * nb + b] <= tol1 * L1[r] && c2[r] + l2[(size_t)r * nb + b] <= tol2 * tol2 * L2[r];
            if (!ok) break;
            for (int r = 0; r < nr; ++r) { c1[r] += l1[(size_t)r * nb + b]; c2[r] += l2[(size_t)r * nb + b]; }
            far[b] = 1;
            ++nf;
        }
        if (nf < 8) { std::fill(far.begin(), far.end(), 0); nf = 0; }
        int *rec = out->data() + (size_t)tn * *stride;
        rec[0] = nb - nf; rec[1] = nf;
        int in = 2, ifar = 2 + nb - nf;
        for (int b = 0; b < nb; ++b) {
            const int k = b * 32;
            const int e = b | ((segLinP ? (k / segLinP) * segChunks + (k % segLinP) / 1024 : 0) << 16);
            if (far[b]) rec[ifar++] = e; else rec[in++] = e;
        }
        *n_near += nb - nf; *n_far += nf;
    }
    return 0;
}

int pick_split(const Channel &c, int forced, bool f16, int n_cu) {
    if (forced > 0) return (c.K % (32 * forced) == 0) ? forced : 1;
    if (f16) {
        // two-piece fp16 kernel (256 x 256 tiles, one workgroup per CU).  The slab length is set by accuracy first: one fp32
        // accumulation chain of 4096 non-negative products shows a bias of -1.3e-7 (three-piece bf16 products lost their
        // small terms beyond about 1024 k), so chains run up to 4352 k (136 K steps); beyond the fewest such slabs, any
        // divisor that leaves at least 16 K steps per slab and fills the last round of workgroups best.
        constexpr int max_steps = 136, tile_m = 256, tile_n = 256;
        const long tiles = (long)((c.NP + tile_m - 1) / tile_m) * ((c.LdetP + tile_n - 1) / tile_n);
        const int steps = c.K / 32;
        int smin = 0;
        for (int s = 1; s <= steps; ++s)
            if (steps % s == 0 && steps / s <= max_steps) { smin = s; break; }
        if (!smin) return 1;
        int best = smin;
        double best_t = -1.0;
        for (int s = smin; s <= steps / 16 && s <= steps; ++s) {
            if (steps % s) continue;
            const double t = (double)((tiles * s + n_cu - 1) / n_cu) * (steps / s) + 3.0 * s;
            if (best_t < 0 || t < best_t) { best_t = t; best = s; }
        }
        return best;
    }
    const int bm = (c.NP % 128 == 0) ? 128 : 64, bn = (c.LdetP % 128 == 0) ? 128 : 64;
    const long tiles = (long)(c.NP / bm) * (c.LdetP / bn);
    int best = 1;
    for (int s : {1, 2, 3, 4, 6, 8, 12, 16, 24, 32}) {
        if (c.K % (32 * s)) continue;
        if (c.K / s < 256) break;
        best = s;
        if (tiles * s >= 768) break;
    }
    return best;
}

// ---------------------------------------------------------------------------------------------
// DFT matrices (ortho).  Forward r2c along beta then c2c along alpha; inverse c2c along alpha then
// c2r along beta with Hermitian weights w_k (1 for k=0 and Nyquist, else 2) -- numpy's rfft2/irfft2.
// ---------------------------------------------------------------------------------------------
void build_dft(const surfh_plan *p, std::vector<float> &Fi, std::vector<float> &Gi, std::vector<float> &Gf,
               std::vector<float> &Ff, std::vector<float> &GiT, std::vector<float> &GfT) {
    const int Na = p->Na, Nb = p->Nb, NAP = p->NAP, NBP = p->NBP, KAP = p->KAP, KBP = p->KBP;
    const int nkb = Nb / 2 + 1;
    const double sa = 1.0 / std::sqrt((double)Na), sb = 1.0 / std::sqrt((double)Nb);
    Fi.assign((size_t)2 * NAP * 2 * KAP, 0.f);
    Ff.assign((size_t)2 * KAP * 2 * NAP, 0.f);
    Gi.assign((size_t)2 * KBP * NBP, 0.f);
    Gf.assign((size_t)NBP * 2 * KBP, 0.f);
    GiT.assign((size_t)NBP * 2 * KBP, 0.f);
    GfT.assign((size_t)2 * KBP * NBP, 0.f);
    for (int a = 0; a < Na; ++a)
        for (int k = 0; k < Na; ++k) {
            const long m = ((long)a * k) % Na;   // exact phase reduction
            const double th = 2.0 * M_PI * (double)m / (double)Na;
            const double c = std::cos(th) * sa, s = std::sin(th) * sa;
            // inverse along alpha: rows (c,alpha), cols (c',k_alpha)
            Fi[((size_t)0 * NAP + a) * (2 * KAP) + 0 * KAP + k] = (float)c;
            Fi[((size_t)0 * NAP + a) * (2 * KAP) + 1 * KAP + k] = (float)(-s);
            Fi[((size_t)1 * NAP + a) * (2 * KAP) + 0 * KAP + k] = (float)s;
            Fi[((size_t)1 * NAP + a) * (2 * KAP) + 1 * KAP + k] = (float)c;
            // forward along alpha: rows (c,k_alpha), cols (c',alpha)
            Ff[((size_t)0 * KAP + k) * (2 * NAP) + 0 * NAP + a] = (float)c;
            Ff[((size_t)0 * KAP + k) * (2 * NAP) + 1 * NAP + a] = (float)s;
            Ff[((size_t)1 * KAP + k) * (2 * NAP) + 0 * NAP + a] = (float)(-s);
            Ff[((size_t)1 * KAP + k) * (2 * NAP) + 1 * NAP + a] = (float)c;
        }
    for (int b = 0; b < Nb; ++b)
        for (int k = 0; k < nkb; ++k) {
            const long m = ((long)b * k) % Nb;
            const double th = 2.0 * M_PI * (double)m / (double)Nb;
            const double c = std::cos(th) * sb, s = std::sin(th) * sb;
            const double w = (k == 0 || (Nb % 2 == 0 && k == Nb / 2)) ? 1.0 : 2.0;
            Gi[((size_t)0 * KBP + k) * NBP + b] = (float)(w * c);
            Gi[((size_t)1 * KBP + k) * NBP + b] = (float)(-w * s);
            Gf[(size_t)b * (2 * KBP) + 0 * KBP + k] = (float)c;
            Gf[(size_t)b * (2 * KBP) + 1 * KBP + k] = (float)(-s);
            GiT[(size_t)b * (2 * KBP) + 0 * KBP + k] = (float)(w * c);
            GiT[(size_t)b * (2 * KBP) + 1 * KBP + k] = (float)(-w * s);
            GfT[((size_t)0 * KBP + k) * NBP + b] = (float)c;
            GfT[((size_t)1 * KBP + k) * NBP + b] = (float)(-s);
        }
}

// q += mu_reg * (the plan's regulariser) d, per image of n_img
int prior_add(surfh_plan *p, hipStream_t st, const float *d, float *q, int n_img, float mu_reg) {
    return p->prior_kind == 1 ? launch_prior_joint_add(st, d, q, n_img, p->Na, p->Nb, mu_reg) : launch_prior_add(st, d, q, n_img, p->Na, p->Nb, mu_reg);
}

// fp32-MFMA GEMM, or its float64-accumulating twin in verification mode
int gemm32(surfh_plan *p, hipStream_t st, const GemmArgs &g) { return p->verify ? launch_gemm_f64acc(st, g) : launch_gemm_f32(st, g); }

// ---- plane-major 2-D transforms (only for the T abundance maps) ---------------------------------
// real [B][NAP][NBP] -> spec [B][2][KAP][KBP]   (tmp = ycol_maps viewed as [B][NAP][2*KBP])
int rfft2_planes(surfh_plan *p, const float *src, float *dst, int B) {
    GemmArgs g;
    g.A0 = src; g.lda = p->NBP; g.sA = p->PLc;
    g.B0 = p->Gf; g.ldb = 2 * p->KBP; g.sB = 0;
    g.C = p->ycol_maps; g.ldc = 2 * p->KBP; g.sC = (long)p->NAP * 2 * p->KBP;
    g.M = p->NAP; g.N = 2 * p->KBP; g.K = p->NBP; g.batch = B;
    {
        Prof pr(p, "gemm_dft_rows_fwd_maps");
        LAUNCH_OK(gemm32(p, p->stream, g));
    }
    GemmArgs h;
    h.A0 = p->Ff; h.lda = 2 * p->NAP; h.sA = 0;
    h.B0 = p->ycol_maps; h.B1 = p->ycol_maps + p->KBP; h.ksplitB = p->NAP; h.ldb = 2 * p->KBP;
    h.sB = (long)p->NAP * 2 * p->KBP;
    h.C = dst; h.ldc = p->KBP; h.sC = 2 * p->PL;
    h.M = 2 * p->KAP; h.N = p->KBP; h.K = 2 * p->NAP; h.batch = B;
    {
        Prof pr(p, "gemm_dft_cols_fwd_maps");
        LAUNCH_OK(gemm32(p, p->stream, h));
    }
    return 0;
}

// spec [B][2][KAP][KBP] -> real [B][NAP][NBP]   (tmp = ycol_maps viewed as [B][2][NAP][KBP])
int irfft2_planes(surfh_plan *p, const float *src, float *dst, int B) {
    GemmArgs g;
    g.A0 = p->Fi; g.lda = 2 * p->KAP; g.sA = 0;
    g.B0 = src; g.ldb = p->KBP; g.sB = 2 * p->PL;
    g.C = p->ycol_maps; g.ldc = p->KBP; g.sC = (long)2 * p->NAP * p->KBP;
    g.M = 2 * p->NAP; g.N = p->KBP; g.K = 2 * p->KAP; g.batch = B;
    {
        Prof pr(p, "gemm_dft_cols_inv_maps");
        LAUNCH_OK(gemm32(p, p->stream, g));
    }
    GemmArgs h;
    h.A0 = p->ycol_maps; h.A1 = p->ycol_maps + (long)p->NAP * p->KBP; h.ksplitA = p->KBP; h.lda = p->KBP;
    h.sA = (long)2 * p->NAP * p->KBP;
    h.B0 = p->Gi; h.ldb = p->NBP; h.sB = 0;
    h.C = dst; h.ldc = p->NBP; h.sC = p->PLc;
    h.M = p->NAP; h.N = p->NBP; h.K = 2 * p->KBP; h.batch = B;
    {
        Prof pr(p, "gemm_dft_rows_inv_maps");
        LAUNCH_OK(gemm32(p, p->stream, h));
    }
    return 0;
}

// ---- wavelength-innermost 2-D transforms of the whole owned cube -------------------------------
// cube [NBP][NAP][LP] -> spec [2][KAP][KBP][LP]        (tmp ycol viewed as Z[2][KBP][NAP][LP])
int rfft2_lam(surfh_plan *p, const float *src, float *dst) {
    const long LP = p->LP;
    GemmArgs g;   // Z[(c,kb)][(a,l)] = GfT[(c,kb)][b] * cube[b][(a,l)]
    g.A0 = p->GfT; g.lda = p->NBP;
    g.B0 = src; g.ldb = p->NAP * LP;
    g.C = p->ycol; g.ldc = p->NAP * LP;
    g.M = 2 * p->KBP; g.N = (int)(p->NAP * LP); g.K = p->NBP;
    {
        Prof pr(p, "gemm_dft_rows_fwd");
        LAUNCH_OK(gemm32(p, p->stream, g));
    }
    GemmArgs h;   // per kb: S[(c,ka)][l] = Ff[(c,ka)][(c',a)] * Z[c'][kb][a][l]
    h.A0 = p->Ff; h.lda = 2 * p->NAP;
    h.B0 = p->ycol; h.B1 = p->ycol + (long)p->KBP * p->NAP * LP; h.ksplitB = p->NAP; h.ldb = LP; h.sB = p->NAP * LP;
    h.C = dst; h.ldc = p->KBP * LP; h.sC = LP;
    h.M = 2 * p->KAP; h.N = (int)LP; h.K = 2 * p->NAP; h.batch = p->KBP;
    {
        Prof pr(p, "gemm_dft_cols_fwd");
        LAUNCH_OK(gemm32(p, p->stream, h));
    }
    return 0;
}

// spec [2][KAP][KBP][LP] -> cube [NBP][NAP][LP]        (tmp ycol viewed as Y[2][NAP][KBP][LP])
int irfft2_lam(surfh_plan *p, const float *src, float *dst) {
    const long LP = p->LP;
    GemmArgs g;   // Y[(c,a)][(kb,l)] = Fi[(c,a)][(c',ka)] * S[(c',ka)][(kb,l)]
    g.A0 = p->Fi; g.lda = 2 * p->KAP;
    g.B0 = src; g.ldb = p->KBP * LP;
    g.C = p->ycol; g.ldc = p->KBP * LP;
    g.M = 2 * p->NAP; g.N = (int)(p->KBP * LP); g.K = 2 * p->KAP;
    {
        Prof pr(p, "gemm_dft_cols_inv");
        LAUNCH_OK(gemm32(p, p->stream, g));
    }
    GemmArgs h;   // per a: cube[b][a][l] = GiT[b][(c,kb)] * Y[c][a][kb][l]
    h.A0 = p->GiT; h.lda = 2 * p->KBP;
    h.B0 = p->ycol; h.B1 = p->ycol + (long)p->NAP * p->KBP * LP; h.ksplitB = p->KBP; h.ldb = LP; h.sB = p->KBP * LP;
    h.C = dst; h.ldc = p->NAP * LP; h.sC = LP;
    h.M = p->NBP; h.N = (int)LP; h.K = 2 * p->KBP; h.batch = p->NAP;
    {
        Prof pr(p, "gemm_dft_rows_inv");
        LAUNCH_OK(gemm32(p, p->stream, h));
    }
    return 0;
}

// ---- two-piece fp16 passes, matrices resident in LDS, interleaved complex arrays (dft_h2.h) --------
// cube [NBP][NAP][LP] -> spec [KAP][KBP][LP][2]        (tmp ycol viewed as Z[KBP][NAP][LP][2])
// `madj` != nullptr: the second pass does not store the spectrum but multiplies it by conj(sotf) and reduces it over the
// wavelengths with the template weights straight into madj [T][2][KAP][KBP] (the adjoint's tail, spectroModel.py:175-181)
// `acols`: the source cube is zero outside the alpha range [a_lo, a_hi) (the adjoint's accumulator): the first pass
// transforms only those columns, into ycol_adj whose other columns are zero for good
// Support of the OTF for the two passes that multiply by it (the forward's complex pass with the fused mix, the fused adjoint
// tail).  A PSF sampled finer than its diffraction limit has an OTF that vanishes beyond a cutoff; the reference's synthetic
// Gaussian PSF (utils.py:40-50) falls below 2^-24 of its peak beyond 50-80 % of the k_beta range.  Products with such entries
// are below the rounding of the plane's leading terms in fp32, so the (k_beta, 128-wavelength chunk) super-tiles in which NO
// entry of any k_alpha reaches 2^-24 of its plane's largest magnitude are dropped from both passes -- the same set in both, so the
// adjoint stays the transpose of the forward.  Nothing is dropped when every tile has such an entry (SURFH_OTF_SUPPORT=0: off).
int otf_support(surfh_plan *p, const surfh_config *cfg) {
    const bool on = [] { const char *e = getenv("SURFH_OTF_SUPPORT"); return !(e && e[0] == '0'); }();      // read at plan creation
    if (cfg->exact & 2) return 0;
    if (!on || !cfg->sotf || !p->ilv || p->T < 1 || !p->fuse_mix || p->LP % 128) return 0;
    const int nkb = p->Nb / 2 + 1, nch = (int)(p->LP / 128);
    std::vector<int> bmax(nch, -1), amaxk(nch, -1);   // largest k_beta / folded k_alpha of the support over a chunk's planes (-1: none)
    std::vector<double> km(nkb), kam(p->Na);
    for (int l = 0; l < p->Lown; ++l) {
        if (p->planes[l] < 0) continue;
        std::fill(km.begin(), km.end(), 0.0);
        std::fill(kam.begin(), kam.end(), 0.0);
        double amax = 0.0;
        const double *pl = cfg->sotf + (size_t)p->planes[l] * p->Na * nkb * 2;
        for (int a = 0; a < p->Na; ++a)
            for (int k = 0; k < nkb; ++k) {
                const double re = pl[((size_t)a * nkb + k) * 2], im = pl[((size_t)a * nkb + k) * 2 + 1], m2 = re * re + im * im;
                if (m2 > km[k]) km[k] = m2;
                if (m2 > kam[a]) kam[a] = m2;
            }
        for (int k = 0; k < nkb; ++k) amax = std::max(amax, km[k]);
        const double thr = amax * std::ldexp(1.0, -48);       // squared magnitudes
        int b = -1;
        for (int k = 0; k < nkb; ++k)
            if (km[k] > thr) b = k;
        bmax[l / 128] = std::max(bmax[l / 128], b);
        int af = -1;
        for (int a = 0; a < p->Na; ++a)
            if (kam[a] > thr) af = std::max(af, std::min(a, p->Na - a));
        amaxk[l / 128] = std::max(amaxk[l / 128], af);
    }
    std::vector<int> vlist, kbstart(nkb + 1, 0);
    for (int kb = 0; kb < nkb; ++kb) {
        kbstart[kb] = (int)vlist.size();
        for (int j = 0; j < nch; ++j)
            if (kb <= bmax[j]) vlist.push_back(kb * nch + j);
    }
    kbstart[nkb] = (int)vlist.size();
    if (vlist.empty() || (long)vlist.size() == (long)nkb * nch) return 0;       // nothing to drop (or nothing to keep: leave the passes as they are)
    const size_t nyc = (size_t)2 * p->NAP * p->KBP * p->LP;
    if (dev_upload(&p->otf_vlist, vlist) || dev_upload(&p->otf_kbstart, kbstart) || dev_alloc(&p->ycol_mix, nyc)) return 1;
    if (hipMemset(p->ycol_mix, 0, nyc * sizeof(float)) != hipSuccess) return fail("memset failed");
    p->otf_nvalid = (int)vlist.size();
    std::vector<int> tabs((size_t)(p->ct ? 4 : 3) * nch);
    for (int j = 0; j < nch; ++j) {
        if (p->ct) {
            // sub-sequence n1 of a length R M: element j stands for the rows R j + n1 and N - (R j - n1); inside the support
            // (folded index <= amax) for j <= (amax + R - 1) / R
            const int ja = (std::max(amaxk[j], 0) + p->ctA.R - 1) / p->ctA.R, jb = (std::max(bmax[j], 0) + p->ctB.R - 1) / p->ctB.R;
            tabs[j] = std::min(std::max(ja / 16 + 1, 2), p->ctA.KT);              // forward's complex pass: k-steps in k_alpha
            tabs[nch + j] = std::min(std::max(jb / 16 + 1, 2), p->ctB.KT);        // forward's pass along beta: k-steps in k_beta
            tabs[2 * nch + j] = bmax[j];                                           // adjoint's first pass: last row k_beta stored; reduction: k_beta limit
            tabs[3 * nch + j] = amaxk[j];                                          // adjoint's complex pass: last folded row k_alpha stored; reduction: limit
            continue;
        }
        tabs[j] = std::min(std::max((amaxk[j] + 16) / 16, 2), p->KPa / 16);
        tabs[nch + j] = std::min(std::max((bmax[j] + 16) / 16, 2), p->KPb / 16);
        tabs[2 * nch + j] = bmax[j] + 1;
    }
    const bool ranges = [] { const char *e = getenv("SURFH_OTF_RANGES"); return !(e && e[0] == '0'); }();
    if (ranges && dev_upload(&p->otf_tabs, tabs)) return 1;
    return 0;
}

// `which`: bit 0 = the pass along beta, bit 1 = the pass along alpha (plans whose axes run on different kernels call one of each)
int rfft2_lam_h2(surfh_plan *p, const float *src, float *dst, float *madj = nullptr, bool acols = false, int which = 3) {
    const long LP = p->LP;
    const int ha = p->Na / 2 + 1, hb = p->Nb / 2 + 1;
    const bool sub = acols && p->ycol_adj && p->a_hi > p->a_lo;
    float *const yc = sub ? p->ycol_adj : p->ycol;
    const int a0 = sub ? p->a_lo : 0, na = sub ? p->a_hi - p->a_lo : p->Na;
    DftH2Args g;   // r2c along beta
    g.kind = 1; g.src = src + (long)a0 * LP; g.ldb = p->NAP * LP; g.Kn = p->Nb;
    g.dst = yc + 2 * (long)a0 * LP; g.ldc = 2 * p->NAP * LP; g.e[0] = 1.f; g.e[3] = -1.f; g.rvalid = hb;
    g.KP = p->KPb; g.N = (int)(na * LP);
    // fused tail with the OTF's support: it reads no k_beta beyond the support of a wavelength chunk, so those rows are not stored
    if (madj && p->otf_vlist && p->ycol_mix && p->otf_tabs) { g.rtab = p->otf_tabs + 2 * (LP / 128); g.tabLP = (int)LP; }
    if (which & 1) {
        Prof pr(p, "dft_h2_rows_fwd");
        LAUNCH_OK(launch_dft_h2(p->stream, g, p->h2img + 2 * DFT_H2_IMAGE_HALFS, p->h2kA[2]));
    }
    if (!(which & 2)) return 0;
    DftH2Args h;   // c2c along alpha, batched over k_beta
    h.kind = 0; h.src = yc; h.ldb = 2 * LP; h.sB = 2 * p->NAP * LP; h.Kn = p->Na;
    h.dst = dst; h.ldc = 2 * p->KBP * LP; h.sC = 2 * LP; h.Rn = p->Na; h.rvalid = ha;
    h.KP = p->KPa; h.N = (int)LP; h.batch = hb;
    h.e[0] = 1.f; h.e[1] = 1.f; h.e[2] = 1.f; h.e[3] = -1.f;                 // Re Z[r] = C ae + S bo, Re Z[N-r] = C ae - S bo
    h.e_alt[0] = 1.f; h.e_alt[1] = -1.f; h.e_alt[2] = 1.f; h.e_alt[3] = 1.f;  // Im Z[r] = C be - S ao, Im Z[N-r] = C be + S ao
    if (madj) {
        DftH2AdjMix am;
        am.hsrc = p->sotf; am.ldh = 2 * p->KBP * LP; am.sH = 2 * LP; am.tpl = p->tpl; am.T = p->T; am.LPt = (int)LP; am.mpart = p->adjmix_part;
        if (p->otf_vlist && p->ycol_mix) { am.vlist = p->otf_vlist; am.kbstart = p->otf_kbstart; am.nvalid = p->otf_nvalid; }
        if (sub) {       // rows alpha < a_lo and alpha >= a_hi of the intermediate are zero: leading k-steps (rows k, Na - k) without a non-zero row
            int kt0 = 0;
            while (16 * kt0 + 15 < p->a_lo && p->Na - (16 * kt0 + 15) >= p->a_hi && h.KP / 16 - (kt0 + 1) >= 5) ++kt0;
            am.kt0 = kt0;
        }
        if (p->spec_out && madj == p->spec_out) {      // the solver's scaled half spectrum, mu and the quadratic prior folded in
            am.out_self = p->spec_mu; am.out_pair = p->spec_mu * 1.41421356237309505f; am.Nb = p->Nb;
            am.prior_src = p->spec_prior_src; am.prior_mu = p->spec_prior_mu;
        }
        Prof pr(p, "dft_h2_cols_fwd_adjmix");
        LAUNCH_OK(launch_dft_h2_adjmix(p->stream, h, am, madj, p->PL, p->KBP, p->h2img, p->h2kA[0]));
        return 0;
    }
    {
        Prof pr(p, "dft_h2_cols_fwd");
        LAUNCH_OK(launch_dft_h2(p->stream, h, p->h2img, p->h2kA[0]));
    }
    return 0;
}

// spec [KAP][KBP][LP][2] -> cube [NBP][NAP][LP]        (tmp ycol viewed as Y[NAP][KBP][LP][2])
// `acols`: only the cube columns alpha in [a_lo, a_hi) are wanted (the gathers read nothing else)
int irfft2_lam_h2(surfh_plan *p, const float *src, float *dst, bool mix, bool acols = false, int which = 3) {
    const long LP = p->LP;
    const int ha = p->Na / 2 + 1, hb = p->Nb / 2 + 1;
    DftH2Args g;   // c2c along alpha (optionally with the spectral mix formed in the loader)
    g.kind = 0; g.src = src; g.ldb = 2 * p->KBP * LP; g.Kn = p->Na;
    g.dst = p->ycol; g.ldc = 2 * p->KBP * LP; g.Rn = p->Na; g.rvalid = ha;
    g.KP = p->KPa; g.N = (int)(hb * LP);
    g.e[0] = 1.f; g.e[1] = -1.f; g.e[2] = 1.f; g.e[3] = 1.f;
    g.e_alt[0] = 1.f; g.e_alt[1] = 1.f; g.e_alt[2] = 1.f; g.e_alt[3] = -1.f;
    if (mix) { g.mhat = p->mhat; g.tpl = p->tpl; g.T = p->T; g.LP = (int)p->LP; g.PL = p->PL; g.KBP = p->KBP; }
    if (mix && p->spec_in) { g.mhat = p->spec_in; g.mhat_self = 1.f; g.mhat_pair = 0.70710678118654752f; g.mix_Nb = p->Nb; }
    // the OTF's support: tiles outside it are neither computed nor stored -- their place in ycol_mix is zero for good
    const bool supp = mix && p->otf_vlist && p->ycol_mix && p->adjmix_part;
    float *const yc = supp ? p->ycol_mix : p->ycol;
    if (supp) { g.vlist = p->otf_vlist; g.nvalid = p->otf_nvalid; g.dst = yc; }
    if (supp && p->otf_tabs) { g.ktab = p->otf_tabs; g.tabLP = (int)LP; }          // k_alpha beyond the support: not read
    if (which & 2) {
        Prof pr(p, mix ? "dft_h2_cols_inv_mix" : "dft_h2_cols_inv");
        LAUNCH_OK(launch_dft_h2(p->stream, g, p->h2img, p->h2kA[0]));
    }
    if (!(which & 1)) return 0;
    const bool sub = acols && p->a_hi > p->a_lo;
    const int a0 = sub ? p->a_lo : 0, na = sub ? p->a_hi - p->a_lo : p->Na;
    DftH2Args h;   // c2r along beta, batched over alpha: cube[b] = Gc Yr - Gs Yi, cube[N-b] = Gc Yr + Gs Yi
    h.kind = 2; h.src = yc + (long)a0 * 2 * p->KBP * LP; h.ldb = 2 * LP; h.sB = 2 * p->KBP * LP;
    h.dst = dst + (long)a0 * LP; h.ldc = p->NAP * LP; h.sC = LP;
    h.e[0] = 1.f; h.e[1] = -1.f; h.e[2] = 1.f; h.e[3] = 1.f; h.Rn = p->Nb; h.rvalid = hb;
    h.KP = p->KPb; h.N = (int)LP; h.batch = na;
    if (supp && p->otf_tabs) { h.ktab = p->otf_tabs + LP / 128; h.tabLP = (int)LP; }   // k_beta beyond the support: zero in ycol_mix
    {
        Prof pr(p, "dft_h2_rows_inv");
        LAUNCH_OK(launch_dft_h2(p->stream, h, p->h2img + DFT_H2_IMAGE_HALFS, p->h2kA[1]));
    }
    return 0;
}

// ---- Cooley-Tukey passes (dft_ct.h): the same four passes for N = R * M, interleaved complex arrays ---------------
// cube [NBP][NAP][LP] -> spec [KAP][KBP][LP][2]        (tmp ycol viewed as Z[KBP][NAP][LP][2])
// `lists`: the caller is the adjoint's tail, whose reduction reads the spectrum only inside the OTF's support
int rfft2_lam_ct(surfh_plan *p, const float *src, float *dst, bool acols = false, bool lists = false, int which = 3) {
    const long LP = p->LP;
    const int hb = p->Nb / 2 + 1;
    const bool sub = acols && p->ycol_adj && p->a_hi > p->a_lo;
    float *const yc = sub ? p->ycol_adj : p->ycol;
    const int a0 = sub ? p->a_lo : 0, na = sub ? p->a_hi - p->a_lo : p->Na;
    DftCtArgs g;   // r2c along beta: neighbouring wavelengths as packed pairs a + i b, separated in the epilogue
    g.R = p->ctB.R; g.M = p->ctB.M; g.loader = DFT_CT_PLAIN; g.epi = DFT_CT_HSEP; g.sgn = -1.f;
    g.scale = (float)(0.5 / std::sqrt((double)p->Nb));
    g.src = src + (long)a0 * LP; g.ldb = p->NAP * LP;
    g.dst = yc + 2 * (long)a0 * LP; g.ldc = 2 * p->NAP * LP;
    g.ncols = (int)(na * LP / 2); g.batch = 1;
    // the reduction reads no k_beta beyond the support of a wavelength chunk: those rows are not stored (chunks of 64 packed pairs)
    const bool supp = lists && p->otf_vlist && p->ycol_mix && p->otf_tabs && p->T > 0;
    if (supp) { g.rtab = p->otf_tabs + 2 * (LP / 128); g.tabLP = (int)(LP / 2); g.tabShift = 6; }
    if (which & 1) {
        Prof pr(p, "dft_ct_rows_fwd");
        LAUNCH_OK(launch_dft_ct(p->stream, g, p->ctB));
    }
    if (!(which & 2)) return 0;
    DftCtArgs h;   // c2c along alpha, batched over k_beta
    h.R = p->ctA.R; h.M = p->ctA.M; h.loader = DFT_CT_PLAIN; h.epi = DFT_CT_STORE; h.sgn = -1.f;
    h.scale = (float)(1.0 / std::sqrt((double)p->Na));
    h.src = yc; h.ldb = 2 * LP; h.sB = 2 * p->NAP * LP;
    h.dst = dst; h.ldc = 2 * p->KBP * LP; h.sC = 2 * LP;
    h.ncols = (int)LP; h.batch = hb;
    if (supp) {       // only the (k_beta, wavelength chunk) super-tiles and the rows k_alpha inside the OTF's support
        h.vlist = p->otf_vlist; h.nvalid = p->otf_nvalid;
        h.rtab = p->otf_tabs + 3 * (LP / 128); h.tabLP = (int)LP;
    }
    {
        Prof pr(p, "dft_ct_cols_fwd");
        LAUNCH_OK(launch_dft_ct(p->stream, h, p->ctA));
    }
    return 0;
}

// element-wise product formed in the loader of the first inverse pass (dft_ct.h, loader PROD): src * prod (sign +1) or
// src * conj(prod) (-1), times `scale` -- the plane-wise path's OTF product without its own kernel and array
struct ProdOperand {
    const float *prod = nullptr;
    float sign = 1.f, scale = 1.f;
    // + add_w |D|^2 add (loader PRODADD): the quadratic prior's term of the plane-wise normal operator, `add` = the spectrum of
    // the vector the operator is applied to, |D|^2 = the circular first differences' transfer function (fusion_CT.py:16-43)
    const float *add = nullptr;
    float add_w = 0.f;
};
// the complex pass along alpha runs on the kernel that has the PROD loader (SURFH_OTF_PROD=0: the separate product kernels)
bool prod_capable(const surfh_plan *p) { return p->otf_prod && p->T == 0 && p->ilv && !p->dense_dft && p->ax_a == 2; }

// spec [KAP][KBP][LP][2] -> cube [NBP][NAP][LP]        (tmp ycol viewed as Y[NAP][KBP][LP][2])
int irfft2_lam_ct(surfh_plan *p, const float *src, float *dst, bool mix, bool acols = false, int which = 3, const ProdOperand *po = nullptr) {
    const long LP = p->LP;
    const int hb = p->Nb / 2 + 1;
    DftCtArgs g;   // c2c along alpha (optionally with the spectral mix formed in the loader)
    g.R = p->ctA.R; g.M = p->ctA.M; g.loader = mix ? DFT_CT_MIX : DFT_CT_PLAIN; g.epi = DFT_CT_STORE; g.sgn = 1.f;
    g.scale = (float)(1.0 / std::sqrt((double)p->Na));
    g.src = src; g.ldb = 2 * p->KBP * LP;
    g.dst = p->ycol; g.ldc = 2 * p->KBP * LP;
    g.ncols = (int)(hb * LP); g.batch = 1;
    if (mix) { g.mhat = p->mhat; g.tpl = p->tpl; g.T = p->T; g.LP = (int)p->LP; g.PL = p->PL; g.KBP = p->KBP; }
    if (mix && p->spec_in) { g.mhat = p->spec_in; g.mhat_self = 1.f; g.mhat_pair = 0.70710678118654752f; g.mix_Nb = p->Nb; }
    if (po && po->prod && !mix) {
        g.loader = DFT_CT_PROD; g.prod = po->prod; g.ldp = g.ldb; g.sP = 0; g.prod_sign = po->sign; g.scale *= po->scale;
        if (po->add && po->add_w != 0.f) {
            g.loader = DFT_CT_PRODADD; g.add = po->add; g.add_w = po->add_w; g.add_Nb = p->Nb; g.LP = (int)p->LP;
        }
    }
    // the OTF's support: tiles outside it are neither computed nor stored -- their place in ycol_mix is zero for good
    const bool supp = mix && p->otf_vlist && p->ycol_mix && p->otf_tabs;
    float *const yc = supp ? p->ycol_mix : p->ycol;
    if (supp) { g.vlist = p->otf_vlist; g.nvalid = p->otf_nvalid; g.dst = yc; g.ktab = p->otf_tabs; g.tabLP = (int)LP; }
    if (which & 2) {
        Prof pr(p, mix ? "dft_ct_cols_inv_mix" : (g.loader == DFT_CT_PROD ? "dft_ct_cols_inv_prod" : g.loader == DFT_CT_PRODADD ? "dft_ct_cols_inv_prodadd" : "dft_ct_cols_inv"));
        LAUNCH_OK(launch_dft_ct(p->stream, g, p->ctA));
    }
    if (!(which & 1)) return 0;
    const bool sub = acols && p->a_hi > p->a_lo;
    const int a0 = sub ? p->a_lo : 0, na = sub ? p->a_hi - p->a_lo : p->Na;
    DftCtArgs h;   // c2r along beta, batched over alpha: two neighbouring half spectra as one Hermitian-extended complex sequence
    h.R = p->ctB.R; h.M = p->ctB.M; h.loader = DFT_CT_HPACK; h.epi = DFT_CT_STORE; h.sgn = 1.f;
    h.scale = (float)(1.0 / std::sqrt((double)p->Nb));
    h.src = yc + (long)a0 * 2 * p->KBP * LP; h.ldb = 2 * LP; h.sB = 2 * p->KBP * LP;
    h.dst = dst + (long)a0 * LP; h.ldc = p->NAP * LP; h.sC = LP;
    h.ncols = (int)(LP / 2); h.batch = na;
    if (supp) { h.ktab = p->otf_tabs + LP / 128; h.tabLP = (int)(LP / 2); h.tabShift = 6; }      // k_beta beyond the support: zero in ycol_mix
    {
        Prof pr(p, "dft_ct_rows_inv");
        LAUNCH_OK(launch_dft_ct(p->stream, h, p->ctB));
    }
    return 0;
}

// the two transforms on interleaved arrays, each pass on the kernel of its axis (surfh_plan::ax_a / ax_b)
int rfft2_lam_ilv(surfh_plan *p, const float *src, float *dst, float *madj = nullptr, bool acols = false, bool lists = false) {
    if (p->h2) return rfft2_lam_h2(p, src, dst, madj, acols);
    if (p->ax_b == 1 ? rfft2_lam_h2(p, src, dst, nullptr, acols, 1) : rfft2_lam_ct(p, src, dst, acols, lists, 1)) return 1;
    return p->ax_a == 1 ? rfft2_lam_h2(p, src, dst, nullptr, acols, 2) : rfft2_lam_ct(p, src, dst, acols, lists, 2);
}
int irfft2_lam_ilv(surfh_plan *p, const float *src, float *dst, bool mix = false, bool acols = false, const ProdOperand *po = nullptr) {
    if (po && p->ax_a != 2) return fail("irfft2: the product loader needs the Cooley-Tukey pass along alpha");
    if (p->h2) return irfft2_lam_h2(p, src, dst, mix, acols);
    if (p->ax_a == 1 ? irfft2_lam_h2(p, src, dst, mix, acols, 2) : irfft2_lam_ct(p, src, dst, mix, acols, 2, po)) return 1;
    return p->ax_b == 1 ? irfft2_lam_h2(p, src, dst, mix, acols, 1) : irfft2_lam_ct(p, src, dst, mix, acols, 1);
}

// ---------------------------------------------------------------------------------------------
// pipelines on device buffers
// ---------------------------------------------------------------------------------------------
int rfft2_cube(surfh_plan *p, const float *src, float *dst) { return p->dense_dft ? rfft2_lam(p, src, dst) : rfft2_lam_ilv(p, src, dst); }
int irfft2_cube(surfh_plan *p, const float *src, float *dst, bool mix = false, bool acols = false, const ProdOperand *po = nullptr) {
    if (po && p->dense_dft) return fail("irfft2: the product loader is not part of the dense plan");
    return p->dense_dft ? irfft2_lam(p, src, dst) : irfft2_lam_ilv(p, src, dst, mix, acols, po);
}

// mhat[t] = sum_l tpl[t][l] conj(sotf[l]) rfft2(cube[l])  (T > 0), or the per-plane product (T == 0)
// `acols`: the cube is zero outside the alpha range of the channels' tables (the adjoint's accumulator)
int adjoint_tail(surfh_plan *p, const float *cube, bool acols = false) {
    if (p->adjmix_part && p->h2 && p->T > 0) return rfft2_lam_h2(p, cube, p->spec, p->spec_out ? p->spec_out : p->mhat, acols);
    if (p->ct && !p->dense_dft) {
        if (rfft2_lam_ilv(p, cube, p->spec, nullptr, acols, true)) return 1;
        if (prod_capable(p)) return 0;      // plane-wise: conj(OTF) x spec is formed by the loader of the inverse transform that follows
        SpecmixAdjOpt o;
        o.Na = p->Na; o.KBP = p->KBP;
        if (p->T > 0 && p->otf_vlist && p->ycol_mix && p->otf_tabs) o.lim = p->otf_tabs + 2 * (p->LP / 128);
        if (p->spec_out) {      // the solver's scaled half spectrum, mu and the quadratic prior folded in
            o.Nb = p->Nb; o.out_self = p->spec_mu; o.out_pair = p->spec_mu * 1.41421356237309505f;
            o.prior_src = p->spec_prior_src; o.prior_mu = p->spec_prior_mu;
        }
        Prof pr(p, "specmix_adj");
        if (p->T == 0 && p->pn_fold_prior) {      // plane-wise normal operator: mu and the quadratic prior in the OTF product (see below)
            SpecmixAdjOpt o2;
            o2.Na = p->Na; o2.Nb = p->Nb; o2.KBP = p->KBP; o2.out_self = (float)p->pl_mu; o2.prior_src = p->mhat; o2.prior_mu = (float)p->pl_mu_reg;
            LAUNCH_OK(launch_specmix_adj(p->stream, p->spec, p->sotf, p->tpl, p->mhat, 0, p->PL, p->LP, false, 1, &o2));
            return 0;
        }
        LAUNCH_OK(launch_specmix_adj(p->stream, p->spec, p->sotf, p->tpl, p->spec_out ? p->spec_out : p->mhat, p->T, p->PL, p->LP, false, 1,
                                     p->T > 0 ? &o : nullptr));
        return 0;
    }
    if (rfft2_cube(p, cube, p->spec)) return 1;
    Prof pr(p, "specmix_adj");
    if (p->T == 0 && p->ilv && p->pn_fold_prior) {
        // plane-wise normal operator: `mhat` still holds the spectrum of the vector the forward half was applied to -- mu and the
        // quadratic prior go into the OTF product, no prior kernel and no scaling pass afterwards
        SpecmixAdjOpt o;
        o.Na = p->Na; o.Nb = p->Nb; o.KBP = p->KBP; o.out_self = (float)p->pl_mu; o.prior_src = p->mhat; o.prior_mu = (float)p->pl_mu_reg;
        LAUNCH_OK(launch_specmix_adj(p->stream, p->spec, p->sotf, p->tpl, p->mhat, 0, p->PL, p->LP, false, 1, &o));
        return 0;
    }
    LAUNCH_OK(launch_specmix_adj(p->stream, p->spec, p->sotf, p->tpl, p->mhat, p->T, p->PL, p->LP, p->verify, p->ilv));
    return 0;
}

// `hand_over`: the caller is the normal operator -- channels with a spectral-blur GEMM do not write y but leave the adjoint's
// GEMM operand (fp16 pieces of ymat + row maxima) behind
int forward_dev(surfh_plan *p, const float *x, float *y, bool hand_over = false) {
    hipStream_t s = p->stream;
    if (p->spec_in) {
        // the maps' spectra are the caller's vector: nothing to transform
    } else if (p->T > 0) {
        {
            Prof pr(p, "pad_planes");
            LAUNCH_OK(launch_pad_planes(s, x, p->maps_pad, p->T, p->Na, p->Nb, p->NAP, p->NBP));
        }
        if (rfft2_planes(p, p->maps_pad, p->mhat, p->T)) return 1;
    } else if (p->pn_native) {
        // plane-wise solver on wavelength-innermost vectors: x is already in the cube's layout [NBP][NAP][LP]
        if (rfft2_cube(p, x, p->mhat)) return 1;
    } else {
        {
            Prof pr(p, "cube_transpose");
            for (auto &g : p->segs)
                LAUNCH_OK(launch_cube_to_lam_inner(s, x, p->cube + g.coff, g.start, g.len, p->Na, p->Nb, p->NAP, p->LP));
        }
        if (rfft2_cube(p, p->cube, p->mhat)) return 1;
    }
    if (p->T > 0 && p->T <= 4 && p->fuse_mix && !p->dense_dft) {
        // spectral mix x OTF fused into the loader of the first inverse pass: `spec` is never written
        // (the normal operator needs the blurred cube only where a gather reads it)
        if (irfft2_cube(p, p->sotf, p->cube, true, hand_over)) return 1;
    } else if (prod_capable(p)) {
        // plane-wise model on the Cooley-Tukey passes: OTF x spectrum in the loader of the first inverse pass, `spec` is never written
        ProdOperand po;
        po.prod = p->sotf;
        if (irfft2_cube(p, p->mhat, p->cube, false, false, &po)) return 1;
    } else {
        {
            Prof pr(p, "specmix_fwd");
            LAUNCH_OK(launch_specmix_fwd(s, p->mhat, p->sotf, p->tpl, p->spec, p->T, p->PL, p->LP, p->ilv));
        }
        if (irfft2_cube(p, p->spec, p->cube)) return 1;
    }
    // gather on the main stream, spectral-blur GEMM + slab sum on the second one: GEMM(c) overlaps gather(c+1)
    hipStream_t sB = (p->overlap && p->stream2) ? p->stream2 : s;
    for (auto &c : p->ch) {
        const bool f16 = c.W16 != nullptr;
        {
            Prof pr(p, "spmm_gather_fwd");
            if (c.Xs16 && c.fwd.g.NG)     // straight to the block-scaled fp16 pieces of the all-consumer GEMM
                LAUNCH_OK(launch_spmm_group_gather_f16(s, c.fwd.g, p->cube, c.Xs16, (long)c.NP * c.K, c.nlam, c.bscale, c.NP, c.K, c.LinP));
            else if (c.Xs16)
                LAUNCH_OK(launch_spmm_rows_f16(s, c.fwd.t, p->cube, c.Xs16, (long)c.NP * c.K, c.nlam, c.bscale, c.NP, c.K, c.LinP));
            else if (p->verify)
                LAUNCH_OK(launch_spmm_rows_f64acc(s, c.fwd.t, p->cube, c.Xs, c.nlam, 0));
            else
                LAUNCH_OK(launch_spmm_rows(s, c.fwd.t, p->cube, c.Xs, c.nlam, 0));
        }
        if (c.bsum) {   // y[l][(p,s,a)] = Xs[(p,s,a)][l]
            Prof pr(p, "y_transpose");
            LAUNCH_OK(launch_cube_from_lam_inner(s, c.Xs + c.shift, y + c.yoff, 0, c.Lin, 1, c.P * c.S * c.aout, 1, c.LinP));
            continue;
        }
        if (chain(p, s, sB)) return 1;
        GemmArgs g;   // y^T[n][l'] = sum_k Xs[n][k] W[l'][k]
        g.A0 = c.Xs; g.lda = c.K;
        g.C = c.Cpart; g.ldc = c.LdetP;
        g.M = c.NP; g.N = c.LdetP; g.K = c.K; g.splitK = c.splitK; g.sCsplit = (long)c.NP * c.LdetP;
        {
            Prof pr(p, "gemm_wblur_fwd", sB);
            if (p->wblur_fp32) {
                g.B0 = c.Wt; g.ldb = c.LdetP;        // B as [K][N]
                LAUNCH_OK(gemm32(p, sB, g));
            } else {
                // both operands as fp16 pieces (the gather wrote the block-scaled pieces of Xs): 256 x 256 all-consumer kernel
                g.ldb = c.K;                         // B as [N][K]
                g.B16 = c.W16; g.pB16 = (long)c.LdetP * c.K; g.sB16 = c.sW;
                g.A3 = c.Xs16; g.pA3 = (long)c.NP * c.K;
                g.bscale = c.bscale; g.segLinP = c.LinP; g.segChunks = (c.LinP + 1023) / 1024;
                g.klist = c.klF; g.klistStride = c.klFs;
                LAUNCH_OK(launch_gemm_nt_f16x2_cc(sB, g));
            }
        }
        if (hand_over && c.ymat16) {
            Prof pr(p, "ymat16_from_cpart", sB);
            LAUNCH_OK(launch_ymat16_from_cpart(sB, c.Cpart, (long)c.NP * c.LdetP, c.splitK, c.ymat16, (long)c.NP * c.LdetP, c.amax, c.NP,
                                               c.P * c.S * c.aout, c.Ldet, c.LdetP));
        } else {
            Prof pr(p, "y_from_cpart", sB);
            LAUNCH_OK(launch_y_from_cpart(sB, c.Cpart, (long)c.NP * c.LdetP, c.splitK, y + c.yoff, c.P * c.S, c.Ldet,
                                          c.aout, c.LdetP));
        }
    }
    if (chain(p, sB, s)) return 1;     // everything after this call sees y complete
    return 0;
}

// `handed_over`: forward_dev(hand_over) has just left the GEMM operands of the channels with a spectral blur behind
int adjoint_dev(surfh_plan *p, const float *y, float *x, bool ref, bool handed_over = false) {
    hipStream_t s = p->stream;
    // detector-side work (y -> ymat, R^T GEMM) on the second stream, cube-side scatter on the main one: GEMM(c+1)
    // overlaps scatter(c); the scatters stay in channel order on one stream because their windows overlap
    hipStream_t sB = (p->overlap && p->stream2) ? p->stream2 : s;
    if (chain(p, s, sB)) return 1;     // y (and the previous users of Xs / ymat) are ordered before the second stream's work
    // the exact adjoint accumulates in its own buffer without clearing it (surfh_plan::gcube); the reference adjoint and the
    // verification plan read-modify-write every row of the cleared work cube
    float *const acc = (!ref && p->gcube) ? p->gcube : p->cube;
    if (acc == p->cube) {
        Prof pr(p, "fill_zero");
        LAUNCH_OK(launch_fill_zero(s, p->cube, (long)p->NBP * p->NAP * p->LP));
    }
    // detector side of one channel: y -> ymat (-> its fp16 pieces, unless the forward half has just left them behind)
    auto prepare = [&](Channel &c) -> int {
        const bool f16 = c.W16 != nullptr;
        if (handed_over && f16 && c.ymat16) return 0;
        {
            Prof pr(p, "ymat_from_y", sB);
            LAUNCH_OK(launch_ymat_from_y(sB, y + c.yoff, c.ymat, c.P * c.S, c.Ldet, c.aout, c.LdetP, f16 ? c.pmax : nullptr,
                                         f16 ? c.amax : nullptr, c.NP));
        }
        if (f16 && !p->wblur_fp32)
            LAUNCH_OK(launch_split_rows2h(sB, c.ymat, c.amax, c.ymat16, c.NP, c.LdetP, (long)c.NP * c.LdetP));   // one scale per row
        return 0;
    };
    auto gemm_args = [&](Channel &c) {   // Xs_t[n][k] = sum_l' y^T[n][l'] W[l'][k]
        GemmArgs g;
        g.A0 = c.ymat; g.lda = c.LdetP;
        g.C = c.Xs; g.ldc = c.K;
        g.M = c.NP; g.N = c.K; g.K = c.LdetP;
        if (p->wblur_fp32) {
            g.B0 = c.W; g.ldb = c.K;             // B as [K'=l'][N'=k]
        } else if (c.W16) {
            g.K = (c.Ldet + 31) / 32 * 32;       // the columns of ymat beyond Ldet are zero: whole K steps of them are skipped
            g.ldb = c.LdetP;                     // B as [N'=k][K'=l']
            g.B16 = c.Wt16; g.pB16 = (long)c.LdetP * c.K; g.sB16 = c.sW; g.amax = c.amax;
            g.A3 = c.ymat16; g.pA3 = (long)c.NP * c.LdetP;
            g.klist = c.klA; g.klistStride = c.klAs;
            if (c.klA && c.permA) { g.permP = c.permA; g.permLin = c.LinP; }
        }
        return g;
    };
    // The two-piece fp16 GEMMs of up to four channels go out as ONE launch (each is 1.5-1.8 rounds of workgroups on its own;
    // their operands and outputs are per channel, so nothing orders them among themselves): detector-side preparation of all
    // of them first, the grouped GEMM, then the scatters in channel order.  SURFH_GEMM_GROUPED=0: one launch per channel.
    std::vector<char> gemm_done(p->ch.size(), 0);
    if (p->gemm_grouped && !p->wblur_fp32 && !p->verify) {
        std::vector<GemmArgs> ga;
        std::vector<size_t> gc;
        for (size_t ci = 0; ci <= p->ch.size(); ++ci) {
            const bool last = ci == p->ch.size();
            if (!last) {
                Channel &c = p->ch[ci];
                if (c.bsum || !c.W16 || (ref && !c.has_ref)) continue;
                if (prepare(c)) return 1;
                ga.push_back(gemm_args(c)); gc.push_back(ci);
            }
            if (!ga.empty() && (last || (int)ga.size() == GEMM_GROUP_MAX)) {
                {
                    Prof pr(p, "gemm_wblur_adj", sB);
                    LAUNCH_OK(launch_gemm_nt_f16x2_cc_group(sB, ga.data(), (int)ga.size()));
                }
                for (size_t i : gc) gemm_done[i] = 1;
                ga.clear(); gc.clear();
            }
        }
    }
    for (size_t ci = 0; ci < p->ch.size(); ++ci) {
        Channel &c = p->ch[ci];
        if (ref && !c.has_ref) return fail("adjoint_ref needs the gridding_t tables (gt_*) in the channel descriptor");
        if (c.bsum) {
            {
                Prof pr(p, "y_transpose");
                LAUNCH_OK(launch_cube_to_lam_inner(s, y + c.yoff, c.Xs + c.shift, 0, c.Lin, 1, c.P * c.S * c.aout, 1, c.LinP));
            }
            Prof pr(p, ref ? "spmm_degrid_ref" : "spmm_scatter_adj");
            if (p->verify)
                LAUNCH_OK(launch_spmm_rows_f64acc(s, ref ? c.adjRef.t : c.adjT.t, c.Xs, acc, c.nlam, 1));
            else if (!ref && c.adjT.g.NG)
                LAUNCH_OK(launch_spmm_group_scatter(s, c.adjT.g, c.Xs, acc, c.nlam));
            else
                LAUNCH_OK(launch_spmm_rows(s, ref ? c.adjRef.t : c.adjT.t, c.Xs, acc, c.nlam, 1));
            continue;
        }
        if (!gemm_done[ci]) {
            if (prepare(c)) return 1;
            const GemmArgs g = gemm_args(c);
            Prof pr(p, "gemm_wblur_adj", sB);
            if (p->wblur_fp32 || !c.W16) LAUNCH_OK(gemm32(p, sB, g));
            else LAUNCH_OK(launch_gemm_nt_f16x2_cc(sB, g));
        }
        if (chain(p, sB, s)) return 1;
        {
            Prof pr(p, ref ? "spmm_degrid_ref" : "spmm_scatter_adj");
            if (p->verify)
                LAUNCH_OK(launch_spmm_rows_f64acc(s, ref ? c.adjRef.t : c.adjT.t, c.Xs, acc, c.nlam, 1));
            else if (!ref && c.adjT.g.NG)
                LAUNCH_OK(launch_spmm_group_scatter(s, c.adjT.g, c.Xs, acc, c.nlam));
            else
                LAUNCH_OK(launch_spmm_rows(s, ref ? c.adjRef.t : c.adjT.t, c.Xs, acc, c.nlam, 1));
        }
    }
    if (adjoint_tail(p, acc, true)) return 1;
    if (p->spec_out) return 0;         // the caller's vector is the spectrum
    if (p->T > 0) {
        if (irfft2_planes(p, p->mhat, p->maps_pad, p->T)) return 1;
        Prof pr(p, "unpad_planes");
        LAUNCH_OK(launch_unpad_planes(s, p->maps_pad, x, p->T, p->Na, p->Nb, p->NAP, p->NBP));
    } else if (prod_capable(p)) {
        // conj(OTF) x spectrum of the accumulated cube in the loader; inside the plane-wise normal operator mu rides on the pass and
        // the quadratic prior comes in as a third operand: `mhat` still holds the spectrum of the vector the forward half was applied to
        ProdOperand po;
        po.prod = p->sotf; po.sign = -1.f; po.scale = p->pn_fold_prior ? (float)p->pl_mu : 1.f;
        if (p->pn_fold_prior && p->pl_mu_reg != 0.0 && p->pl_mu != 0.0) {      // q = mu (A^T A d + (mu_r / mu) D^T D d)
            po.add = p->mhat; po.add_w = (float)(p->pl_mu_reg / p->pl_mu);
        }
        float *const out = p->pn_native ? x : p->cube;
        if (irfft2_cube(p, p->spec, out, false, false, &po)) return 1;
        if (!p->pn_native) {
            const long pl = (long)p->Na * p->Nb;
            if (p->Lown < p->Lc) LAUNCH_OK(launch_fill_zero(s, x, (long)p->Lc * pl));   // planes no channel observes
            Prof pr(p, "cube_transpose");
            for (auto &g : p->segs)
                LAUNCH_OK(launch_cube_from_lam_inner(s, p->cube + g.coff, x, g.start, g.len, p->Na, p->Nb, p->NAP, p->LP));
        }
    } else if (p->pn_native) {
        if (irfft2_cube(p, p->mhat, x)) return 1;           // straight into the caller's wavelength-innermost vector
    } else {
        if (irfft2_cube(p, p->mhat, p->cube)) return 1;
        const long pl = (long)p->Na * p->Nb;
        if (p->Lown < p->Lc) LAUNCH_OK(launch_fill_zero(s, x, (long)p->Lc * pl));   // planes no channel observes
        Prof pr(p, "cube_transpose");
        for (auto &g : p->segs)
            LAUNCH_OK(launch_cube_from_lam_inner(s, p->cube + g.coff, x, g.start, g.len, p->Na, p->Nb, p->NAP, p->LP));
    }
    return 0;
}

// the normal operator's two halves exchange the GEMM operands directly (SURFH_NORMAL_FUSED=0: through y)
bool normal_hand_over(const surfh_plan *p) {
    static const bool fused = [] { const char *e = getenv("SURFH_NORMAL_FUSED"); return !(e && e[0] == '0'); }();
    return fused && !p->verify && !p->wblur_fp32;
}

int normal_dev(surfh_plan *p, const float *d, float *q, double mu) {
    // y is only the hand-over between the two halves: the channels' slab sums go straight into the adjoint's GEMM operands
    // (SURFH_NORMAL_FUSED=0: through y, as forward() + adjoint() do)
    const bool ho = normal_hand_over(p);
    if (forward_dev(p, d, p->cg_y, ho)) return 1;
    if (adjoint_dev(p, p->cg_y, q, false, ho)) return 1;
    if (mu != 1.0) {
        Prof pr(p, "scale");
        LAUNCH_OK(launch_scale(p->stream, q, p->pn_native ? (long)p->NBP * p->NAP * p->LP : p->isize, (float)mu));
    }
    return 0;
}

// explicit per-frequency Hessian of Model_WCT and its work buffer: published only once both allocations and the launch
// that fills `hth` have succeeded (a half-built pair would make the next call skip the launch)
int ensure_hessian(surfh_plan *p) {
    if (p->hth && p->mhat2) return 0;
    float *h = nullptr, *m2 = nullptr;
    if (dev_alloc(&h, (size_t)p->T * p->T * p->PL) || dev_alloc(&m2, (size_t)p->T * 2 * p->PL)) {
        hipFree(h);
        hipFree(m2);
        return 1;
    }
    const int rc = launch_wct_hessian(p->stream, p->sotf, p->tpl, h, p->T, p->PL, p->LP, p->ilv);
    if (rc != 0) {
        hipFree(h);
        hipFree(m2);
        return fail("launch_wct_hessian failed: %s", hipGetErrorString((hipError_t)rc));
    }
    p->hth = h;
    p->mhat2 = m2;
    return 0;
}

int ensure_cg(surfh_plan *p) {
    if (p->cg_x) return 0;
    // room for the vectors in either basis: the maps [T][Na][Nb] or their scaled half spectra [T][2][KAP][KBP]
    const size_t n = std::max((size_t)p->isize, (size_t)2 * std::max(p->T, 0) * (size_t)p->PL);
    for (float **v : {&p->cg_x, &p->cg_r, &p->cg_d, &p->cg_q, &p->cg_b})
        if (dev_alloc(v, n)) return 1;
    return 0;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char *surfh_last_error(void) { return g_err.c_str(); }
int surfh_version(void) { return 100; }

int surfh_plan_destroy(surfh_plan *p) {
    if (!p) return 0;
    hipSetDevice(p->dev);
    if (p->stream) hipStreamSynchronize(p->stream);
    for (float *v : {p->sotf, p->tpl, p->mhat, p->spec, p->ycol, p->cube, p->ycol_maps, p->maps_pad, p->Fi, p->Gi, p->Gf,
                     p->Ff, p->GiT, p->GfT, p->Cma, p->Sma, p->Gc, p->Gs, p->Cf, p->Sf, p->io_cube, p->hth, p->mhat2, p->gcube, p->io_x, p->io_y, p->cg_x, p->cg_r, p->cg_d, p->cg_q, p->cg_b, p->cg_y, p->cg_qm, p->cg_dd})
        hipFree(v);
    hipFree(p->h2img);
    if (p->ctB.img != p->ctA.img) dft_ct_plan_destroy(&p->ctB);
    else p->ctB = DftCtPlan();
    dft_ct_plan_destroy(&p->ctA);
    hipFree(p->adjmix_part);
    hipFree(p->otf_vlist);
    hipFree(p->otf_kbstart);
    hipFree(p->otf_tabs);
    hipFree(p->ycol_mix);
    hipFree(p->ycol_adj);
    hipFree(p->dscal);
    hipFree(p->dscratch);
    hipFree(p->cg_hist);
    hipFree(p->pl_sc);
    for (float *v : p->pn_v) hipFree(v);
    hipFree(p->pn_sc);
    hipFree(p->pn_part);
    for (auto &c : p->ch) {
        for (float *v : {c.W, c.Wt, c.Xs, c.Cpart, c.ymat}) hipFree(v);
        hipFree(c.W16);
        hipFree(c.Wt16);
        hipFree(c.klF);
        hipFree(c.klA);
        hipFree(c.Xs16);
        hipFree(c.bscale);
        hipFree(c.ymat16);
        hipFree(c.amax);
        hipFree(c.pmax);
        free_ell(&c.fwd);
        free_ell(&c.adjT);
        free_ell(&c.adjRef);
    }
    for (auto &r : p->pending) {
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    for (auto e : p->pool) hipEventDestroy(e);
    for (hipEvent_t e : p->sync_ev) hipEventDestroy(e);
    if (p->stream2) hipStreamDestroy(p->stream2);
    if (p->own_stream && p->stream) hipStreamDestroy(p->stream);
    delete p;
    return 0;
}

int surfh_plan_create(const surfh_config *cfg, surfh_plan **out) {
    if (!cfg || !out) return fail("null argument");
    *out = nullptr;
    if (cfg->n_alpha < 2 || cfg->n_beta < 2 || cfg->n_lambda < 1) return fail("bad cube shape");
    if (cfg->n_channels < 0 || (cfg->n_channels > 0 && !cfg->channels)) return fail("bad channel list");
    if (cfg->n_channels == 0 && cfg->n_templates < 1) return fail("a plan without channels needs templates (Model_WCT)");
    if (!cfg->sotf && cfg->n_templates > 0) return fail("sotf is NULL");      // NULL = no spatial blur, plane-wise plans only
    if (cfg->n_templates > 0 && !cfg->templates) return fail("templates is NULL");
    if (cfg->n_templates > SURFH_MAX_TEMPLATES)
        return fail("n_templates = %d: at most %d templates are supported", cfg->n_templates, SURFH_MAX_TEMPLATES);
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail("device %d not available (%d devices): the HIP path has no CPU fallback", cfg->device, ndev);
    HIP_OK(hipSetDevice(cfg->device));

    surfh_plan *p = new surfh_plan();
    p->dev = cfg->device;
    if (hipDeviceGetAttribute(&p->n_cu, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || p->n_cu < 1) p->n_cu = 256;
    auto bail = [&](int) {
        surfh_plan_destroy(p);
        return 1;
    };
    if (cfg->stream) {
        p->stream = (hipStream_t)cfg->stream;
    } else {
        if (hipStreamCreate(&p->stream) != hipSuccess) return bail(fail("hipStreamCreate failed"));
        p->own_stream = true;
    }
    p->Na = cfg->n_alpha;
    p->Nb = cfg->n_beta;
    p->Lc = cfg->n_lambda;
    p->T = cfg->n_templates;
    p->NAP = pad64(p->Na);
    p->NBP = pad64(p->Nb);
    p->KAP = p->NAP;
    p->KBP = pad64(p->Nb / 2 + 1);
    p->PL = (long)p->KAP * p->KBP;
    p->PLc = (long)p->NAP * p->NBP;
    p->lo = p->Lc;
    p->hi = 0;
    for (int i = 0; i < cfg->n_channels; ++i) {
        p->lo = std::min(p->lo, (int)cfg->channels[i].wslice_start);
        p->hi = std::max(p->hi, (int)cfg->channels[i].wslice_stop);
    }
    if (cfg->n_channels == 0) { p->lo = 0; p->hi = p->Lc; }
    if (p->lo < 0 || p->hi > p->Lc || p->lo >= p->hi) return bail(fail("channel wslices outside the cube"));
    {   // merge the channel windows into disjoint segments; the plan stores only those planes
        std::vector<std::pair<int, int>> iv;
        if (cfg->n_channels == 0) iv.push_back({0, p->Lc});
        for (int i = 0; i < cfg->n_channels; ++i) iv.push_back({cfg->channels[i].wslice_start, cfg->channels[i].wslice_stop});
        std::sort(iv.begin(), iv.end());
        int coff = 0;
        for (auto &v : iv) {
            if (v.first >= v.second) return bail(fail("empty channel window"));
            if (!p->segs.empty() && v.first <= p->segs.back().start + p->segs.back().len) {
                auto &g = p->segs.back();
                const int grow = std::max(0, v.second - (g.start + g.len));
                g.len += grow;
                coff += grow;
            } else {
                // every segment starts on a multiple of 4 compact planes (16-byte vector access per channel window)
                coff = (coff + 3) / 4 * 4;
                p->segs.push_back({v.first, v.second - v.first, coff});
                coff += v.second - v.first;
            }
        }
        p->Lown = coff;
        p->planes.assign(p->Lown, -1);
        for (auto &g : p->segs)
            for (int l = 0; l < g.len; ++l) p->planes[g.coff + l] = g.start + l;
    }
    p->LP = (p->Lown + 127) / 128 * 128;
    p->isize = (long)(p->T > 0 ? p->T : p->Lc) * p->Na * p->Nb;
    const size_t LP = (size_t)p->LP;

    {   // which transform kernels run decides the layout of the complex arrays: two-piece fp16 passes (default where
        // the matrices fit LDS) keep them interleaved.  The kernel is chosen per axis: dft_h2 (16 < n/2+1 <= 128, row offsets below
        // 4 GB), else dft_ct (n = R M), else none -- then the whole plan runs the dense fp32 products on planar arrays.
        // SURFH_DFT_H2=0 / SURFH_DFT_CT=0 take a kernel out of the choice (A/B), SURFH_DFT_DENSE=1 forces the dense products.
        const char *eh = getenv("SURFH_DFT_H2"), *ec = getenv("SURFH_DFT_CT"), *ed = getenv("SURFH_DFT_DENSE");
        const bool h2_on = !(eh && eh[0] == '0'), ct_on = !(ec && ec[0] == '0');
        auto axis = [&](int n) {
            if (h2_on && dft_h2_supported(n, n, p->NAP, p->KBP, p->LP)) return 1;
            if (ct_on && dft_ct_factor(n, nullptr, nullptr)) return 2;
            return 0;
        };
        p->ax_a = axis(p->Na);
        p->ax_b = axis(p->Nb);
        p->ilv = !cfg->verify && !(ed && ed[0] == '1') && p->LP % 128 == 0 && p->ax_a && p->ax_b;
        if (!p->ilv) p->ax_a = p->ax_b = 0;
        p->h2 = p->ilv && p->ax_a == 1 && p->ax_b == 1;
        p->ct = p->ilv && !p->h2;
    }
    // ---- constants ------------------------------------------------------------------------
    {   // sotf [Lc][Na][Nb/2+1] complex128  ->  [2][KAP][KBP][LP] float (h2: [KAP][KBP][LP][2]), wavelength innermost
        const int nkb = p->Nb / 2 + 1;
        const size_t nsp = (size_t)2 * p->PL * LP;
        if (dev_alloc(&p->sotf, nsp)) return bail(1);
        if (hipMemset(p->sotf, 0, nsp * sizeof(float)) != hipSuccess) return bail(fail("memset failed"));
        std::vector<float> row((size_t)2 * p->KBP * LP);
        for (int a = 0; a < p->Na; ++a) {
            std::fill(row.begin(), row.end(), 0.f);
            for (int l = 0; l < p->Lown; ++l) {
                if (p->planes[l] < 0) continue;     // alignment gap between segments: stays zero
                const double *src = cfg->sotf ? cfg->sotf + ((size_t)p->planes[l] * p->Na + a) * nkb * 2 : nullptr;
                for (int k = 0; k < nkb; ++k) {
                    const float vr = src ? (float)src[2 * k] : 1.f, vi = src ? (float)src[2 * k + 1] : 0.f;
                    if (p->ilv) {
                        row[((size_t)k * LP + l) * 2] = vr;
                        row[((size_t)k * LP + l) * 2 + 1] = vi;
                    } else {
                        row[(size_t)k * LP + l] = vr;
                        row[((size_t)p->KBP + k) * LP + l] = vi;
                    }
                }
            }
            if (p->ilv) {
                if (hipMemcpy(p->sotf + (size_t)a * p->KBP * LP * 2, row.data(), (size_t)2 * p->KBP * LP * sizeof(float),
                              hipMemcpyHostToDevice) != hipSuccess)
                    return bail(fail("sotf upload failed"));
            } else
            for (int c = 0; c < 2; ++c)
                if (hipMemcpy(p->sotf + ((size_t)c * p->PL + (size_t)a * p->KBP) * LP, row.data() + (size_t)c * p->KBP * LP,
                              (size_t)p->KBP * LP * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
                    return bail(fail("sotf upload failed"));
        }
    }
    if (p->T > 0) {
        std::vector<float> t((size_t)p->T * LP, 0.f);
        for (int k = 0; k < p->T; ++k)
            for (int l = 0; l < p->Lown; ++l)
                if (p->planes[l] >= 0) t[(size_t)k * LP + l] = (float)cfg->templates[(size_t)k * p->Lc + p->planes[l]];
        if (dev_upload(&p->tpl, t)) return bail(1);
    }
    {
        std::vector<float> Fi, Gi, Gf, Ff, GiT, GfT;
        build_dft(p, Fi, Gi, Gf, Ff, GiT, GfT);
        if (dev_upload(&p->Fi, Fi) || dev_upload(&p->Gi, Gi) || dev_upload(&p->Gf, Gf) || dev_upload(&p->Ff, Ff) ||
            dev_upload(&p->GiT, GiT) || dev_upload(&p->GfT, GfT))
            return bail(1);
    }
    {   // folded-DFT matrices
        const char *e = getenv("SURFH_DFT_DENSE");
        p->dense_dft = e && e[0] == '1';
        const char *e3 = getenv("SURFH_NO_FUSED_MIX");
        p->fuse_mix = !(e3 && e3[0] == '1');
        const char *e4 = getenv("SURFH_WBLUR_FP32");
        p->wblur_fp32 = e4 && e4[0] == '1';       // R / R^T on the fp32-input MFMA instead of the split-bf16 path
        // measured on config 3: 7.64 -> 7.53 ms per iteration (+1.5 %), the overlapped kernels slow each other down by
        // almost what they save; off by default so that per-kernel times in profiles stay those of a kernel running alone
        const char *e7 = getenv("SURFH_OVERLAP");
        p->overlap = e7 && e7[0] == '1';
        if (p->overlap && hipStreamCreateWithFlags(&p->stream2, hipStreamNonBlocking) != hipSuccess) return bail(fail("hipStreamCreate failed"));
        if (const char *ep = getenv("SURFH_OTF_PROD")) p->otf_prod = !(ep[0] == '0');
        const char *e15 = getenv("SURFH_GATHER_GROUPED");
        p->gather_grouped = !(e15 && e15[0] == '0');
        { const char *eg = getenv("SURFH_GEMM_GROUPED"); p->gemm_grouped = !(eg && eg[0] == '0'); }
        const char *e14 = getenv("SURFH_SCATTER_GROUPED");
        p->scatter_grouped = !(e14 && e14[0] == '0');
        const char *e12 = getenv("SURFH_GATHER_SORTED");
        p->gather_sorted = !(e12 && e12[0] == '0');   // 0: gather rows in (pointing, alpha, beta) order
        const int ha = p->Na / 2 + 1, hb = p->Nb / 2 + 1;
        p->MPa = (ha + 127) / 128 * 128; p->KPa = (ha + 15) / 16 * 16;
        p->MPb = (hb + 127) / 128 * 128; p->KPb = (hb + 15) / 16 * 16;
        if (p->KPa > p->NAP || p->KPb > p->NBP || p->KPb > p->KBP) return bail(fail("cube too small for the folded DFT"));
        const double sa = 1.0 / std::sqrt((double)p->Na), sb = 1.0 / std::sqrt((double)p->Nb);
        std::vector<float> Cma((size_t)p->MPa * p->KPa, 0.f), Sma(Cma.size(), 0.f);
        for (int r = 0; r < ha; ++r)
            for (int k = 0; k < ha; ++k) {
                const double th = 2.0 * M_PI * (double)(((long)r * k) % p->Na) / (double)p->Na;
                Cma[(size_t)r * p->KPa + k] = (float)(std::cos(th) * sa);
                Sma[(size_t)r * p->KPa + k] = (float)(std::sin(th) * sa);
            }
        std::vector<float> Gc((size_t)p->MPb * p->KPb, 0.f), Gs(Gc.size(), 0.f), Cf(Gc.size(), 0.f), Sf(Gc.size(), 0.f);
        for (int b = 0; b < hb; ++b)
            for (int k = 0; k < hb; ++k) {
                const double th = 2.0 * M_PI * (double)(((long)b * k) % p->Nb) / (double)p->Nb;
                const double w = (k == 0 || (p->Nb % 2 == 0 && k == p->Nb / 2)) ? 1.0 : 2.0;
                Gc[(size_t)b * p->KPb + k] = (float)(w * std::cos(th) * sb);     // rows beta, cols k_beta
                Gs[(size_t)b * p->KPb + k] = (float)(w * std::sin(th) * sb);
                Cf[(size_t)k * p->KPb + b] = (float)(std::cos(th) * sb);         // rows k_beta, cols beta
                Sf[(size_t)k * p->KPb + b] = (float)(std::sin(th) * sb);
            }
        if (dev_upload(&p->Cma, Cma) || dev_upload(&p->Sma, Sma) || dev_upload(&p->Gc, Gc) || dev_upload(&p->Gs, Gs) ||
            dev_upload(&p->Cf, Cf) || dev_upload(&p->Sf, Sf))
            return bail(1);
        if (cfg->verify) {      // verification plan: dense DFT products, unfused spectral mix, fp32-operand spectral blur -- all float64-accumulated
            p->verify = true;
            p->dense_dft = true;
            p->fuse_mix = false;
            p->wblur_fp32 = true;
            p->overlap = false;
        }
        if (p->ax_a == 1 || p->ax_b == 1) {   // LDS images of the matrix pairs of the axes that run on dft_h2 (dft_h2.h)
            if ((p->ax_a == 1 && p->MPa != 128) || (p->ax_b == 1 && p->MPb != 128)) return bail(fail("internal: dft_h2 needs 128-row folded matrices"));
            std::vector<unsigned short> im(3 * DFT_H2_IMAGE_HALFS, 0);
            if (p->ax_a == 1) p->h2kA[0] = dft_h2_build_image(Cma.data(), Sma.data(), p->MPa, p->KPa, p->KPa, im.data());
            if (p->ax_b == 1) {
                p->h2kA[1] = dft_h2_build_image(Gc.data(), Gs.data(), p->MPb, p->KPb, p->KPb, im.data() + DFT_H2_IMAGE_HALFS);
                p->h2kA[2] = dft_h2_build_image(Cf.data(), Sf.data(), p->MPb, p->KPb, p->KPb, im.data() + 2 * DFT_H2_IMAGE_HALFS);
            }
            if (dev_upload(&p->h2img, im)) return bail(1);
        }
        if (p->h2) {
            // fused adjoint tail: the last pass of rfft2 multiplies by conj(sotf) and reduces over the wavelengths itself
            // (needs T <= 4 templates and 127 <= Na <= 255 output rows; SURFH_ADJ_FUSED=0: separate pass + reduction)
            const char *eaf = getenv("SURFH_ADJ_FUSED");
            if (!(eaf && eaf[0] == '0') && p->T >= 1 && p->T <= 4 && p->Na >= 127 && p->Na <= 255 && p->LP % 128 == 0) {
                if (otf_support(p, cfg)) return bail(1);
                const size_t npart = dft_h2_adjmix_part_floats(p->LP, p->Nb / 2 + 1, p->otf_nvalid);
                if (npart && dev_alloc(&p->adjmix_part, npart)) return bail(1);
            }
        }
        if (p->ct) {   // image + twiddles per transform length of the axes that run on dft_ct (dft_ct.h)
            if (p->ax_a == 2 && dft_ct_plan_create(p->Na, &p->ctA)) return bail(fail("dft_ct plan (n_alpha = %d) failed", p->Na));
            if (p->ax_b == 2) {
                if (p->ax_a == 2 && p->Nb == p->Na) p->ctB = p->ctA;
                else if (dft_ct_plan_create(p->Nb, &p->ctB)) return bail(fail("dft_ct plan (n_beta = %d) failed", p->Nb));
            }
            if (p->ax_a == 2 && p->ax_b == 2 && p->T >= 1 && p->T <= 4 && otf_support(p, cfg)) return bail(1);      // lists: both axes on dft_ct
        }
        if (!p->ilv) p->dense_dft = true;        // no fast kernel for one of the axes: dense fp32 products
    }
    // ---- work buffers ---------------------------------------------------------------------
    const size_t nspec = (size_t)2 * p->PL * LP, ncube = (size_t)p->NBP * p->NAP * LP;
    const size_t nycol = (size_t)2 * p->NAP * p->KBP * LP;
    const size_t nmhat = p->T > 0 ? (size_t)p->T * 2 * p->PL : nspec;
    const size_t nmaps = (size_t)std::max(p->T, 1) * p->PLc;
    const size_t nycm = (size_t)std::max(p->T, 1) * 2 * p->NAP * p->KBP;
    if (dev_alloc(&p->spec, nspec) || dev_alloc(&p->ycol, nycol) || dev_alloc(&p->cube, ncube) ||
        dev_alloc(&p->mhat, nmhat) || dev_alloc(&p->maps_pad, nmaps) || dev_alloc(&p->ycol_maps, nycm))
        return bail(1);
    hipMemset(p->spec, 0, nspec * sizeof(float));
    hipMemset(p->ycol, 0, nycol * sizeof(float));
    hipMemset(p->cube, 0, ncube * sizeof(float));
    hipMemset(p->mhat, 0, nmhat * sizeof(float));
    hipMemset(p->maps_pad, 0, nmaps * sizeof(float));
    hipMemset(p->ycol_maps, 0, nycm * sizeof(float));
    // ---- channels -------------------------------------------------------------------------
    p->a_lo = p->b_lo = 1 << 30; p->a_hi = p->b_hi = 0;      // alpha / beta range of the pixels the channels' tables touch (build_channel)
    p->ch.resize(cfg->n_channels);
    long yoff = 0;
    for (int i = 0; i < cfg->n_channels; ++i) {
        if (build_channel(p, cfg->channels[i], &p->ch[i])) return bail(1);
        Channel &c = p->ch[i];
        c.yoff = yoff;
        yoff += c.ysize;
        if (c.bsum) continue;
        const bool f16 = !p->wblur_fp32;          // spectral-blur GEMMs as two-piece fp16 products (gemm_cc16.hip)
        c.splitK = p->verify ? 1 : pick_split(c, cfg->split_k_forward, f16, p->n_cu);
        if (f16) {
            const long nw = (long)c.LdetP * c.K;
            const long nwv = ymat_from_y_waves(c.P * c.S, c.Ldet, c.aout);
            if (dev_alloc(&c.W16, (size_t)2 * nw) || dev_alloc(&c.Wt16, (size_t)2 * nw) || dev_alloc(&c.amax, (size_t)c.NP) ||
                dev_alloc(&c.pmax, (size_t)nwv) || dev_alloc(&c.ymat16, (size_t)2 * c.NP * c.LdetP) || dev_alloc(&c.Xs16, (size_t)2 * c.NP * c.K))
                return bail(1);
            hipMemset(c.pmax, 0, (size_t)nwv * sizeof(unsigned));       // entries of workgroups that exit early stay 0
            hipMemset(c.amax, 0, (size_t)c.NP * sizeof(unsigned));
            hipMemset(c.Xs16, 0, (size_t)2 * c.NP * c.K * sizeof(unsigned short));       // padding rows / columns stay zero
            std::vector<float> ones((size_t)c.nbs * ((c.LinP + 1023) / 1024) * c.NP, 1.f);   // segments never written: scale 1
            if (dev_upload(&c.bscale, ones)) return bail(1);
            if (launch_split2h(p->stream, c.W, c.W16, nw, nw, c.sW) || launch_split2h(p->stream, c.Wt, c.Wt16, nw, nw, c.sW))
                return bail(fail("operand split failed"));
            const bool far_steps = !(cfg->exact & 1) && [] { const char *e = getenv("SURFH_WBLUR_FAR"); return !(e && e[0] == '0'); }();      // read at plan creation
            const double far_tol2 = [] { const char *e = getenv("SURFH_WBLUR_FAR_TOL2"); return std::ldexp(1.0, -(e ? atoi(e) : 10)); }();
            const int segChunks = (c.LinP + 1023) / 1024, KA = (c.Ldet + 31) / 32 * 32;
            if (far_steps && c.K / 32 <= 2048 && c.nbs * segChunks <= 64 && KA / 32 <= 2048) {
                std::vector<float> hw((size_t)nw);
                std::vector<int> kl;
                if (hipMemcpy(hw.data(), c.W, (size_t)nw * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return bail(fail("copy failed"));
                build_klist(hw.data(), c.LdetP, c.K, c.K, c.LinP, segChunks, 1.0 / 256, far_tol2, &kl, &c.klFs, &c.ksteps[0], &c.ksteps[1]);
                if (c.ksteps[1] > 0 && dev_upload(&c.klF, kl)) return bail(1);
                if (hipMemcpy(hw.data(), c.Wt, (size_t)nw * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return bail(fail("copy failed"));
                // the adjoint's constant operand has one row per (beta column, wavelength): a tile of 64 wavelengths of four
                // neighbouring columns sees the response's diagonal in 2-3 of its K steps, 256 wavelengths of one column in 9
                const bool perm = [] { const char *e = getenv("SURFH_WBLUR_PERM"); return !(e && e[0] == '0'); }();
                const int pP = perm && c.LinP % 64 == 0 && c.nbs >= 4 ? 4 : 0;
                build_klist(hw.data(), c.K, KA, c.LdetP, 0, 0, 1.0 / 256, far_tol2, &kl, &c.klAs, &c.ksteps[2], &c.ksteps[3], pP, c.LinP);
                if (c.ksteps[3] > 0) {
                    if (dev_upload(&c.klA, kl)) return bail(1);
                    c.permA = pP;
                }
            }
        }
        if (dev_alloc(&c.Cpart, (size_t)c.splitK * c.LdetP * c.NP)) return bail(1);
        hipMemset(c.Cpart, 0, (size_t)c.splitK * c.LdetP * c.NP * sizeof(float));
    }
    p->osize = yoff;
    if (p->a_hi <= p->a_lo) { p->a_lo = 0; p->a_hi = p->Na; }
    if (p->b_hi <= p->b_lo) { p->b_lo = 0; p->b_hi = p->Nb; }
    {   // transform passes batched over alpha skip the columns no table touches (SURFH_ALPHA_RANGE=0: whole cube)
        const char *ear = getenv("SURFH_ALPHA_RANGE");
        if (ear && ear[0] == '0') { p->a_lo = 0; p->a_hi = p->Na; p->b_lo = 0; p->b_hi = p->Nb; }
        if (p->ilv && p->a_hi - p->a_lo < p->Na) {      // the adjoint's intermediate: columns outside the range zero for good
            const size_t nyc = (size_t)2 * p->NAP * p->KBP * p->LP;
            if (dev_alloc(&p->ycol_adj, nyc)) return bail(1);
            if (hipMemset(p->ycol_adj, 0, nyc * sizeof(float)) != hipSuccess) return bail(fail("memset failed"));
        }
    }
    {   // The adjoint scatters channel after channel into the cleared cube.  A (pixel, 1024-wavelength chunk) of channel c
        // must be read-modify-written only if an earlier channel's table has that pixel and its window reaches into the
        // chunk; everywhere else the destination is still zero and the kernel stores without reading (the windows of
        // adjacent bands overlap by a tenth, so most of the traffic is of the second kind).
        const long npixrows = (long)p->NBP * p->NAP;
        bool exact_ok = !p->verify;
        {
            const char *ec = getenv("SURFH_ADJ_CLEAR");
            if (ec && ec[0] == '1') exact_ok = false;      // A/B: clear the accumulator every call, chunk masks
        }
        std::vector<std::vector<uint8_t>> touched(p->ch.size());
        for (size_t ci = 0; ci < p->ch.size(); ++ci) {
            Channel &c = p->ch[ci];
            const std::vector<int64_t> &dst = c.adjT.host_dst;
            touched[ci].assign((size_t)npixrows, 0);
            const int nchunk = (c.nlam / 4 + 255) / 256;
            std::vector<uint32_t> mask(dst.size(), nchunk > 32 ? 0xFFFFFFFFu : 0u);
            std::vector<int2> ranges(dst.size(), make_int2(0, 0));
            for (size_t r = 0; r < dst.size(); ++r) {
                const long pix = (dst[r] - c.ws0a) / p->LP;
                if (pix < 0 || pix >= npixrows) return bail(fail("scatter table: bad destination"));
                touched[ci][pix] = 1;
                {   // exact form: union of the earlier channels' windows at this pixel, inside this channel's window
                    int lo = INT32_MAX, hi = INT32_MIN, covered = 0;
                    for (size_t cj = 0; cj < ci; ++cj) {
                        if (!touched[cj][pix]) continue;
                        const Channel &o = p->ch[cj];
                        const int a = std::max(o.ws0a, c.ws0a), b = std::min(o.ws0a + o.nlam, c.ws0a + c.nlam);
                        if (a >= b) continue;
                        if (covered && (a > hi || b < lo)) exact_ok = false;      // two separate pieces: not one range
                        lo = std::min(lo, a);
                        hi = std::max(hi, b);
                        covered = 1;
                    }
                    if (covered) ranges[r] = make_int2(lo - c.ws0a, hi - c.ws0a);
                }
                if (nchunk > 32) continue;
                for (size_t cj = 0; cj < ci; ++cj) {
                    if (!touched[cj][pix]) continue;
                    const Channel &o = p->ch[cj];
                    for (int j = 0; j < nchunk; ++j) {
                        const int lo = c.ws0a + 1024 * j, hi = std::min(c.ws0a + c.nlam, lo + 1024);
                        if (lo < o.ws0a + o.nlam && o.ws0a < hi) mask[r] |= 1u << j;
                    }
                }
            }
            {
                const char *ea0 = getenv("SURFH_SCATTER_RMW_ALL");
                if (ea0 && ea0[0] == '1') std::fill(mask.begin(), mask.end(), 0xFFFFFFFFu);
            }
            if (p->scatter_grouped) {
                // rows are in pixel order: take runs of neighbouring pixels (destinations LP apart), SCATTER_G at a time
                const HostEll &h = c.adjT_host;
                std::vector<std::pair<size_t, size_t>> runs;
                for (size_t r = 0; r < h.rows.size();) {
                    size_t n = 1;
                    while (n < (size_t)SCATTER_G && r + n < h.rows.size() && h.dst[r + n] == h.dst[r + n - 1] + p->LP) ++n;
                    runs.push_back({r, n});
                    r += n;
                }
                if (upload_groups(h, runs, &mask, &c.adjT, SCATTER_G, &ranges)) return bail(1);
            }
            c.adjT_host = HostEll();
            if (dev_upload(&c.adjT.rmw, mask) || dev_upload(&c.adjT.rng, ranges)) return bail(1);
            c.adjT.t.rng = c.adjT.rng;
            const char *ea = getenv("SURFH_SCATTER_RMW_ALL");
            if (!(ea && ea[0] == '1')) c.adjT.t.rmw = c.adjT.rmw;        // 1: read-modify-write everywhere (A/B)
            c.adjT.host_dst.clear();
            c.adjT.host_dst.shrink_to_fit();
        }
        if (exact_ok && !p->ch.empty()) {      // dedicated accumulator, cleared once: the exact ranges keep it consistent
            const size_t ncube = (size_t)p->NBP * p->NAP * p->LP;
            if (dev_alloc(&p->gcube, ncube)) return bail(1);
            if (hipMemset(p->gcube, 0, ncube * sizeof(float)) != hipSuccess) return bail(fail("memset failed"));
        } else {
            for (auto &c : p->ch) {
                c.adjT.t.rng = nullptr;
                c.adjT.g.rng = nullptr;
            }
        }
    }
    if (dev_alloc(&p->io_x, (size_t)p->isize) || dev_alloc(&p->io_y, (size_t)p->osize) || dev_alloc(&p->cg_y, (size_t)p->osize) ||
        dev_alloc(&p->dscal, 8) || dev_alloc(&p->dscratch, 1024))
        return bail(1);
    if (hipDeviceSynchronize() != hipSuccess) return bail(fail("device error during plan creation: %s", hipGetErrorString(hipGetLastError())));
    *out = p;
    return 0;
}

int64_t surfh_isize(const surfh_plan *p) { return p ? p->isize : -1; }
int64_t surfh_osize(const surfh_plan *p) { return p ? p->osize : -1; }
void *surfh_stream(const surfh_plan *p) { return p ? (void *)p->stream : nullptr; }

int surfh_forward_dev(surfh_plan *p, const float *x, float *y) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    return forward_dev(p, x, y);
}
int surfh_adjoint_dev(surfh_plan *p, const float *y, float *x) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    return adjoint_dev(p, y, x, false);
}
int surfh_adjoint_ref_dev(surfh_plan *p, const float *y, float *x) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    return adjoint_dev(p, y, x, true);
}
int surfh_fwadj_dev(surfh_plan *p, const float *x, float *out) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    return normal_dev(p, x, out, 1.0);
}

static int host_call(surfh_plan *p, const float *in, long nin, float *outp, long nout, int which) {
    if (!p || !in || !outp) return fail("null argument");
    HIP_OK(hipSetDevice(p->dev));
    float *din = (which == 0 || which == 3) ? p->io_x : p->io_y;
    float *dout = (which == 0) ? p->io_y : p->io_x;
    if (which == 3) {
        if (ensure_cg(p)) return 1;
        dout = p->cg_q;
    }
    HIP_OK(hipMemcpyAsync(din, in, nin * sizeof(float), hipMemcpyHostToDevice, p->stream));
    int rc = 0;
    if (which == 0) rc = forward_dev(p, din, dout);
    else if (which == 1) rc = adjoint_dev(p, din, dout, false);
    else if (which == 2) rc = adjoint_dev(p, din, dout, true);
    else rc = normal_dev(p, din, dout, 1.0);
    if (rc) return rc;
    HIP_OK(hipMemcpyAsync(outp, dout, nout * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));
    return 0;
}

int surfh_forward(surfh_plan *p, const float *maps, float *y) { return host_call(p, maps, p ? p->isize : 0, y, p ? p->osize : 0, 0); }
int surfh_adjoint(surfh_plan *p, const float *y, float *maps) { return host_call(p, y, p ? p->osize : 0, maps, p ? p->isize : 0, 1); }
int surfh_adjoint_ref(surfh_plan *p, const float *y, float *maps) { return host_call(p, y, p ? p->osize : 0, maps, p ? p->isize : 0, 2); }
int surfh_fwadj(surfh_plan *p, const float *x, float *o) { return host_call(p, x, p ? p->isize : 0, o, p ? p->isize : 0, 3); }

// ---- Model_WCT: the T.C stage alone, cube in the reference's [Lc][Na][Nb] layout ------------------
static int wct_check(surfh_plan *p) {
    if (!p) return fail("null plan");
    if (p->T < 1) return fail("Model_WCT needs templates");
    if (p->segs.size() != 1 || p->segs[0].start != 0 || p->segs[0].len != p->Lc) return fail("Model_WCT needs a plan that owns every cube plane");
    if (hipSetDevice(p->dev) != hipSuccess) return fail("hipSetDevice failed");
    const size_t n = (size_t)p->Lc * p->Na * p->Nb;
    if (!p->io_cube && dev_alloc(&p->io_cube, n)) return 1;
    return 0;
}

int surfh_wct_forward(surfh_plan *p, const float *maps, float *cube) {
    if (wct_check(p)) return 1;
    if (!maps || !cube) return fail("null argument");
    hipStream_t s = p->stream;
    HIP_OK(hipMemcpyAsync(p->io_x, maps, p->isize * sizeof(float), hipMemcpyHostToDevice, s));
    LAUNCH_OK(launch_pad_planes(s, p->io_x, p->maps_pad, p->T, p->Na, p->Nb, p->NAP, p->NBP));
    if (rfft2_planes(p, p->maps_pad, p->mhat, p->T)) return 1;
    LAUNCH_OK(launch_specmix_fwd(s, p->mhat, p->sotf, p->tpl, p->spec, p->T, p->PL, p->LP, p->ilv));
    if (irfft2_cube(p, p->spec, p->cube)) return 1;
    LAUNCH_OK(launch_cube_from_lam_inner(s, p->cube, p->io_cube, 0, p->Lc, p->Na, p->Nb, p->NAP, p->LP));
    HIP_OK(hipMemcpyAsync(cube, p->io_cube, (size_t)p->Lc * p->Na * p->Nb * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return 0;
}

int surfh_wct_adjoint(surfh_plan *p, const float *cube, float *maps) {
    if (wct_check(p)) return 1;
    if (!maps || !cube) return fail("null argument");
    hipStream_t s = p->stream;
    HIP_OK(hipMemcpyAsync(p->io_cube, cube, (size_t)p->Lc * p->Na * p->Nb * sizeof(float), hipMemcpyHostToDevice, s));
    LAUNCH_OK(launch_fill_zero(s, p->cube, (long)p->NBP * p->NAP * p->LP));
    LAUNCH_OK(launch_cube_to_lam_inner(s, p->io_cube, p->cube, 0, p->Lc, p->Na, p->Nb, p->NAP, p->LP));
    if (adjoint_tail(p, p->cube)) return 1;
    if (irfft2_planes(p, p->mhat, p->maps_pad, p->T)) return 1;
    LAUNCH_OK(launch_unpad_planes(s, p->maps_pad, p->io_x, p->T, p->Na, p->Nb, p->NAP, p->NBP));
    HIP_OK(hipMemcpyAsync(maps, p->io_x, p->isize * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return 0;
}

int surfh_wct_fwadj(surfh_plan *p, const float *x, float *out) {
    if (wct_check(p)) return 1;
    if (!x || !out) return fail("null argument");
    hipStream_t s = p->stream;
    if (ensure_hessian(p)) return 1;
    HIP_OK(hipMemcpyAsync(p->io_x, x, p->isize * sizeof(float), hipMemcpyHostToDevice, s));
    LAUNCH_OK(launch_pad_planes(s, p->io_x, p->maps_pad, p->T, p->Na, p->Nb, p->NAP, p->NBP));
    if (rfft2_planes(p, p->maps_pad, p->mhat, p->T)) return 1;
    LAUNCH_OK(launch_wct_hess_apply(s, p->hth, p->mhat, p->mhat2, p->T, p->PL));
    if (irfft2_planes(p, p->mhat2, p->maps_pad, p->T)) return 1;
    LAUNCH_OK(launch_unpad_planes(s, p->maps_pad, p->io_x, p->T, p->Na, p->Nb, p->NAP, p->NBP));
    HIP_OK(hipMemcpyAsync(out, p->io_x, p->isize * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return 0;
}

// explicit inverse of the regularised normal operator (QuadCriterion3.run_expsol, fusion_mixing.py:309-438)
int surfh_wct_expsol(surfh_plan *p, const float *cube, const double *mu_reg, const double *reg_freq, float *maps) {
    if (wct_check(p)) return 1;
    if (!maps || !cube || !mu_reg || !reg_freq) return fail("null argument");
    for (int t = 0; t < p->T; ++t)
        if (!(mu_reg[t] >= 0.0)) return fail("mu_reg[%d] must be >= 0", t);
    hipStream_t s = p->stream;
    if (ensure_hessian(p)) return 1;
    // |D(f)|^2 into the padded spectral layout [KAP][KBP]; -1 marks the padding bins
    const int hb = p->Nb / 2 + 1;
    std::vector<float> reg((size_t)p->PL, -1.f);
    for (int a = 0; a < p->Na; ++a)
        for (int b = 0; b < hb; ++b) {
            const double v = reg_freq[(size_t)a * hb + b];
            if (!(v >= 0.0)) return fail("reg_freq must be >= 0");
            reg[(size_t)a * p->KBP + b] = (float)v;
        }
    float *dreg = nullptr;
    double *dmu = nullptr;
    int *dflag = nullptr;
    auto done = [&](int r) { hipFree(dreg); hipFree(dmu); hipFree(dflag); return r; };
    if (dev_upload(&dreg, reg) || dev_alloc(&dmu, (size_t)p->T) || dev_alloc(&dflag, 1)) return done(1);
    if (hipMemcpy(dmu, mu_reg, p->T * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(dflag, 0, sizeof(int)) != hipSuccess) return done(fail("copy failed"));
    // b = H^T y in the Fourier domain (surfh_wct_adjoint up to the inverse transform)
    if (hipMemcpyAsync(p->io_cube, cube, (size_t)p->Lc * p->Na * p->Nb * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess)
        return done(fail("copy failed"));
    int rc = launch_fill_zero(s, p->cube, (long)p->NBP * p->NAP * p->LP);
    if (!rc) rc = launch_cube_to_lam_inner(s, p->io_cube, p->cube, 0, p->Lc, p->Na, p->Nb, p->NAP, p->LP);
    if (rc) return done(fail("launch failed: %s", hipGetErrorString((hipError_t)rc)));
    if (adjoint_tail(p, p->cube)) return done(1);
    rc = launch_wct_solve(s, p->hth, dreg, dmu, p->mhat, p->mhat2, p->T, p->PL, dflag);
    if (rc) return done(fail("launch failed: %s", hipGetErrorString((hipError_t)rc)));
    if (irfft2_planes(p, p->mhat2, p->maps_pad, p->T)) return done(1);
    rc = launch_unpad_planes(s, p->maps_pad, p->io_x, p->T, p->Na, p->Nb, p->NAP, p->NBP);
    int flag = 0;
    if (!rc) rc = (int)hipMemcpyAsync(maps, p->io_x, p->isize * sizeof(float), hipMemcpyDeviceToHost, s);
    if (!rc) rc = (int)hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, s);
    if (!rc) rc = (int)hipStreamSynchronize(s);
    if (rc) return done(fail("expsol failed: %s", hipGetErrorString((hipError_t)rc)));
    if (flag) return done(fail("the regularised normal matrix is singular at some frequency (numpy.linalg.inv would raise LinAlgError)"));
    return done(0);
}

// ---- CG building blocks ---------------------------------------------------------------------
int surfh_normal_dev(surfh_plan *p, const float *d, float *q, double mu) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    return normal_dev(p, d, q, mu);
}
int surfh_prior_add_dev(surfh_plan *p, const float *d, float *q, double mu_reg) {
    if (!p) return fail("null plan");
    if (p->T <= 0) return fail("prior is defined on abundance maps (needs templates)");
    HIP_OK(hipSetDevice(p->dev));
    Prof pr(p, "prior_add");
    LAUNCH_OK(prior_add(p, p->stream, d, q, p->T, (float)mu_reg));
    return 0;
}
// ---- the normal operator on the maps' half spectra (the solver's vectors live in the Fourier domain) -------------------------
// A vector is [T][2 (re, im)][KAP][KBP] floats (padding zero), bin (ka, kb) multiplied by sqrt(2) unless it is its own conjugate
// (kb = 0, or 2 kb = Nb): the transforms are unitary, so plain dot products of such vectors are the dot products of the maps.
namespace {
int spec_check(surfh_plan *p) {
    if (!p) return fail("null plan");
    if (!(((p->adjmix_part && p->h2) || p->ct) && p->T > 0 && p->T <= 4 && p->fuse_mix && !p->dense_dft && !p->verify))
        return fail("spectral-domain calls need the fused transform passes (dft_h2 with the fused adjoint tail, or dft_ct)");
    HIP_OK(hipSetDevice(p->dev));
    return 0;
}
struct SpecScope {      // the transient pointers never outlive a call
    surfh_plan *p;
    ~SpecScope() { p->spec_in = nullptr; p->spec_out = nullptr; p->spec_prior_src = nullptr; p->spec_mu = 1.f; p->spec_prior_mu = 0.f; }
};
}  // namespace

int surfh_spec_supported(surfh_plan *p) {
    return p && ((p->adjmix_part && p->h2) || p->ct) && p->T > 0 && p->T <= 4 && p->fuse_mix && !p->dense_dft && !p->verify && p->prior_kind == 0;
}
int64_t surfh_spec_size(surfh_plan *p) { return p ? (int64_t)2 * p->T * p->PL : 0; }

// xt = scaled half spectra of the maps x [T][Na][Nb]
int surfh_to_spec_dev(surfh_plan *p, const float *x, float *xt) {
    if (spec_check(p)) return 1;
    hipStream_t s = p->stream;
    LAUNCH_OK(launch_pad_planes(s, x, p->maps_pad, p->T, p->Na, p->Nb, p->NAP, p->NBP));
    if (rfft2_planes(p, p->maps_pad, p->mhat, p->T)) return 1;
    LAUNCH_OK(launch_spec_scale(s, p->mhat, xt, 2 * p->T, p->PL, p->KBP, p->Nb, 1.f, 1.41421356237309505f));
    return 0;
}
// x = maps of the scaled half spectra xt
int surfh_from_spec_dev(surfh_plan *p, const float *xt, float *x) {
    if (spec_check(p)) return 1;
    hipStream_t s = p->stream;
    LAUNCH_OK(launch_spec_scale(s, xt, p->mhat, 2 * p->T, p->PL, p->KBP, p->Nb, 1.f, 0.70710678118654752f));
    if (irfft2_planes(p, p->mhat, p->maps_pad, p->T)) return 1;
    LAUNCH_OK(launch_unpad_planes(s, p->maps_pad, x, p->T, p->Na, p->Nb, p->NAP, p->NBP));
    return 0;
}
// y = A maps(dt)
int surfh_forward_spec_dev(surfh_plan *p, const float *dt, float *y) {
    if (spec_check(p)) return 1;
    SpecScope sc{p};
    p->spec_in = dt;
    return forward_dev(p, nullptr, y);
}
// qt = mu * spectra(A^T y)  (+ mu_reg * prior(dt) when dt != NULL: only where q is not summed over ranks afterwards)
int surfh_adjoint_spec_dev(surfh_plan *p, const float *y, float *qt, double mu, const float *dt, double mu_reg) {
    if (spec_check(p)) return 1;
    if (dt && p->prior_kind != 0) return fail("the fused spectral prior is the separated first differences");
    SpecScope sc{p};
    p->spec_out = qt; p->spec_mu = (float)mu; p->spec_prior_src = dt; p->spec_prior_mu = dt ? (float)mu_reg : 0.f;
    return adjoint_dev(p, y, nullptr, false);
}
// qt = mu * spectra(A^T A maps(dt)) (+ mu_reg * prior(dt) if mu_reg != 0): the CG's normal operator without a single transform
// of the maps -- no padding, no small DFTs, no prior kernel
int surfh_normal_spec_dev(surfh_plan *p, const float *dt, float *qt, double mu, double mu_reg) {
    if (spec_check(p)) return 1;
    if (mu_reg != 0.0 && p->prior_kind != 0) return fail("the fused spectral prior is the separated first differences");
    SpecScope sc{p};
    p->spec_in = dt;
    p->spec_out = qt; p->spec_mu = (float)mu; p->spec_prior_src = mu_reg != 0.0 ? dt : nullptr; p->spec_prior_mu = (float)mu_reg;
    const bool ho = normal_hand_over(p);
    if (forward_dev(p, nullptr, p->cg_y, ho)) return 1;
    return adjoint_dev(p, p->cg_y, nullptr, false, ho);
}
// qt += mu_reg * prior(dt) on scaled half spectra (after an all-reduce of qt over ranks)
int surfh_prior_spec_add_dev(surfh_plan *p, const float *dt, float *qt, double mu_reg) {
    if (spec_check(p)) return 1;
    if (p->prior_kind != 0) return fail("the spectral prior is the separated first differences");
    LAUNCH_OK(launch_spec_prior_add(p->stream, dt, qt, 2 * p->T, p->Na, p->Nb, p->PL, p->KBP, (float)mu_reg));
    return 0;
}

int surfh_set_prior(surfh_plan *p, int32_t kind) {
    if (!p) return fail("null plan");
    if (kind != 0 && kind != 1) return fail("prior kind %d: 0 = separated first differences, 1 = joint Laplacian", (int)kind);
    p->prior_kind = kind;
    return 0;
}
int surfh_dot_dev(surfh_plan *p, const float *a, const float *b, int64_t n, double *out) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    LAUNCH_OK(launch_dot(p->stream, a, b, n, p->dscratch, p->dscal + 7));
    HIP_OK(hipMemcpyAsync(out, p->dscal + 7, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));
    return 0;
}
int surfh_cg_step_dev(surfh_plan *p, float *x, float *r, const float *d, const float *q, int64_t n, double rr_in,
                      double *rr_out) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    HIP_OK(hipMemcpyAsync(p->dscal + 0, &rr_in, sizeof(double), hipMemcpyHostToDevice, p->stream));
    LAUNCH_OK(launch_dot(p->stream, d, q, n, p->dscratch, p->dscal + 1));
    LAUNCH_OK(launch_cg_step(p->stream, x, r, d, q, n, p->dscal + 0, p->dscal + 1, p->dscratch, p->dscal + 2));
    HIP_OK(hipMemcpyAsync(rr_out, p->dscal + 2, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));
    return 0;
}
int surfh_cg_dir_dev(surfh_plan *p, float *d, const float *r, int64_t n, double beta) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    const double one = 1.0;
    HIP_OK(hipMemcpyAsync(p->dscal + 3, &beta, sizeof(double), hipMemcpyHostToDevice, p->stream));
    HIP_OK(hipMemcpyAsync(p->dscal + 4, &one, sizeof(double), hipMemcpyHostToDevice, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));   // host scalars are stack variables
    LAUNCH_OK(launch_cg_dir(p->stream, d, r, n, p->dscal + 3, p->dscal + 4));
    return 0;
}
// cg_step + cg_dir in one call with one host synchronisation: x += s d, r -= s q, rr' = r.r, d = r + (rr'/rr) d
int surfh_cg_iter_dev(surfh_plan *p, float *x, float *r, float *d, const float *q, int64_t n, double rr_in, double *rr_out) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    HIP_OK(hipMemcpyAsync(p->dscal + 0, &rr_in, sizeof(double), hipMemcpyHostToDevice, p->stream));
    LAUNCH_OK(launch_dot(p->stream, d, q, n, p->dscratch, p->dscal + 1));
    LAUNCH_OK(launch_cg_step(p->stream, x, r, d, q, n, p->dscal + 0, p->dscal + 1, p->dscratch, p->dscal + 2));
    LAUNCH_OK(launch_cg_dir(p->stream, d, r, n, p->dscal + 2, p->dscal + 0));
    HIP_OK(hipMemcpyAsync(rr_out, p->dscal + 2, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));   // also covers the pageable rr_in copy
    return 0;
}
// ---- the same blocks with every scalar kept on the device: nothing here synchronises with the host.  The trace cg_hist IS the
// scalar store: r.r of the current iterate is its last entry, an iteration reads it there and writes the next entry (read back
// with surfh_cg_trace).  An iteration is three launches -- partial sums of d.q; step (sums them, leaves partial sums of the new
// r.r); direction (sums those) -- and no copies (round 2: five launches and two 8-byte device-to-device copies, 42 us of an
// iteration's 2.87 ms on config 3).  For the multi-GPU loop: the only other work of an iteration is the normal operator and
// the all-reduce, both asynchronous on the plan's stream.
static constexpr int CG_HIST_CAP = 1 << 16;
static int cg_hist_room(surfh_plan *p) {
    if (!p->cg_hist && dev_alloc(&p->cg_hist, (size_t)CG_HIST_CAP)) return 1;
    if (p->cg_hist_n >= CG_HIST_CAP) return fail("CG trace full (%d iterations): read it with surfh_cg_trace", CG_HIST_CAP);
    return 0;
}
int surfh_cg_begin_dev(surfh_plan *p, const float *r, int64_t n) {         /* rr = r.r; trace restarts with it */
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    p->cg_hist_n = 0;
    if (cg_hist_room(p)) return 1;
    LAUNCH_OK(launch_dot(p->stream, r, r, n, p->dscratch, p->cg_hist + 0));
    p->cg_hist_n = 1;
    return 0;
}
int surfh_cg_iter_nosync_dev(surfh_plan *p, float *x, float *r, float *d, const float *q, int64_t n) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    if (p->cg_hist_n < 1) return fail("surfh_cg_iter_nosync_dev before surfh_cg_begin_dev");
    if (cg_hist_room(p)) return 1;
    double *const rr = p->cg_hist + p->cg_hist_n - 1, *const pa = p->dscratch, *const pb = p->dscratch + dot_parts_stride();
    LAUNCH_OK(launch_dot_parts(p->stream, d, q, n, pa));
    LAUNCH_OK(launch_cg_step_parts(p->stream, x, r, d, q, n, rr, pa, p->dscal + 1, pb));
    LAUNCH_OK(launch_cg_dir_parts(p->stream, d, r, n, pb, rr, rr + 1));
    ++p->cg_hist_n;
    return 0;
}
/* the residual-refresh iteration of qmm.lcg in two halves around the caller's normal operator on x:
 * x += (rr / d.q) d   ...   r = b - q; rr' = r.r; d = r + (rr' / rr) d; rr = rr'                                   */
int surfh_cg_xupdate_nosync_dev(surfh_plan *p, float *x, const float *d, const float *q, int64_t n) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    if (p->cg_hist_n < 1) return fail("surfh_cg_xupdate_nosync_dev before surfh_cg_begin_dev");
    LAUNCH_OK(launch_dot(p->stream, d, q, n, p->dscratch, p->dscal + 1));
    LAUNCH_OK(launch_cg_xupdate(p->stream, x, d, n, p->cg_hist + p->cg_hist_n - 1, p->dscal + 1));
    return 0;
}
int surfh_cg_refresh_nosync_dev(surfh_plan *p, float *r, const float *b, const float *q, float *d, int64_t n) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    if (p->cg_hist_n < 1) return fail("surfh_cg_refresh_nosync_dev before surfh_cg_begin_dev");
    if (cg_hist_room(p)) return 1;
    double *const rr = p->cg_hist + p->cg_hist_n - 1, *const pb = p->dscratch + dot_parts_stride();
    LAUNCH_OK(launch_residual(p->stream, r, b, q, n));
    LAUNCH_OK(launch_dot_parts(p->stream, r, r, n, pb));
    LAUNCH_OK(launch_cg_dir_parts(p->stream, d, r, n, pb, rr, rr + 1));
    ++p->cg_hist_n;
    return 0;
}
/* synchronises the plan's stream and copies the r.r trace (entry 0 = surfh_cg_begin_dev); returns the number of entries */
int32_t surfh_cg_trace(surfh_plan *p, double *out, int32_t cap) {
    if (!p || !out) return -1;
    if (hipSetDevice(p->dev) != hipSuccess) return -1;
    const int n = p->cg_hist_n < cap ? p->cg_hist_n : cap;
    if (hipStreamSynchronize(p->stream) != hipSuccess) return -1;
    if (n > 0 && hipMemcpy(out, p->cg_hist, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
int surfh_residual_dev(surfh_plan *p, float *r, const float *b, const float *q, int64_t n) {
    if (!p) return fail("null plan");
    HIP_OK(hipSetDevice(p->dev));
    LAUNCH_OK(launch_residual(p->stream, r, b, q, n));
    return 0;
}

// ---- full CG on one GPU (qmm.lcg semantics, see oracle/surfh_oracle.py:lcg) -------------------
namespace {
// The loop bench.py times, behind the exported solver: vectors = the maps' Parseval-scaled half spectra (surfh_normal_spec_dev:
// no transform of the maps, no padding, no prior kernel inside an iteration), every scalar on the device
// (surfh_cg_iter_nosync_dev), and the host reads the r.r trace -- the stopping test of qmm.lcg -- only every CG_CHECK iterations:
// the loop may run up to CG_CHECK - 1 iterations past the one that met the tolerance (nit and x are those of the last iteration
// run, grad_norm holds every r.r).  With a callback installed the trace and the iterate go to the host after every iteration,
// as the callback's contract says.
constexpr int CG_CHECK = 8;
int cg_spectral(surfh_plan *p, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter, double tol, int32_t refresh,
                float *x, double *grad_norm, int32_t *nit, surfh_cg_callback callback, void *user) {
    std::vector<float> hx;
    if (callback) hx.resize((size_t)p->isize);
    hipStream_t s = p->stream;
    const long n = p->isize, nv = 2L * p->T * p->PL;
    HIP_OK(hipMemcpyAsync(p->io_y, y, p->osize * sizeof(float), hipMemcpyHostToDevice, s));
    if (surfh_adjoint_spec_dev(p, p->io_y, p->cg_b, mu, nullptr, 0.0)) return 1;           // b = mu A^T y
    if (x0) {
        HIP_OK(hipMemcpyAsync(p->io_x, x0, n * sizeof(float), hipMemcpyHostToDevice, s));
        if (surfh_to_spec_dev(p, p->io_x, p->cg_x) || surfh_normal_spec_dev(p, p->cg_x, p->cg_q, mu, mu_reg)) return 1;
        LAUNCH_OK(launch_residual(s, p->cg_r, p->cg_b, p->cg_q, nv));
    } else {                                                                                // r = b - Q 0
        LAUNCH_OK(launch_fill_zero(s, p->cg_x, nv));
        HIP_OK(hipMemcpyAsync(p->cg_r, p->cg_b, nv * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    HIP_OK(hipMemcpyAsync(p->cg_d, p->cg_r, nv * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (surfh_cg_begin_dev(p, p->cg_r, nv)) return 1;
    *nit = 0;
    for (int it = 0; it < max_iter; ++it) {
        if (surfh_normal_spec_dev(p, p->cg_d, p->cg_q, mu, mu_reg)) return 1;
        if (refresh > 0 && it % refresh == 0) {             // residual recomputed from scratch
            if (surfh_cg_xupdate_nosync_dev(p, p->cg_x, p->cg_d, p->cg_q, nv) || surfh_normal_spec_dev(p, p->cg_x, p->cg_q, mu, mu_reg) ||
                surfh_cg_refresh_nosync_dev(p, p->cg_r, p->cg_b, p->cg_q, p->cg_d, nv))
                return 1;
        } else if (surfh_cg_iter_nosync_dev(p, p->cg_x, p->cg_r, p->cg_d, p->cg_q, nv)) {
            return 1;
        }
        *nit = it + 1;
        if (!callback && (it + 1) % CG_CHECK != 0 && it + 1 != max_iter) continue;
        if (surfh_cg_trace(p, grad_norm, it + 2) != it + 2) return fail("CG trace read failed");     // synchronises
        if (callback) {
            if (surfh_from_spec_dev(p, p->cg_x, p->io_x)) return 1;
            HIP_OK(hipMemcpyAsync(hx.data(), p->io_x, n * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
            if (callback(user, it + 1, grad_norm, hx.data())) break;
            HIP_OK(hipSetDevice(p->dev));
        }
        if (std::sqrt(grad_norm[it + 1]) < (double)n * tol) break;
    }
    if (surfh_cg_trace(p, grad_norm, *nit + 1) != *nit + 1) return fail("CG trace read failed");
    if (surfh_from_spec_dev(p, p->cg_x, p->io_x)) return 1;
    HIP_OK(hipMemcpyAsync(x, p->io_x, n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return 0;
}
}  // namespace

int surfh_cg_cb(surfh_plan *p, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter, double tol,
                int32_t refresh, float *x, double *grad_norm, int32_t *nit, surfh_cg_callback callback, void *user) {
    if (!p || !y || !x || !grad_norm || !nit) return fail("null argument");
    if (p->T <= 0) return fail("surfh_cg needs templates (the priors act on abundance maps)");
    HIP_OK(hipSetDevice(p->dev));
    if (ensure_cg(p)) return 1;
    {   // SURFH_SPECTRAL_CG=0: vectors are the maps (the loop below)
        const char *e = getenv("SURFH_SPECTRAL_CG");
        if (!(e && e[0] == '0') && surfh_spec_supported(p) && max_iter < (1 << 16) - 1)
            return cg_spectral(p, y, mu, mu_reg, x0, max_iter, tol, refresh, x, grad_norm, nit, callback, user);
    }
    std::vector<float> hx;             // host copy of the iterate handed to the callback
    if (callback) hx.resize((size_t)p->isize);
    hipStream_t s = p->stream;
    const long n = p->isize;
    double *rr = p->dscal + 0, *dq = p->dscal + 1, *rrn = p->dscal + 2;
    auto Q = [&](const float *v, float *out) -> int {
        if (normal_dev(p, v, out, mu)) return 1;
        if (mu_reg != 0.0) {
            Prof pr(p, "prior_add");
            LAUNCH_OK(prior_add(p, s, v, out, p->T, (float)mu_reg));
        }
        return 0;
    };
    // b = mu A^T y
    HIP_OK(hipMemcpyAsync(p->io_y, y, p->osize * sizeof(float), hipMemcpyHostToDevice, s));
    if (adjoint_dev(p, p->io_y, p->cg_b, false)) return 1;
    if (mu != 1.0) LAUNCH_OK(launch_scale(s, p->cg_b, n, (float)mu));
    if (x0)
        HIP_OK(hipMemcpyAsync(p->cg_x, x0, n * sizeof(float), hipMemcpyHostToDevice, s));
    else
        LAUNCH_OK(launch_fill_zero(s, p->cg_x, n));
    if (Q(p->cg_x, p->cg_q)) return 1;
    LAUNCH_OK(launch_residual(s, p->cg_r, p->cg_b, p->cg_q, n));
    HIP_OK(hipMemcpyAsync(p->cg_d, p->cg_r, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    LAUNCH_OK(launch_dot(s, p->cg_r, p->cg_r, n, p->dscratch, rr));
    HIP_OK(hipMemcpyAsync(&grad_norm[0], rr, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    *nit = 0;
    for (int it = 0; it < max_iter; ++it) {
        if (Q(p->cg_d, p->cg_q)) return 1;
        LAUNCH_OK(launch_dot(s, p->cg_d, p->cg_q, n, p->dscratch, dq));
        if (refresh > 0 && it % refresh == 0) {
            LAUNCH_OK(launch_cg_xupdate(s, p->cg_x, p->cg_d, n, rr, dq));
            if (Q(p->cg_x, p->cg_q)) return 1;
            LAUNCH_OK(launch_residual(s, p->cg_r, p->cg_b, p->cg_q, n));
            LAUNCH_OK(launch_dot(s, p->cg_r, p->cg_r, n, p->dscratch, rrn));
        } else {
            Prof pr(p, "cg_step");
            LAUNCH_OK(launch_cg_step(s, p->cg_x, p->cg_r, p->cg_d, p->cg_q, n, rr, dq, p->dscratch, rrn));
        }
        LAUNCH_OK(launch_cg_dir(s, p->cg_d, p->cg_r, n, rrn, rr));
        HIP_OK(hipMemcpyAsync(rr, rrn, sizeof(double), hipMemcpyDeviceToDevice, s));
        HIP_OK(hipMemcpyAsync(&grad_norm[it + 1], rrn, sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        *nit = it + 1;
        if (callback) {
            // the work buffers hold nothing live between iterations, so the callback may run forward / adjoint on this plan
            HIP_OK(hipMemcpyAsync(hx.data(), p->cg_x, n * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
            if (callback(user, it + 1, grad_norm, hx.data())) break;
            HIP_OK(hipSetDevice(p->dev));
        }
        if (std::sqrt(grad_norm[it + 1]) < (double)n * tol) break;
    }
    HIP_OK(hipMemcpyAsync(x, p->cg_x, n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return 0;
}

int surfh_cg(surfh_plan *p, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter, double tol,
             int32_t refresh, float *x, double *grad_norm, int32_t *nit) {
    return surfh_cg_cb(p, y, mu, mu_reg, x0, max_iter, tol, refresh, x, grad_norm, nit, nullptr, nullptr);
}

// ---- 3MG (majorize-minimize memory gradient, qmm.mmmg; selected by method != 'lcg' at fusion_CT.py:194-198) ----
// For the quadratic objectives of this path the quadratic majorant is the criterion itself, so the MM step is the exact
// minimiser of the criterion over span{-grad, previous move}.  qmm solves the 2x2 system in the basis [-grad, move] with the
// operator applied to the gradient; in fp32 that form loses the conjugacy (the determinant r.Qr m.Qm - (r.Qm)^2 cancels) and
// was measured to converge visibly slower than CG.  The same subspace is therefore spanned by [d, m], d = r + beta m made
// Q-orthogonal to the previous move m with the carried image Qm, and the operator is applied to d: the 2x2 system
//   [[d.Qd, d.Qm], [d.Qm, m.Qm]] step = [d.r, m.r]
// is then nearly diagonal.  Same iterates in exact arithmetic, one normal-operator application per iteration; r is carried as
// r -= Q move and recomputed from scratch every `refresh` iterations.  numpy's pinv cut (1e-15 of the unscaled matrix),
// which in qmm drops the memory direction once |move|^2 / |grad|^2 < 1e-15, is not reproduced: the direction is dropped
// only when the scaled system is singular.
int surfh_mmmg(surfh_plan *p, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter, double tol,
               int32_t refresh, float *x, double *grad_norm, int32_t *nit, surfh_cg_callback callback, void *user) {
    if (!p || !y || !x || !grad_norm || !nit) return fail("null argument");
    if (p->T <= 0) return fail("surfh_mmmg needs templates (the priors act on abundance maps)");
    std::vector<float> hx;
    if (callback) hx.resize((size_t)p->isize);
    HIP_OK(hipSetDevice(p->dev));
    if (ensure_cg(p)) return 1;
    if (!p->cg_qm && (dev_alloc(&p->cg_qm, (size_t)p->isize) || dev_alloc(&p->cg_dd, (size_t)p->isize))) return 1;
    hipStream_t s = p->stream;
    const long n = p->isize;
    float *r = p->cg_r, *m = p->cg_d, *d = p->cg_dd, *qd = p->cg_q, *qm = p->cg_qm;
    auto Q = [&](const float *v, float *out) -> int {
        if (normal_dev(p, v, out, mu)) return 1;
        if (mu_reg != 0.0) {
            Prof pr(p, "prior_add");
            LAUNCH_OK(prior_add(p, s, v, out, p->T, (float)mu_reg));
        }
        return 0;
    };
    HIP_OK(hipMemcpyAsync(p->io_y, y, p->osize * sizeof(float), hipMemcpyHostToDevice, s));
    if (adjoint_dev(p, p->io_y, p->cg_b, false)) return 1;
    if (mu != 1.0) LAUNCH_OK(launch_scale(s, p->cg_b, n, (float)mu));
    if (x0)
        HIP_OK(hipMemcpyAsync(p->cg_x, x0, n * sizeof(float), hipMemcpyHostToDevice, s));
    else
        LAUNCH_OK(launch_fill_zero(s, p->cg_x, n));
    LAUNCH_OK(launch_fill_zero(s, m, n));
    LAUNCH_OK(launch_fill_zero(s, qm, n));
    if (Q(p->cg_x, qd)) return 1;
    LAUNCH_OK(launch_residual(s, r, p->cg_b, qd, n));
    double h[6];
    *nit = 0;
    for (int it = 0;; ++it) {
        // h0 = r.r (stopping quantity and trace entry), h1 = r.Qm, h2 = m.Qm
        LAUNCH_OK(launch_dot(s, r, r, n, p->dscratch, p->dscal + 0));
        LAUNCH_OK(launch_dot(s, r, qm, n, p->dscratch, p->dscal + 1));
        LAUNCH_OK(launch_dot(s, m, qm, n, p->dscratch, p->dscal + 2));
        HIP_OK(hipMemcpyAsync(h, p->dscal, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        grad_norm[it] = std::sqrt(h[0]);
        if (it > 0 && callback) {
            HIP_OK(hipMemcpyAsync(hx.data(), p->cg_x, n * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
            if (callback(user, it, grad_norm, hx.data())) break;
            HIP_OK(hipSetDevice(p->dev));
        }
        if (it >= max_iter || grad_norm[it] < (double)n * tol) break;
        const double mQm = h[2], beta = mQm > 0.0 ? -h[1] / mQm : 0.0;
        LAUNCH_OK(launch_lincomb(s, d, r, m, n, beta));
        if (Q(d, qd)) return 1;
        LAUNCH_OK(launch_dot(s, d, qd, n, p->dscratch, p->dscal + 3));
        LAUNCH_OK(launch_dot(s, d, qm, n, p->dscratch, p->dscal + 4));
        LAUNCH_OK(launch_dot(s, d, r, n, p->dscratch, p->dscal + 5));
        LAUNCH_OK(launch_dot(s, m, r, n, p->dscratch, p->dscal + 6));
        HIP_OK(hipMemcpyAsync(h, p->dscal + 3, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        const double dQd = h[0], dQm = h[1], dr = h[2], mr = h[3];
        if (!(dQd > 0.0)) return fail("3MG: non-positive curvature d.Qd = %g at iteration %d", dQd, it);
        double s0 = dr / dQd, s1 = 0.0;
        if (mQm > 0.0) {
            const double c = dQm / std::sqrt(dQd * mQm), det = 1.0 - c * c;      // scaled 2x2 system
            if (det > 1e-12) {
                s0 = (dr / dQd - c * mr / std::sqrt(dQd * mQm)) / det;
                s1 = (mr / mQm - c * dr / std::sqrt(dQd * mQm)) / det;
            }
        }
        const bool fresh = refresh > 0 && it % refresh == 0;
        {
            Prof pr(p, "mmmg_update");
            LAUNCH_OK(launch_mmmg_update(s, p->cg_x, r, d, m, qm, qd, n, s0, s1, fresh ? 0 : 1));
        }
        if (fresh) {
            if (Q(p->cg_x, qd)) return 1;
            LAUNCH_OK(launch_residual(s, r, p->cg_b, qd, n));
        }
        *nit = it + 1;
    }
    HIP_OK(hipMemcpyAsync(x, p->cg_x, n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return 0;
}

// ---- CG on independent planes: the 2-D deconvolution path (criterion_2D.py:60-250 per image, batched over wavelength)
int surfh_cg_planes_cb(surfh_plan *p, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter, double tol,
                       int32_t refresh, float *x, double *grad_norm, int32_t *nit, surfh_cg_callback callback, void *user) {
    if (!p || !y || !x || !grad_norm || !nit) return fail("null argument");
    std::vector<float> hx;             // host copy of the iterate handed to the callback
    if (callback) hx.resize((size_t)p->isize);
    if (p->T != 0) return fail("surfh_cg_planes is the solver of the plane-wise (no template) model; use surfh_cg with templates");
    if (p->ch.empty()) return fail("plan has no channel");
    HIP_OK(hipSetDevice(p->dev));
    if (ensure_cg(p)) return 1;
    hipStream_t s = p->stream;
    const int L = p->Lc;
    const long npix = (long)p->Na * p->Nb, n = p->isize;
    double *sc = nullptr;              // [3][L]: rr, dq, rr'
    HIP_OK(hipMalloc((void **)&sc, (size_t)3 * L * sizeof(double)));
    double *rr = sc, *dq = sc + L, *rrn = sc + 2 * L;
    auto done = [&](int rc) { hipFree(sc); return rc; };
    auto Q = [&](const float *v, float *out) -> int {
        if (normal_dev(p, v, out, mu)) return 1;
        if (mu_reg != 0.0) LAUNCH_OK(prior_add(p, s, v, out, L, (float)mu_reg));
        return 0;
    };
    if (hipMemcpyAsync(p->io_y, y, p->osize * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) return done(fail("copy failed"));
    if (adjoint_dev(p, p->io_y, p->cg_b, false)) return done(1);
    int rc = 0;
    if (mu != 1.0) rc = launch_scale(s, p->cg_b, n, (float)mu);
    if (!rc) rc = x0 ? (int)hipMemcpyAsync(p->cg_x, x0, n * sizeof(float), hipMemcpyHostToDevice, s) : launch_fill_zero(s, p->cg_x, n);
    if (rc) return done(fail("cg setup failed"));
    if (Q(p->cg_x, p->cg_q)) return done(1);
    rc = launch_residual(s, p->cg_r, p->cg_b, p->cg_q, n);
    if (!rc) rc = (int)hipMemcpyAsync(p->cg_d, p->cg_r, n * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (!rc) rc = launch_dot_planes(s, p->cg_r, p->cg_r, L, npix, rr);
    if (!rc) rc = (int)hipMemcpyAsync(grad_norm, rr, L * sizeof(double), hipMemcpyDeviceToHost, s);
    if (!rc) rc = (int)hipStreamSynchronize(s);
    if (rc) return done(fail("cg setup failed: %s", hipGetErrorString((hipError_t)rc)));
    *nit = 0;
    for (int it = 0; it < max_iter; ++it) {
        if (Q(p->cg_d, p->cg_q)) return done(1);
        rc = launch_dot_planes(s, p->cg_d, p->cg_q, L, npix, dq);
        if (!rc && refresh > 0 && it % refresh == 0) {     // residual recomputed from scratch (qmm.lcg restated, see surfh_cg)
            rc = launch_cg_step_planes(s, p->cg_x, p->cg_r, p->cg_d, p->cg_q, L, npix, rr, dq, rrn, 0);
            if (rc) return done(fail("launch failed"));
            if (Q(p->cg_x, p->cg_q)) return done(1);
            rc = launch_residual(s, p->cg_r, p->cg_b, p->cg_q, n);
            if (!rc) rc = launch_dot_planes(s, p->cg_r, p->cg_r, L, npix, rrn);
        } else if (!rc) {
            rc = launch_cg_step_planes(s, p->cg_x, p->cg_r, p->cg_d, p->cg_q, L, npix, rr, dq, rrn, 1);
        }
        if (!rc) rc = launch_cg_dir_planes(s, p->cg_d, p->cg_r, L, npix, rrn, rr);
        double *gn = grad_norm + (size_t)(it + 1) * L;
        if (!rc) rc = (int)hipMemcpyAsync(gn, rr, L * sizeof(double), hipMemcpyDeviceToHost, s);
        if (!rc) rc = (int)hipStreamSynchronize(s);
        if (rc) return done(fail("cg iteration failed: %s", hipGetErrorString((hipError_t)rc)));
        *nit = it + 1;
        if (callback) {      // qmm.lcg's per-iteration callback (criterion_2D.py:163-225): trace so far [it + 2][L], current iterate
            if (hipMemcpyAsync(hx.data(), p->cg_x, n * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                return done(fail("copy failed"));
            if (callback(user, it + 1, grad_norm, hx.data())) break;
            if (hipSetDevice(p->dev) != hipSuccess) return done(fail("hipSetDevice failed"));
        }
        double worst = 0.0;
        for (int l = 0; l < L; ++l) worst = std::max(worst, gn[l]);
        if (std::sqrt(worst) < (double)npix * tol) break;
    }
    rc = (int)hipMemcpyAsync(x, p->cg_x, n * sizeof(float), hipMemcpyDeviceToHost, s);
    if (!rc) rc = (int)hipStreamSynchronize(s);
    if (rc) return done(fail("copy failed"));
    return done(0);
}

int surfh_cg_planes(surfh_plan *p, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter, double tol,
                    int32_t refresh, float *x, double *grad_norm, int32_t *nit) {
    return surfh_cg_planes_cb(p, y, mu, mu_reg, x0, max_iter, tol, refresh, x, grad_norm, nit, nullptr, nullptr);
}

// ---- the same loop with the data and the iterate resident on the device and no host synchronisation inside: begin (b = mu A^T y,
// r = b - Q x, d = r), any number of step calls, r.r per plane on request.  x_dev stays the caller's buffer and holds the iterate.
namespace {
struct PnScope {                       // forward_dev / adjoint_dev read and write wavelength-innermost vectors for the duration of a call
    surfh_plan *p;
    explicit PnScope(surfh_plan *pl) : p(pl) { p->pn_native = true; }
    ~PnScope() { p->pn_native = false; }
};
// q = mu A^T A v (+ mu_reg prior, fused with the dot product v . q -> dq) on wavelength-innermost vectors
int pn_normal(surfh_plan *p, const float *v, float *q, double *dq) {
    PnScope sc(p);
    // with interleaved spectra and a prior weight the OTF product of the adjoint applies mu and adds the prior (adjoint_tail): the
    // two halves are called directly so that no scaling pass follows
    // (plans whose inverse transform forms the OTF product in its loader -- prod_capable -- take mu and the prior on that pass)
    const bool prod = prod_capable(p) && p->pl_mu != 0.0;
    const bool fold = prod || (p->ilv && !p->dense_dft && p->pl_mu_reg != 0.0);
    if (fold) {
        p->pn_fold_prior = true;
        const bool ho = normal_hand_over(p);
        const int rc = forward_dev(p, v, p->cg_y, ho) || adjoint_dev(p, p->cg_y, q, false, ho);
        p->pn_fold_prior = false;
        if (rc) return 1;
    } else if (normal_dev(p, v, q, p->pl_mu)) {
        return 1;
    }
    Prof pr(p, "pn_prior_dot");
    if (fold) LAUNCH_OK(launch_pn_dot(p->stream, v, q, p->Na, p->Nb, p->NAP, p->LP, p->pn_part, dq));
    else LAUNCH_OK(launch_pn_prior_dot(p->stream, v, q, p->Na, p->Nb, p->NAP, p->LP, (float)p->pl_mu_reg, p->pn_part, dq));
    return 0;
}
bool pn_capable(const surfh_plan *p) {
    const char *e = getenv("SURFH_PLANES_NATIVE");       // 0: vectors in the caller's [Lc][Na][Nb] layout (two transposes per operator application)
    return !(e && e[0] == '0') && p->T == 0 && p->segs.size() == 1 && p->segs[0].coff == 0 && p->segs[0].start == 0 && p->Lown == p->Lc && p->prior_kind == 0 &&
           p->LP % 64 == 0;
}
}  // namespace

int surfh_cg_planes_begin_dev(surfh_plan *p, const float *y_dev, double mu, double mu_reg, float *x_dev) {
    if (!p || !y_dev || !x_dev) return fail("null argument");
    if (p->T != 0) return fail("surfh_cg_planes is the solver of the plane-wise (no template) model; use surfh_cg with templates");
    if (p->ch.empty()) return fail("plan has no channel");
    HIP_OK(hipSetDevice(p->dev));
    hipStream_t s = p->stream;
    const int L = p->Lc;
    p->pn_active = pn_capable(p);
    if (p->pn_active) {
        // vectors in the cube's layout: the caller's x is transposed in here and out again at the end of every step call
        const size_t nc = (size_t)p->NBP * p->NAP * p->LP;
        for (float *&v : p->pn_v)
            if (!v) {
                if (dev_alloc(&v, nc)) return 1;
                HIP_OK(hipMemsetAsync(v, 0, nc * sizeof(float), s));       // the padding (rows >= Nb, columns >= Na, planes >= Lc) stays zero
            }
        if (!p->pn_sc && (dev_alloc(&p->pn_sc, (size_t)3 * p->LP) || dev_alloc(&p->pn_part, pn_part_doubles(p->LP)))) return 1;
        float *xn = p->pn_v[0], *r = p->pn_v[1], *d = p->pn_v[2], *q = p->pn_v[3], *b = p->pn_v[4];
        double *rr = p->pn_sc, *dq = p->pn_sc + p->LP;
        p->pl_x = x_dev; p->pl_mu = mu; p->pl_mu_reg = mu_reg; p->pl_it = 0;
        LAUNCH_OK(launch_cube_to_lam_inner(s, x_dev, xn, 0, L, p->Na, p->Nb, p->NAP, p->LP));
        {
            PnScope sc(p);
            if (adjoint_dev(p, y_dev, b, false)) return 1;
        }
        if (mu != 1.0) LAUNCH_OK(launch_scale(s, b, (long)nc, (float)mu));
        if (pn_normal(p, xn, q, dq)) return 1;
        LAUNCH_OK(launch_residual(s, r, b, q, (long)nc));
        HIP_OK(hipMemcpyAsync(d, r, nc * sizeof(float), hipMemcpyDeviceToDevice, s));
        LAUNCH_OK(launch_pn_dot(s, r, r, p->Na, p->Nb, p->NAP, p->LP, p->pn_part, rr));
        return 0;
    }
    if (ensure_cg(p)) return 1;
    const long npix = (long)p->Na * p->Nb, n = p->isize;
    if (!p->pl_sc) HIP_OK(hipMalloc((void **)&p->pl_sc, (size_t)3 * L * sizeof(double)));
    p->pl_x = x_dev; p->pl_mu = mu; p->pl_mu_reg = mu_reg; p->pl_it = 0;
    if (adjoint_dev(p, y_dev, p->cg_b, false)) return 1;
    if (mu != 1.0) LAUNCH_OK(launch_scale(s, p->cg_b, n, (float)mu));
    if (normal_dev(p, x_dev, p->cg_q, mu)) return 1;
    if (mu_reg != 0.0) LAUNCH_OK(prior_add(p, s, x_dev, p->cg_q, L, (float)mu_reg));
    LAUNCH_OK(launch_residual(s, p->cg_r, p->cg_b, p->cg_q, n));
    HIP_OK(hipMemcpyAsync(p->cg_d, p->cg_r, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    LAUNCH_OK(launch_dot_planes(s, p->cg_r, p->cg_r, L, npix, p->pl_sc));
    return 0;
}
int surfh_cg_planes_step_dev(surfh_plan *p, int32_t iters, int32_t refresh) {
    if (!p || !p->pl_x || !(p->pn_active ? (void *)p->pn_sc : (void *)p->pl_sc)) return fail("surfh_cg_planes_begin_dev has not been called");
    HIP_OK(hipSetDevice(p->dev));
    hipStream_t s = p->stream;
    const int L = p->Lc;
    if (p->pn_active) {
        float *xn = p->pn_v[0], *r = p->pn_v[1], *d = p->pn_v[2], *q = p->pn_v[3], *b = p->pn_v[4];
        double *rr = p->pn_sc, *dq = p->pn_sc + p->LP, *rrn = p->pn_sc + 2 * p->LP;
        const long nc = (long)p->NBP * p->NAP * p->LP;
        for (int i = 0; i < iters; ++i, ++p->pl_it) {
            if (pn_normal(p, d, q, dq)) return 1;
            if (refresh > 0 && p->pl_it % refresh == 0) {
                LAUNCH_OK(launch_pn_step(s, xn, r, d, q, p->Na, p->Nb, p->NAP, p->LP, rr, dq, p->pn_part, rrn, 0));
                if (pn_normal(p, xn, q, dq)) return 1;
                LAUNCH_OK(launch_residual(s, r, b, q, nc));
                LAUNCH_OK(launch_pn_dot(s, r, r, p->Na, p->Nb, p->NAP, p->LP, p->pn_part, rrn));
            } else {
                Prof pr(p, "pn_step");
                LAUNCH_OK(launch_pn_step(s, xn, r, d, q, p->Na, p->Nb, p->NAP, p->LP, rr, dq, p->pn_part, rrn, 1));
            }
            {
                Prof pr(p, "pn_dir");
                LAUNCH_OK(launch_pn_dir(s, d, r, p->Na, p->Nb, p->NAP, p->LP, rrn, rr));
            }
            HIP_OK(hipMemcpyAsync(rr, rrn, (size_t)p->LP * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        LAUNCH_OK(launch_cube_from_lam_inner(s, xn, p->pl_x, 0, L, p->Na, p->Nb, p->NAP, p->LP));      // the caller's iterate
        return 0;
    }
    const long npix = (long)p->Na * p->Nb, n = p->isize;
    double *rr = p->pl_sc, *dq = p->pl_sc + L, *rrn = p->pl_sc + 2 * L;
    float *x = p->pl_x;
    auto Q = [&](const float *v, float *out) -> int {
        if (normal_dev(p, v, out, p->pl_mu)) return 1;
        if (p->pl_mu_reg != 0.0) LAUNCH_OK(prior_add(p, s, v, out, L, (float)p->pl_mu_reg));
        return 0;
    };
    for (int i = 0; i < iters; ++i, ++p->pl_it) {
        if (Q(p->cg_d, p->cg_q)) return 1;
        LAUNCH_OK(launch_dot_planes(s, p->cg_d, p->cg_q, L, npix, dq));
        if (refresh > 0 && p->pl_it % refresh == 0) {
            LAUNCH_OK(launch_cg_step_planes(s, x, p->cg_r, p->cg_d, p->cg_q, L, npix, rr, dq, rrn, 0));
            if (Q(x, p->cg_q)) return 1;
            LAUNCH_OK(launch_residual(s, p->cg_r, p->cg_b, p->cg_q, n));
            LAUNCH_OK(launch_dot_planes(s, p->cg_r, p->cg_r, L, npix, rrn));
        } else {
            LAUNCH_OK(launch_cg_step_planes(s, x, p->cg_r, p->cg_d, p->cg_q, L, npix, rr, dq, rrn, 1));
        }
        LAUNCH_OK(launch_cg_dir_planes(s, p->cg_d, p->cg_r, L, npix, rrn, rr));      // also rr = rr'
    }
    return 0;
}
int surfh_cg_planes_rr(surfh_plan *p, double *rr_host) {
    if (!p || !rr_host || !(p->pn_active ? p->pn_sc : p->pl_sc)) return fail("surfh_cg_planes_begin_dev has not been called");
    HIP_OK(hipSetDevice(p->dev));
    HIP_OK(hipMemcpyAsync(rr_host, p->pn_active ? p->pn_sc : p->pl_sc, (size_t)p->Lc * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));
    return 0;
}

// ---- 3MG on independent planes: what `method = "qmm"` of the 2-D deconvolution driver runs
// (scripts/deconvolution_mrs_noRotation.py:199-212 -> criterion_2D.py:190-193 -> qmm.mmmg); see surfh_mmmg for the scheme.
int surfh_mmmg_planes_cb(surfh_plan *p, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter, double tol,
                         int32_t refresh, float *x, double *grad_norm, int32_t *nit, surfh_cg_callback callback, void *user) {
    if (!p || !y || !x || !grad_norm || !nit) return fail("null argument");
    std::vector<float> hx;
    if (callback) hx.resize((size_t)p->isize);
    if (p->T != 0) return fail("surfh_mmmg_planes is the solver of the plane-wise (no template) model; use surfh_mmmg with templates");
    if (p->ch.empty()) return fail("plan has no channel");
    HIP_OK(hipSetDevice(p->dev));
    if (ensure_cg(p)) return 1;
    if (!p->cg_qm && (dev_alloc(&p->cg_qm, (size_t)p->isize) || dev_alloc(&p->cg_dd, (size_t)p->isize))) return 1;
    hipStream_t s = p->stream;
    const int L = p->Lc;
    const long npix = (long)p->Na * p->Nb, n = p->isize;
    float *r = p->cg_r, *m = p->cg_d, *d = p->cg_dd, *qd = p->cg_q, *qm = p->cg_qm;
    double *sc = nullptr;              // [2][L]: r.r, m.Qm
    HIP_OK(hipMalloc((void **)&sc, (size_t)2 * L * sizeof(double)));
    double *rr = sc, *mqm = sc + L;
    auto done = [&](int rc) { hipFree(sc); return rc; };
    auto Q = [&](const float *v, float *out) -> int {
        if (normal_dev(p, v, out, mu)) return 1;
        if (mu_reg != 0.0) LAUNCH_OK(prior_add(p, s, v, out, L, (float)mu_reg));
        return 0;
    };
    if (hipMemcpyAsync(p->io_y, y, p->osize * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) return done(fail("copy failed"));
    if (adjoint_dev(p, p->io_y, p->cg_b, false)) return done(1);
    int rc = 0;
    if (mu != 1.0) rc = launch_scale(s, p->cg_b, n, (float)mu);
    if (!rc) rc = x0 ? (int)hipMemcpyAsync(p->cg_x, x0, n * sizeof(float), hipMemcpyHostToDevice, s) : launch_fill_zero(s, p->cg_x, n);
    if (!rc) rc = launch_fill_zero(s, m, n);
    if (!rc) rc = launch_fill_zero(s, qm, n);
    if (rc) return done(fail("3MG setup failed"));
    if (Q(p->cg_x, qd)) return done(1);
    rc = launch_residual(s, r, p->cg_b, qd, n);
    if (rc) return done(fail("3MG setup failed"));
    *nit = 0;
    for (int it = 0;; ++it) {
        rc = launch_mmmg_dir_planes(s, d, r, m, qm, L, npix, rr, mqm);
        double *gn = grad_norm + (size_t)it * L;
        if (!rc) rc = (int)hipMemcpyAsync(gn, rr, L * sizeof(double), hipMemcpyDeviceToHost, s);
        if (!rc) rc = (int)hipStreamSynchronize(s);
        if (rc) return done(fail("3MG iteration failed: %s", hipGetErrorString((hipError_t)rc)));
        double worst = 0.0;
        for (int l = 0; l < L; ++l) {
            gn[l] = std::sqrt(gn[l]);
            worst = std::max(worst, gn[l]);
        }
        if (it > 0 && callback) {
            if (hipMemcpyAsync(hx.data(), p->cg_x, n * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                return done(fail("copy failed"));
            if (callback(user, it, grad_norm, hx.data())) break;
            if (hipSetDevice(p->dev) != hipSuccess) return done(fail("hipSetDevice failed"));
        }
        if (it >= max_iter || worst < (double)npix * tol) break;
        if (Q(d, qd)) return done(1);
        const bool fresh = refresh > 0 && it % refresh == 0;
        rc = launch_mmmg_step_planes(s, p->cg_x, r, d, m, qm, qd, L, npix, mqm, fresh ? 0 : 1);
        if (rc) return done(fail("launch failed"));
        if (fresh) {
            if (Q(p->cg_x, qd)) return done(1);
            rc = launch_residual(s, r, p->cg_b, qd, n);
            if (rc) return done(fail("launch failed"));
        }
        *nit = it + 1;
    }
    rc = (int)hipMemcpyAsync(x, p->cg_x, n * sizeof(float), hipMemcpyDeviceToHost, s);
    if (!rc) rc = (int)hipStreamSynchronize(s);
    if (rc) return done(fail("copy failed"));
    return done(0);
}

int surfh_mmmg_planes(surfh_plan *p, const float *y, double mu, double mu_reg, const float *x0, int32_t max_iter, double tol,
                      int32_t refresh, float *x, double *grad_norm, int32_t *nit) {
    return surfh_mmmg_planes_cb(p, y, mu, mu_reg, x0, max_iter, tol, refresh, x, grad_norm, nit, nullptr, nullptr);
}

// ---- drivers' LMM helpers on the device (spectroModel.py:187-198) -----------------------------
static int lmm_host(surfh_plan *p, const double *templates, int32_t T, int32_t L, const float *in, float *out, bool to_cube) {
    if (!p || !templates || !in || !out) return fail("null argument");
    if (T < 1 || L < 1) return fail("bad template shape");
    HIP_OK(hipSetDevice(p->dev));
    const long npix = (long)p->Na * p->Nb;
    std::vector<float> t((size_t)T * L);
    for (size_t i = 0; i < t.size(); ++i) t[i] = (float)templates[i];
    float *dt = nullptr, *dm = nullptr, *dc = nullptr;
    int rc = 0;
    auto done = [&](int r) { hipFree(dt); hipFree(dm); hipFree(dc); return r; };
    if (dev_upload(&dt, t) || dev_alloc(&dm, (size_t)T * npix) || dev_alloc(&dc, (size_t)L * npix)) return done(1);
    hipStream_t s = p->stream;
    if (to_cube) {
        if (hipMemcpyAsync(dm, in, (size_t)T * npix * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) return done(fail("copy failed"));
        rc = launch_lmm_maps2cube(s, dm, dt, dc, T, L, npix);
        if (!rc) rc = (int)hipMemcpyAsync(out, dc, (size_t)L * npix * sizeof(float), hipMemcpyDeviceToHost, s);
    } else {
        if (hipMemcpyAsync(dc, in, (size_t)L * npix * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) return done(fail("copy failed"));
        rc = launch_lmm_cube2maps(s, dc, dt, dm, T, L, npix);
        if (!rc) rc = (int)hipMemcpyAsync(out, dm, (size_t)T * npix * sizeof(float), hipMemcpyDeviceToHost, s);
    }
    if (!rc) rc = (int)hipStreamSynchronize(s);
    if (rc) return done(fail("lmm: %s", hipGetErrorString((hipError_t)rc)));
    return done(0);
}
int surfh_maps_to_cube(surfh_plan *p, const double *templates, int32_t T, int32_t L, const float *maps, float *cube) {
    return lmm_host(p, templates, T, L, maps, cube, true);
}
int surfh_cube_to_maps(surfh_plan *p, const double *templates, int32_t T, int32_t L, const float *cube, float *maps) {
    return lmm_host(p, templates, T, L, cube, maps, false);
}

// ---- instrumentation ------------------------------------------------------------------------
int surfh_profile_enable(surfh_plan *p, int32_t on) {
    if (!p) return fail("null plan");
    p->prof = on != 0;
    return 0;
}
int surfh_profile_filter(surfh_plan *p, const char *prefix) {
    if (!p) return fail("null plan");
    p->prof_filter = prefix ? prefix : "";
    return 0;
}
int32_t surfh_profile_count(surfh_plan *p) {
    if (!p) return -1;
    hipSetDevice(p->dev);
    prof_collect(p);
    return (int32_t)p->acc_names.size();
}
int surfh_profile_get(surfh_plan *p, int32_t i, const char **name, int64_t *launches, double *ms) {
    if (!p || i < 0 || i >= (int32_t)p->acc_names.size()) return fail("bad profile index");
    auto &e = p->acc[p->acc_names[i]];
    *name = p->acc_names[i].c_str();
    *launches = e.first;
    *ms = e.second;
    return 0;
}
int surfh_profile_reset(surfh_plan *p) {
    if (!p) return fail("null plan");
    hipSetDevice(p->dev);
    prof_collect(p);
    p->acc.clear();
    p->acc_names.clear();
    return 0;
}

static int resolve(surfh_plan *p, const char *which, const float **ptr, int64_t dims[4]) {
    std::string w(which ? which : "");
    dims[0] = dims[1] = dims[2] = dims[3] = 1;
    *ptr = nullptr;
    if (w == "blurred" || w == "gcube") {          // [beta][alpha][lambda]; the exact adjoint's accumulator may be its own buffer
        *ptr = (w == "gcube" && p->gcube) ? p->gcube : p->cube; dims[0] = p->NBP; dims[1] = p->NAP; dims[2] = p->LP;
    } else if (w == "spec") {                       // [2][k_alpha][k_beta][lambda]; h2 plans: [k_alpha][k_beta][lambda][2]
        *ptr = p->spec;
        if (p->ilv) { dims[0] = p->KAP; dims[1] = p->KBP; dims[2] = p->LP; dims[3] = 2; }
        else { dims[0] = 2; dims[1] = p->KAP; dims[2] = p->KBP; dims[3] = p->LP; }
    } else if (w == "mhat" && p->T > 0) {
        *ptr = p->mhat; dims[0] = p->T; dims[1] = 2; dims[2] = p->KAP; dims[3] = p->KBP;
    } else if (w.rfind("xs:", 0) == 0 || w.rfind("xsinfo:", 0) == 0) {
        const bool info = w[2] == 'i';
        const int c = atoi(w.c_str() + (info ? 7 : 3));
        if (c < 0 || c >= (int)p->ch.size()) return fail("bad channel index");
        if (info) {                                 // (LinP, first valid lambda column, n_beta_slit, Lin)
            dims[0] = p->ch[c].LinP; dims[1] = p->ch[c].shift; dims[2] = p->ch[c].nbs; dims[3] = p->ch[c].Lin;
        } else {                                    // [(p,s,a)][b'][LinP]
            Channel &ch = p->ch[c];
            if (ch.Xs16) {     // the forward operand lives as block-scaled fp16 pieces: rebuilt in fp32 for inspection
                if (launch_dequant_f16x2(p->stream, ch.Xs16, (long)ch.NP * ch.K, ch.bscale, ch.Xs, ch.NP, ch.K, ch.LinP,
                                         (ch.LinP + 1023) / 1024))
                    return fail("dequant launch failed");
                if (hipStreamSynchronize(p->stream) != hipSuccess) return fail("dequant failed");
            }
            *ptr = p->ch[c].Xs; dims[0] = p->ch[c].NP; dims[1] = p->ch[c].bsum ? 1 : p->ch[c].nbs; dims[2] = p->ch[c].LinP;
        }
    } else if (w == "range") {         // cube columns / rows the channels' tables touch: [a_lo, a_hi) x [b_lo, b_hi)
        dims[0] = p->a_lo; dims[1] = p->a_hi; dims[2] = p->b_lo; dims[3] = p->b_hi;
    } else if (w == "otf") {           // super-tiles (k_beta, 128 wavelengths) inside the OTF's support / all of them
        dims[0] = p->otf_vlist ? p->otf_nvalid : (long)(p->Nb / 2 + 1) * (p->LP / 128); dims[1] = (long)(p->Nb / 2 + 1) * (p->LP / 128);
    } else if (w == "ksteps") {        // (tile, K step) pairs of the spectral-blur GEMMs: near / far of the forward, near / far of the adjoint
        for (auto &c : p->ch)
            for (int i = 0; i < 4; ++i) dims[i] += c.ksteps[i];
        for (int i = 0; i < 4; ++i) dims[i] -= 1;
    } else if (w == "info") {
        dims[0] = p->lo; dims[1] = p->hi; dims[2] = p->Lown; dims[3] = (int64_t)p->segs.size();
    } else {
        return fail("unknown debug buffer '%s'", w.c_str());
    }
    return 0;
}

int surfh_debug_dims(surfh_plan *p, const char *which, int64_t dims[4]) {
    if (!p) return fail("null plan");
    const float *ptr;
    return resolve(p, which, &ptr, dims);
}

int64_t surfh_debug_copy(surfh_plan *p, const char *which, float *out, int64_t cap) {
    if (!p || !out) return -1;
    const float *ptr = nullptr;
    int64_t d[4];
    if (resolve(p, which, &ptr, d) || !ptr) return -1;
    const int64_t n = d[0] * d[1] * d[2] * d[3];
    if (n > cap) {
        fail("capacity %lld < %lld", (long long)cap, (long long)n);
        return -1;
    }
    hipSetDevice(p->dev);
    hipStreamSynchronize(p->stream);
    if (hipMemcpy(out, ptr, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}

int32_t surfh_klist_classify(const float *B, int32_t n, int32_t k, int64_t ldb, int32_t perm_p, int32_t perm_lin, int32_t *records,
                             int64_t capacity) {
    if (!B || !records || n < 1 || k < 32 || k % 32 || ldb < k) return -fail("surfh_klist_classify: bad arguments");
    if (perm_p && ((perm_p != 1 && perm_p != 2 && perm_p != 4 && perm_p != 8) || perm_lin < 1 || perm_lin % (256 / perm_p) || n % perm_lin))
        return -fail("surfh_klist_classify: bad tile shape");
    std::vector<int> kl;
    int stride = 0;
    long nn = 0, nf = 0;
    build_klist(B, n, k, ldb, 0, 0, 1.0 / 256, 1.0 / 1024, &kl, &stride, &nn, &nf, perm_p, perm_lin);
    if ((int64_t)kl.size() > capacity) return -fail("surfh_klist_classify: capacity too small");
    std::memcpy(records, kl.data(), kl.size() * sizeof(int));
    return (int32_t)(kl.size() / (size_t)stride);
}

static long g_selftest_ksteps[2] = {0, 0};
int surfh_gemm_selftest_ksteps(int64_t near_far[2]) {
    near_far[0] = g_selftest_ksteps[0]; near_far[1] = g_selftest_ksteps[1];
    return 0;
}

int surfh_gemm_selftest(int32_t device, int32_t M, int32_t N, int32_t K, int32_t split_k, const float *A,
                        const float *B, float *C) {
    HIP_OK(hipSetDevice(device));
    float *dA = nullptr, *dB = nullptr, *dC = nullptr;
    const int sk = std::max(1, (int)split_k);
    HIP_OK(hipMalloc((void **)&dA, (size_t)M * K * 4));
    HIP_OK(hipMalloc((void **)&dB, (size_t)K * N * 4));
    HIP_OK(hipMalloc((void **)&dC, (size_t)sk * M * N * 4));
    HIP_OK(hipMemcpy(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dB, B, (size_t)K * N * 4, hipMemcpyHostToDevice));
    GemmArgs g;
    g.A0 = dA; g.lda = K; g.B0 = dB; g.ldb = N; g.C = dC; g.ldc = N;
    g.M = M; g.N = N; g.K = K; g.splitK = sk; g.sCsplit = (long)M * N;
    int rc;
    const char *mode = getenv("SURFH_SELFTEST_F16X2");
    if (mode && (mode[0] == '1' || mode[0] == '2')) {
        // two-piece fp16 kernel, NT form: B is handed over as [K][N]; transpose it on the host into [N][K]
        std::vector<float> bt((size_t)N * K);
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < N; ++n) bt[(size_t)n * K + k] = B[(size_t)k * N + n];
        HIP_OK(hipMemcpy(dB, bt.data(), bt.size() * 4, hipMemcpyHostToDevice));
        g.ldb = K;
        unsigned short *dB16 = nullptr, *dA16 = nullptr;
        unsigned *dmax = nullptr;
        float amB = 0.f;
        for (float v : bt) amB = std::max(amB, std::fabs(v));
        std::vector<unsigned> rows((size_t)M, 0u);           // max |A[m][:]| as bit patterns
        for (int m = 0; m < M; ++m) {
            float am = 0.f;
            for (int k = 0; k < K; ++k) am = std::max(am, std::fabs(A[(size_t)m * K + k]));
            memcpy(&rows[m], &am, 4);
        }
        HIP_OK(hipMalloc((void **)&dB16, bt.size() * 4));
        HIP_OK(hipMalloc((void **)&dmax, rows.size() * sizeof(unsigned)));
        HIP_OK(hipMalloc((void **)&dA16, (size_t)M * K * 4));
        HIP_OK(hipMemcpy(dmax, rows.data(), rows.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        g.sB16 = gemm_f16x2_scale(amB); g.B16 = dB16; g.pB16 = (long)bt.size(); g.amax = dmax;
        rc = launch_split2h(nullptr, dB, dB16, (long)bt.size(), (long)bt.size(), g.sB16);
        if (rc == 0) rc = launch_split_rows2h(nullptr, dA, dmax, dA16, M, K, (long)M * K);
        g.A3 = dA16; g.pA3 = (long)M * K;
        int *dkl = nullptr;
        g_selftest_ksteps[0] = g_selftest_ksteps[1] = 0;
        if (mode[0] == '2') {      // with K-step lists, classes and tolerances as at plan creation
            std::vector<int> kl;
            // SURFH_SELFTEST_PERM=<rows per column of B>: tiles of 64 rows of four neighbouring columns, as the adjoint spectral-blur GEMM
            const char *ep = getenv("SURFH_SELFTEST_PERM");
            const int lin = ep ? atoi(ep) : 0;
            if (lin > 0) { g.permP = 4; g.permLin = lin; }
            build_klist(bt.data(), N, K, K, 0, 0, 1.0 / 256, 1.0 / 1024, &kl, &g.klistStride, &g_selftest_ksteps[0], &g_selftest_ksteps[1], g.permP,
                        g.permLin);
            if (dev_upload(&dkl, kl)) return 1;
            g.klist = dkl;
        }
        if (rc == 0) rc = launch_gemm_nt_f16x2_cc(nullptr, g);
        if (rc == 0) rc = (int)hipDeviceSynchronize();
        hipFree(dkl);
        hipFree(dB16);
        hipFree(dmax);
        hipFree(dA16);
    } else {
        rc = launch_gemm_f32(nullptr, g);
    }
    if (rc == 0) rc = (int)hipDeviceSynchronize();
    std::vector<float> h((size_t)sk * M * N);
    if (rc == 0) rc = (int)hipMemcpy(h.data(), dC, h.size() * 4, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dC);
    if (rc) return fail("gemm selftest failed: %s", hipGetErrorString((hipError_t)rc));
    for (size_t i = 0; i < (size_t)M * N; ++i) {
        float s = 0.f;
        for (int k = 0; k < sk; ++k) s += h[(size_t)k * M * N + i];
        C[i] = s;
    }
    return 0;
}

}  // extern "C"
