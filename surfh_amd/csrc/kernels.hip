// HBM-bound kernels of the surfh hot path for gfx950 (see kernels.h for the contracts).
#include "kernels.h"
#include <cstdlib>

namespace {

constexpr int TPB = 256;

// ---------------------------------------------------------------------------------------------
// spectral mix x OTF   (wavelength innermost; spectra are planar [2][PL][LP], or with `ilv` interleaved
// [PL][LP][2] -- the layout of the plans whose transforms run in dft_h2.hip)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cld4(const float *__restrict__ a, long PL, long k, int LP, int l4, int ilv, float4 &re, float4 &im) {
    if (ilv) {
        const float4 v0 = *reinterpret_cast<const float4 *>(a + (k * LP + l4) * 2);
        const float4 v1 = *reinterpret_cast<const float4 *>(a + (k * LP + l4) * 2 + 4);
        re = make_float4(v0.x, v0.z, v1.x, v1.z);
        im = make_float4(v0.y, v0.w, v1.y, v1.w);
    } else {
        re = *reinterpret_cast<const float4 *>(a + k * LP + l4);
        im = *reinterpret_cast<const float4 *>(a + (PL + k) * LP + l4);
    }
}
__device__ __forceinline__ void cst4(float *__restrict__ a, long PL, long k, int LP, int l4, int ilv, const float4 &re, const float4 &im) {
    if (ilv) {
        *reinterpret_cast<float4 *>(a + (k * LP + l4) * 2) = make_float4(re.x, im.x, re.y, im.y);
        *reinterpret_cast<float4 *>(a + (k * LP + l4) * 2 + 4) = make_float4(re.z, im.z, re.w, im.w);
    } else {
        *reinterpret_cast<float4 *>(a + k * LP + l4) = re;
        *reinterpret_cast<float4 *>(a + (PL + k) * LP + l4) = im;
    }
}

__global__ __launch_bounds__(TPB) void specmix_fwd_kernel(const float *__restrict__ mhat,
                                                          const float *__restrict__ sotf,
                                                          const float *__restrict__ tpl, float *__restrict__ spec,
                                                          int T, long PL, int LP, int ilv) {
    const int l4 = (blockIdx.x * TPB + threadIdx.x) * 4;
    if (l4 >= LP) return;
    const long k = blockIdx.y;
    float4 sr = make_float4(0.f, 0.f, 0.f, 0.f), si = sr;
    if (T > 0) {
        for (int t = 0; t < T; ++t) {
            const float4 w = *reinterpret_cast<const float4 *>(tpl + (long)t * LP + l4);
            const float mr = mhat[((long)t * 2 + 0) * PL + k];
            const float mi = mhat[((long)t * 2 + 1) * PL + k];
            sr.x += w.x * mr; sr.y += w.y * mr; sr.z += w.z * mr; sr.w += w.w * mr;
            si.x += w.x * mi; si.y += w.y * mi; si.z += w.z * mi; si.w += w.w * mi;
        }
    } else {
        cld4(mhat, PL, k, LP, l4, ilv, sr, si);
    }
    float4 hr, hi;
    cld4(sotf, PL, k, LP, l4, ilv, hr, hi);
    float4 xr, xi;
    xr.x = hr.x * sr.x - hi.x * si.x; xi.x = hr.x * si.x + hi.x * sr.x;
    xr.y = hr.y * sr.y - hi.y * si.y; xi.y = hr.y * si.y + hi.y * sr.y;
    xr.z = hr.z * sr.z - hi.z * si.z; xi.z = hr.z * si.z + hi.z * sr.z;
    xr.w = hr.w * sr.w - hi.w * si.w; xi.w = hr.w * si.w + hi.w * sr.w;
    cst4(spec, PL, k, LP, l4, ilv, xr, xi);
}

constexpr int MAXT = SURFH_MAX_TEMPLATES;

// one workgroup per frequency bin k: madj[t][c][k] = sum_l tpl[t][l] (conj(H) Y)[c][k][l]
template <typename ACC>
__global__ __launch_bounds__(TPB) void specmix_adj_kernel(const float *__restrict__ spec,
                                                          const float *__restrict__ sotf,
                                                          const float *__restrict__ tpl, float *__restrict__ madj,
                                                          int T, long PL, int LP, int ilv) {
    const long k = blockIdx.x;
    ACC ar[MAXT], ai[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) ar[t] = ai[t] = (ACC)0;
    for (int l4 = threadIdx.x * 4; l4 < LP; l4 += TPB * 4) {
        float4 hr, hi, yr, yi;
        cld4(sotf, PL, k, LP, l4, ilv, hr, hi);
        cld4(spec, PL, k, LP, l4, ilv, yr, yi);
        float4 pr, pi;
        pr.x = hr.x * yr.x + hi.x * yi.x; pi.x = hr.x * yi.x - hi.x * yr.x;
        pr.y = hr.y * yr.y + hi.y * yi.y; pi.y = hr.y * yi.y - hi.y * yr.y;
        pr.z = hr.z * yr.z + hi.z * yi.z; pi.z = hr.z * yi.z - hi.z * yr.z;
        pr.w = hr.w * yr.w + hi.w * yi.w; pi.w = hr.w * yi.w - hi.w * yr.w;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < T) {
                const float4 w = *reinterpret_cast<const float4 *>(tpl + (long)t * LP + l4);
                ar[t] += (ACC)w.x * (ACC)pr.x + (ACC)w.y * (ACC)pr.y + (ACC)w.z * (ACC)pr.z + (ACC)w.w * (ACC)pr.w;
                ai[t] += (ACC)w.x * (ACC)pi.x + (ACC)w.y * (ACC)pi.y + (ACC)w.z * (ACC)pi.z + (ACC)w.w * (ACC)pi.w;
            }
    }
    __shared__ ACC red[TPB / 64][2 * MAXT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        ACC a = ar[t], b = ai[t];
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o, 64);
            b += __shfl_down(b, o, 64);
        }
        if (lane == 0) {
            red[wv][2 * t] = a;
            red[wv][2 * t + 1] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * T) {
        ACC s = (ACC)0;
        for (int w = 0; w < TPB / 64; ++w) s += red[w][threadIdx.x];
        const int t = threadIdx.x >> 1, c = threadIdx.x & 1;
        madj[((long)t * 2 + c) * PL + k] = (float)s;
    }
}

// no-LMM adjoint: out[c][k][l] = (conj(H) Y)[c][k][l]
__global__ __launch_bounds__(TPB) void specmix_adj_plane_kernel(const float *__restrict__ spec,
                                                                const float *__restrict__ sotf,
                                                                float *__restrict__ out, long PL, int LP, int ilv) {
    const int l4 = (blockIdx.x * TPB + threadIdx.x) * 4;
    if (l4 >= LP) return;
    const long k = blockIdx.y;
    float4 hr, hi, yr, yi;
    cld4(sotf, PL, k, LP, l4, ilv, hr, hi);
    cld4(spec, PL, k, LP, l4, ilv, yr, yi);
    float4 pr, pi;
    pr.x = hr.x * yr.x + hi.x * yi.x; pi.x = hr.x * yi.x - hi.x * yr.x;
    pr.y = hr.y * yr.y + hi.y * yi.y; pi.y = hr.y * yi.y - hi.y * yr.y;
    pr.z = hr.z * yr.z + hi.z * yi.z; pi.z = hr.z * yi.z - hi.z * yr.z;
    pr.w = hr.w * yr.w + hi.w * yi.w; pi.w = hr.w * yi.w - hi.w * yr.w;
    cst4(out, PL, k, LP, l4, ilv, pr, pi);
}

// ---- the same three operations on interleaved spectra [PL][LP][2]: a thread handles two complex values = one
// float4 of each array, so every wave instruction reads whole contiguous lines -------------------------------------
__global__ __launch_bounds__(TPB) void specmix_fwd_ilv_kernel(const float *__restrict__ mhat, const float *__restrict__ sotf,
                                                              const float *__restrict__ tpl, float *__restrict__ spec,
                                                              int T, long PL, int LP) {
    const int l2 = (blockIdx.x * TPB + threadIdx.x) * 2;
    if (l2 >= LP) return;
    const long k = blockIdx.y;
    float sr0 = 0.f, si0 = 0.f, sr1 = 0.f, si1 = 0.f;
    if (T > 0) {
        for (int t = 0; t < T; ++t) {
            const float2 w = *reinterpret_cast<const float2 *>(tpl + (long)t * LP + l2);
            const float mr = mhat[((long)t * 2 + 0) * PL + k];
            const float mi = mhat[((long)t * 2 + 1) * PL + k];
            sr0 += w.x * mr; si0 += w.x * mi; sr1 += w.y * mr; si1 += w.y * mi;
        }
    } else {
        const float4 m = *reinterpret_cast<const float4 *>(mhat + (k * LP + l2) * 2);
        sr0 = m.x; si0 = m.y; sr1 = m.z; si1 = m.w;
    }
    const float4 hh = *reinterpret_cast<const float4 *>(sotf + (k * LP + l2) * 2);
    *reinterpret_cast<float4 *>(spec + (k * LP + l2) * 2) =
        make_float4(hh.x * sr0 - hh.y * si0, hh.x * si0 + hh.y * sr0, hh.z * sr1 - hh.w * si1, hh.z * si1 + hh.w * sr1);
}

template <typename ACC>
__global__ __launch_bounds__(TPB) void specmix_adj_ilv_kernel(const float *__restrict__ spec, const float *__restrict__ sotf,
                                                              const float *__restrict__ tpl, float *__restrict__ madj,
                                                              int T, long PL, int LP, SpecmixAdjOpt o) {
    const long k = blockIdx.x;
    ACC ar[MAXT], ai[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) ar[t] = ai[t] = (ACC)0;
    const float *hk = sotf + k * LP * 2, *yk = spec + k * LP * 2;
    // support of the OTF (plan.hip otf_support): chunk c of 128 wavelengths holds nothing at k_beta > lim[c] or at a folded
    // k_alpha > lim[LP / 128 + c]; `spec` was not written there
    const int ka = o.KBP ? (int)(k / o.KBP) : 0, kb = o.KBP ? (int)(k % o.KBP) : 0;
    const int af = ka < o.Na - ka ? ka : o.Na - ka, nch = LP >> 7;
    for (int l0 = threadIdx.x * 2; l0 < LP; l0 += TPB * 4) {      // two float4 of each array in flight per thread
        const int l1 = l0 + TPB * 2;
        // outside the support the loads are redirected to the bin's first wavelengths (cached lines, finite values) and weighted
        // by zero: no branch around a load, no traffic for the skipped chunk
        const bool one = !o.lim || (kb <= o.lim[l0 >> 7] && af <= o.lim[nch + (l0 >> 7)]);
        const bool two = l1 < LP && (!o.lim || (kb <= o.lim[l1 >> 7] && af <= o.lim[nch + (l1 >> 7)]));
        const int e0 = one ? l0 : (int)threadIdx.x % 64 * 2, e1 = two ? l1 : (int)threadIdx.x % 64 * 2;
        const float4 ha = *reinterpret_cast<const float4 *>(hk + (long)e0 * 2);
        const float4 ya = *reinterpret_cast<const float4 *>(yk + (long)e0 * 2);
        const float4 hb = *reinterpret_cast<const float4 *>(hk + (long)e1 * 2);
        const float4 yb = *reinterpret_cast<const float4 *>(yk + (long)e1 * 2);
        const float pr0 = ha.x * ya.x + ha.y * ya.y, pi0 = ha.x * ya.y - ha.y * ya.x;
        const float pr1 = ha.z * ya.z + ha.w * ya.w, pi1 = ha.z * ya.w - ha.w * ya.z;
        const float pr2 = hb.x * yb.x + hb.y * yb.y, pi2 = hb.x * yb.y - hb.y * yb.x;
        const float pr3 = hb.z * yb.z + hb.w * yb.w, pi3 = hb.z * yb.w - hb.w * yb.z;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < T) {
                const float2 wa = one ? *reinterpret_cast<const float2 *>(tpl + (long)t * LP + l0) : make_float2(0.f, 0.f);
                const float2 wb = two ? *reinterpret_cast<const float2 *>(tpl + (long)t * LP + l1) : make_float2(0.f, 0.f);
                ar[t] += (ACC)wa.x * (ACC)pr0 + (ACC)wa.y * (ACC)pr1 + (ACC)wb.x * (ACC)pr2 + (ACC)wb.y * (ACC)pr3;
                ai[t] += (ACC)wa.x * (ACC)pi0 + (ACC)wa.y * (ACC)pi1 + (ACC)wb.x * (ACC)pi2 + (ACC)wb.y * (ACC)pi3;
            }
    }
    __shared__ ACC red[TPB / 64][2 * MAXT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        ACC a = ar[t], b = ai[t];
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o, 64);
            b += __shfl_down(b, o, 64);
        }
        if (lane == 0) {
            red[wv][2 * t] = a;
            red[wv][2 * t + 1] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * T) {
        ACC s = (ACC)0;
        for (int w = 0; w < TPB / 64; ++w) s += red[w][threadIdx.x];
        const int t = threadIdx.x >> 1, c = threadIdx.x & 1;
        const long oo = ((long)t * 2 + c) * PL + k;
        float v = (float)s;
        if (o.Nb) {     // the solver's Parseval-scaled half spectrum, mu and the quadratic prior folded in (dft_h2.h DftH2AdjMix)
            v *= (kb == 0 || 2 * kb == o.Nb) ? o.out_self : o.out_pair;
            if (o.prior_src && ka < o.Na && 2 * kb <= o.Nb)
                v += o.prior_mu * (4.f - 2.f * cospif(2.f * (float)kb / (float)o.Nb) - 2.f * cospif(2.f * (float)ka / (float)o.Na)) * o.prior_src[oo];
        }
        madj[oo] = v;
    }
}

// `out` may hold, on entry, the spectrum of the vector the normal operator was applied to (the forward model left it there): with
// prior_w != 0 the quadratic prior is added in the same pass, out = mu conj(H) Y + prior_w |D(k)|^2 out_old (circular first
// differences are diagonal in the Fourier domain: |D|^2 = 4 - 2 cos(2 pi ka / Na) - 2 cos(2 pi kb / Nb))
__global__ __launch_bounds__(TPB) void specmix_adj_plane_ilv_kernel(const float *__restrict__ spec, const float *__restrict__ sotf,
                                                                    float *out, long PL, int LP, float mu, float prior_w, int Na, int Nb, long KBP) {
    const int l2 = (blockIdx.x * TPB + threadIdx.x) * 2;
    if (l2 >= LP) return;
    const long k = blockIdx.y;
    const float4 hh = *reinterpret_cast<const float4 *>(sotf + (k * LP + l2) * 2);
    const float4 y = *reinterpret_cast<const float4 *>(spec + (k * LP + l2) * 2);
    float4 v = make_float4(hh.x * y.x + hh.y * y.y, hh.x * y.y - hh.y * y.x, hh.z * y.z + hh.w * y.w, hh.z * y.w - hh.w * y.z);
    if (prior_w != 0.f) {
        const int ka = (int)(k / KBP), kb = (int)(k % KBP);
        const float w = prior_w * (4.f - 2.f * cospif(2.f * (float)ka / (float)Na) - 2.f * cospif(2.f * (float)kb / (float)Nb));
        const float4 o = *reinterpret_cast<const float4 *>(out + (k * LP + l2) * 2);
        v = make_float4(mu * v.x + w * o.x, mu * v.y + w * o.y, mu * v.z + w * o.z, mu * v.w + w * o.w);
    }
    *reinterpret_cast<float4 *>(out + (k * LP + l2) * 2) = v;
}

// one workgroup per frequency bin: the T x T Hessian block sum_l tpl tpl' |H|^2
__global__ __launch_bounds__(TPB) void wct_hessian_kernel(const float *__restrict__ sotf, const float *__restrict__ tpl,
                                                          float *__restrict__ hth, int T, long PL, int LP, int ilv) {
    const long k = blockIdx.x;
    float acc[MAXT * (MAXT + 1) / 2];
#pragma unroll
    for (int i = 0; i < MAXT * (MAXT + 1) / 2; ++i) acc[i] = 0.f;
    for (int l = threadIdx.x; l < LP; l += TPB) {
        const float hr = ilv ? sotf[(k * LP + l) * 2] : sotf[k * LP + l], hi = ilv ? sotf[(k * LP + l) * 2 + 1] : sotf[(PL + k) * LP + l];
        const float h2 = hr * hr + hi * hi;
        float w[MAXT];
#pragma unroll
        for (int t = 0; t < MAXT; ++t) w[t] = (t < T) ? tpl[(long)t * LP + l] : 0.f;
        int i = 0;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int u = t; u < MAXT; ++u, ++i) acc[i] += w[t] * w[u] * h2;
    }
    __shared__ float red[TPB / 64][MAXT * (MAXT + 1) / 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < MAXT * (MAXT + 1) / 2; ++i) {
        float a = acc[i];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if (lane == 0) red[wv][i] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int i = 0;
        for (int t = 0; t < MAXT; ++t)
            for (int u = t; u < MAXT; ++u, ++i) {
                if (u >= T) continue;
                float s = 0.f;
                for (int w = 0; w < TPB / 64; ++w) s += red[w][i];
                hth[((long)t * T + u) * PL + k] = s;
                hth[((long)u * T + t) * PL + k] = s;
            }
    }
}

__global__ __launch_bounds__(TPB) void wct_hess_apply_kernel(const float *__restrict__ hth, const float *__restrict__ in,
                                                             float *__restrict__ out, int T, long PL) {
    const long k = (long)blockIdx.x * TPB + threadIdx.x;
    if (k >= PL) return;
    for (int t = 0; t < T; ++t) {
        float sr = 0.f, si = 0.f;
        for (int u = 0; u < T; ++u) {
            const float h = hth[((long)t * T + u) * PL + k];
            sr += h * in[((long)u * 2 + 0) * PL + k];
            si += h * in[((long)u * 2 + 1) * PL + k];
        }
        out[((long)t * 2 + 0) * PL + k] = sr;
        out[((long)t * 2 + 1) * PL + k] = si;
    }
}

// explicit inverse of the regularised normal operator, one thread per frequency bin: solves
// (HtH(f) + diag(mu_t reg(f))) z = b(f) for the real and the imaginary part of b (HtH is real symmetric for
// di = dj = 1).  T <= 8 unknowns, Cholesky in fp64 registers; a vanishing pivot raises `flag`.
__global__ __launch_bounds__(TPB) void wct_solve_kernel(const float *__restrict__ hth, const float *__restrict__ reg,
                                                        const double *__restrict__ mu, const float *__restrict__ in,
                                                        float *__restrict__ out, int T, long PL, int *flag) {
    const long k = (long)blockIdx.x * TPB + threadIdx.x;
    if (k >= PL) return;
    const float rg = reg[k];
    if (rg < 0.f) {                     // padding bin of the spectral layout
        for (int t = 0; t < T; ++t) out[((long)t * 2 + 0) * PL + k] = out[((long)t * 2 + 1) * PL + k] = 0.f;
        return;
    }
    double A[MAXT][MAXT], br[MAXT], bi[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
#pragma unroll
        for (int u = 0; u < MAXT; ++u) A[t][u] = (t < T && u < T) ? (double)hth[((long)t * T + u) * PL + k] : (t == u ? 1.0 : 0.0);
        if (t < T) {
            A[t][t] += mu[t] * (double)rg;
            br[t] = in[((long)t * 2 + 0) * PL + k];
            bi[t] = in[((long)t * 2 + 1) * PL + k];
        } else {
            br[t] = bi[t] = 0.0;
        }
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {    // A = L L^T in place (lower triangle), forward substitution on the fly
        const double ajj = A[j][j];
        double d = ajj;
#pragma unroll
        for (int q = 0; q < MAXT; ++q)
            if (q < j) d -= A[j][q] * A[j][q];
        // HtH reaches this kernel in fp32: a pivot below 1e-6 of its diagonal entry is rounding noise, the matrix is singular
        if (!(d > 1e-6 * ajj)) { bad = true; d = 1.0; }
        const double l = sqrt(d), il = 1.0 / l;
        A[j][j] = l;
#pragma unroll
        for (int i = 0; i < MAXT; ++i)
            if (i > j) {
                double v = A[i][j];
#pragma unroll
                for (int q = 0; q < MAXT; ++q)
                    if (q < j) v -= A[i][q] * A[j][q];
                A[i][j] = v * il;
            }
        double vr = br[j], vi = bi[j];
#pragma unroll
        for (int q = 0; q < MAXT; ++q)
            if (q < j) { vr -= A[j][q] * br[q]; vi -= A[j][q] * bi[q]; }
        br[j] = vr * il;
        bi[j] = vi * il;
    }
#pragma unroll
    for (int j = MAXT - 1; j >= 0; --j) {   // back substitution with L^T
        double vr = br[j], vi = bi[j];
#pragma unroll
        for (int q = 0; q < MAXT; ++q)
            if (q > j) { vr -= A[q][j] * br[q]; vi -= A[q][j] * bi[q]; }
        br[j] = vr / A[j][j];
        bi[j] = vi / A[j][j];
    }
    if (bad) atomicOr(flag, 1);
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
        if (t < T) {
            out[((long)t * 2 + 0) * PL + k] = (float)br[t];
            out[((long)t * 2 + 1) * PL + k] = (float)bi[t];
        }
}

// ---------------------------------------------------------------------------------------------
// sparse row gather vectorised over wavelength: one workgroup = one table row x 1024 wavelengths
// ---------------------------------------------------------------------------------------------
// max |value| of ymat for the per-row operand scales of the two-piece fp16 GEMM: every wave stores its maximum (bit pattern
// of a non-negative float) in its own entry of `pmax`, one small workgroup reduces the entries afterwards (no atomics:
// atomicMax into shared slots serialises in L2 at about 140 ns each).
__device__ __forceinline__ void wave_amax_store(float m, unsigned *pmax, unsigned wave_id) {
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) pmax[wave_id] = __float_as_uint(m);
}

// per-wave maxima -> max |A[m][:]| per GEMM row.  ymat: the entries of a row are contiguous (one thread per row).
__global__ __launch_bounds__(TPB) void rowmax_contiguous_kernel(const unsigned *__restrict__ pmax, int per_row, int nrows, int NP,
                                                                unsigned *__restrict__ rowmax) {
    const int n = blockIdx.x * TPB + threadIdx.x;
    if (n >= NP) return;
    unsigned m = 0;
    if (n < nrows)
        for (int i = 0; i < per_row; ++i) m = max(m, pmax[(long)n * per_row + i]);
    rowmax[n] = m;                                                   // padding rows of the operand are zero
}

__global__ __launch_bounds__(TPB) void spmm_rows_kernel(EllTable t, const float *__restrict__ src,
                                                        float *__restrict__ dst, int nlam, int accumulate) {
    // workgroups are dealt round-robin over the 8 XCDs (each with its own L2): give every XCD one
    // contiguous band of table rows, so neighbouring rows -- which share most of their taps -- hit the same L2
    const int per = (t.R + 7) / 8;
    const int r = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int l4 = (blockIdx.y * TPB + threadIdx.x) * 4;
    if (r >= t.R) return;                                 // workgroup-uniform
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l4 < nlam) {
        const int n = t.cnt[r];
        const int64_t *col = t.col + (long)r * t.W;
        const float *val = t.val + (long)r * t.W;
        int e = 0;
        for (; e + 4 <= n; e += 4) {
            const float4 x0 = *reinterpret_cast<const float4 *>(src + col[e] + l4);
            const float4 x1 = *reinterpret_cast<const float4 *>(src + col[e + 1] + l4);
            const float4 x2 = *reinterpret_cast<const float4 *>(src + col[e + 2] + l4);
            const float4 x3 = *reinterpret_cast<const float4 *>(src + col[e + 3] + l4);
            const float v0 = val[e], v1 = val[e + 1], v2 = val[e + 2], v3 = val[e + 3];
            acc.x += v0 * x0.x; acc.y += v0 * x0.y; acc.z += v0 * x0.z; acc.w += v0 * x0.w;
            acc.x += v1 * x1.x; acc.y += v1 * x1.y; acc.z += v1 * x1.z; acc.w += v1 * x1.w;
            acc.x += v2 * x2.x; acc.y += v2 * x2.y; acc.z += v2 * x2.z; acc.w += v2 * x2.w;
            acc.x += v3 * x3.x; acc.y += v3 * x3.y; acc.z += v3 * x3.z; acc.w += v3 * x3.w;
        }
        for (; e < n; ++e) {
            const float4 x0 = *reinterpret_cast<const float4 *>(src + col[e] + l4);
            const float v0 = val[e];
            acc.x += v0 * x0.x; acc.y += v0 * x0.y; acc.z += v0 * x0.z; acc.w += v0 * x0.w;
        }
        float4 *p = reinterpret_cast<float4 *>(dst + t.dst_off[r] + l4);
        bool rm = accumulate != 0;
        if (rm && t.rng) { const int2 g2 = t.rng[r]; rm = l4 >= g2.x && l4 < g2.y; }
        else if (rm && t.rmw) rm = ((t.rmw[r] >> blockIdx.y) & 1u) != 0;      // workgroup-uniform
        if (rm) {
            const float4 o = *p;
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        *p = acc;
    }
}

// verification twin of spmm_rows_kernel: the same rows with float64 accumulation (surfh_config.verify)
__global__ __launch_bounds__(TPB) void spmm_rows_f64acc_kernel(EllTable t, const float *__restrict__ src, float *__restrict__ dst, int nlam,
                                                               int accumulate) {
    const int r = blockIdx.x;
    const int l = blockIdx.y * TPB + threadIdx.x;
    if (r >= t.R || l >= nlam) return;
    const int n = t.cnt[r];
    const int64_t *col = t.col + (long)r * t.W;
    const float *val = t.val + (long)r * t.W;
    double acc = 0.0;
    for (int e = 0; e < n; ++e) acc += (double)val[e] * (double)src[col[e] + l];
    float *p = dst + t.dst_off[r] + l;
    *p = (float)(accumulate ? (double)*p + acc : acc);
}

// grouped scatter: one workgroup = SCATTER_G neighbouring cube pixels x 1024 wavelengths (see GroupTable)
template <int G>
__global__ __launch_bounds__(TPB) void spmm_group_scatter_kernel(GroupTable t, const float *__restrict__ src, float *__restrict__ dst,
                                                                 int nlam) {
    const int per = (t.NG + 7) / 8;
    const int gi = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int l4 = (blockIdx.y * TPB + threadIdx.x) * 4;
    if (gi >= t.NG || l4 >= nlam) return;
    const int n = t.cnt[gi];
    const int64_t *col = t.col + (long)gi * t.W;
    const float *val = t.val + (long)gi * t.W * G;
    float4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = 0;
    constexpr int SCATTER_UNROLL = 8;      // taps in flight (groups of four: 2 ... 16 alike; groups of eight: 4 -> 8 taps 0.289 -> 0.277 ms per step)
    for (; e + SCATTER_UNROLL <= n; e += SCATTER_UNROLL) {
        float4 xv[SCATTER_UNROLL];
#pragma unroll
        for (int u = 0; u < SCATTER_UNROLL; ++u) xv[u] = *reinterpret_cast<const float4 *>(src + col[e + u] + l4);
#pragma unroll
        for (int u = 0; u < SCATTER_UNROLL; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float v0 = val[(e + u) * G + g];
                acc[g].x += v0 * xv[u].x; acc[g].y += v0 * xv[u].y; acc[g].z += v0 * xv[u].z; acc[g].w += v0 * xv[u].w;
            }
    }
    for (; e < n; ++e) {
        const float4 x0 = *reinterpret_cast<const float4 *>(src + col[e] + l4);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float v0 = val[e * G + g];
            acc[g].x += v0 * x0.x; acc[g].y += v0 * x0.y; acc[g].z += v0 * x0.z; acc[g].w += v0 * x0.w;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t d = t.dst[(long)gi * G + g];
        if (d < 0) continue;                               // workgroup-uniform
        float4 *p = reinterpret_cast<float4 *>(dst + d + l4);
        float4 a = acc[g];
        bool rm;
        if (t.rng) { const int2 g2 = t.rng[(long)gi * G + g]; rm = l4 >= g2.x && l4 < g2.y; }
        else rm = ((t.rmw[(long)gi * G + g] >> blockIdx.y) & 1u) != 0;
        if (rm) {
            const float4 o = *p;
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        *p = a;
    }
}

// The gather with its output written as the two fp16 pieces the all-consumer GEMM reads (gemm_cc16.hip).  The scale is
// per workgroup, i.e. per (operand row, segment of <= 1024 wavelengths of one beta column): the workgroup's maximum is known
// before anything is stored, so no second pass over the operand is needed, and the scale is finer than one per row.
// bscale[seg][NP] receives the power of two; seg = (column offset / LinP) * nchunk + chunk.
__device__ __forceinline__ float f16x2_block_scale(float amax) {       // as f16x2_scale_of in gemm_cc16.hip
    if (!(amax > 0.f)) return 1.f;
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127;
    int sc = e - 13;
    sc = sc < -126 ? -126 : (sc > 127 ? 127 : sc);
    return __uint_as_float((unsigned)(sc + 127) << 23);
}

__global__ __launch_bounds__(TPB) void spmm_rows_f16_kernel(EllTable t, const float *__restrict__ src, unsigned short *__restrict__ dst16,
                                                            long plane, int nlam, float *__restrict__ bscale, int NP, long K, int LinP,
                                                            int nchunk) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int per = (t.R + 7) / 8;
    const int r = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int l4 = (blockIdx.y * TPB + threadIdx.x) * 4;
    if (r >= t.R) return;                                 // workgroup-uniform
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l4 < nlam) {
        const int n = t.cnt[r];
        const int64_t *col = t.col + (long)r * t.W;
        const float *val = t.val + (long)r * t.W;
        int e = 0;
        for (; e + 4 <= n; e += 4) {
            const float4 x0 = *reinterpret_cast<const float4 *>(src + col[e] + l4);
            const float4 x1 = *reinterpret_cast<const float4 *>(src + col[e + 1] + l4);
            const float4 x2 = *reinterpret_cast<const float4 *>(src + col[e + 2] + l4);
            const float4 x3 = *reinterpret_cast<const float4 *>(src + col[e + 3] + l4);
            const float v0 = val[e], v1 = val[e + 1], v2 = val[e + 2], v3 = val[e + 3];
            acc.x += v0 * x0.x; acc.y += v0 * x0.y; acc.z += v0 * x0.z; acc.w += v0 * x0.w;
            acc.x += v1 * x1.x; acc.y += v1 * x1.y; acc.z += v1 * x1.z; acc.w += v1 * x1.w;
            acc.x += v2 * x2.x; acc.y += v2 * x2.y; acc.z += v2 * x2.z; acc.w += v2 * x2.w;
            acc.x += v3 * x3.x; acc.y += v3 * x3.y; acc.z += v3 * x3.z; acc.w += v3 * x3.w;
        }
        for (; e < n; ++e) {
            const float4 x0 = *reinterpret_cast<const float4 *>(src + col[e] + l4);
            const float v0 = val[e];
            acc.x += v0 * x0.x; acc.y += v0 * x0.y; acc.z += v0 * x0.z; acc.w += v0 * x0.w;
        }
    }
    // workgroup maximum -> power-of-two scale
    float m = fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w)));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float sm[TPB / 64];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    const float scale = f16x2_block_scale(m), inv = 1.f / scale;
    const int64_t off = t.dst_off[r];
    if (threadIdx.x == 0) bscale[((off % K) / LinP * nchunk + blockIdx.y) * (long)NP + off / K] = scale;
    if (l4 < nlam) {
        const float x0 = acc.x * inv, x1 = acc.y * inv, x2 = acc.z * inv, x3 = acc.w * inv;
        const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1, h2 = (_Float16)x2, h3 = (_Float16)x3;
        f16x4 h = {h0, h1, h2, h3};
        f16x4 l = {(_Float16)(x0 - (float)h0), (_Float16)(x1 - (float)h1), (_Float16)(x2 - (float)h2), (_Float16)(x3 - (float)h3)};
        *reinterpret_cast<f16x4 *>(dst16 + off + l4) = h;
        *reinterpret_cast<f16x4 *>(dst16 + plane + off + l4) = l;
    }
}

// the same on a grouped table: the members' taps are read once; every member keeps its own block scale
template <int G>
__global__ __launch_bounds__(TPB) void spmm_group_gather_f16_kernel(GroupTable t, const float *__restrict__ src,
                                                                    unsigned short *__restrict__ dst16, long plane, int nlam,
                                                                    float *__restrict__ bscale, int NP, long K, int LinP, int nchunk) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int per = (t.NG + 7) / 8;
    const int gi = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int l4 = (blockIdx.y * TPB + threadIdx.x) * 4;
    if (gi >= t.NG) return;                               // workgroup-uniform
    float4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l4 < nlam) {
        const int n = t.cnt[gi];
        const int64_t *col = t.col + (long)gi * t.W;
        const float *val = t.val + (long)gi * t.W * G;
        int e = 0;
        // eight taps requested before the first is used: with two the kernel waited on memory latency (0.45 -> 0.34 ms per
        // step on config 3; 4: 0.36, 12 / 16 / 24: 0.36 / 0.38 / 0.36; fewer resident workgroups only cost time)
        constexpr int GATHER_UNROLL = 8;
        for (; e + GATHER_UNROLL <= n; e += GATHER_UNROLL) {
            float4 xv[GATHER_UNROLL];
#pragma unroll
            for (int u = 0; u < GATHER_UNROLL; ++u) xv[u] = *reinterpret_cast<const float4 *>(src + col[e + u] + l4);
#pragma unroll
            for (int u = 0; u < GATHER_UNROLL; ++u)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float v0 = val[(e + u) * G + g];
                    acc[g].x += v0 * xv[u].x; acc[g].y += v0 * xv[u].y; acc[g].z += v0 * xv[u].z; acc[g].w += v0 * xv[u].w;
                }
        }
        for (; e < n; ++e) {
            const float4 x0 = *reinterpret_cast<const float4 *>(src + col[e] + l4);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float v0 = val[e * G + g];
                acc[g].x += v0 * x0.x; acc[g].y += v0 * x0.y; acc[g].z += v0 * x0.z; acc[g].w += v0 * x0.w;
            }
        }
    }
    __shared__ float sm[G][TPB / 64];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float m = fmaxf(fmaxf(fabsf(acc[g].x), fabsf(acc[g].y)), fmaxf(fabsf(acc[g].z), fabsf(acc[g].w)));
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) sm[g][threadIdx.x >> 6] = m;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t off = t.dst[(long)gi * G + g];
        if (off < 0) continue;                             // workgroup-uniform
        const float scale = f16x2_block_scale(fmaxf(fmaxf(sm[g][0], sm[g][1]), fmaxf(sm[g][2], sm[g][3]))), inv = 1.f / scale;
        if (threadIdx.x == 0) bscale[((off % K) / LinP * nchunk + blockIdx.y) * (long)NP + off / K] = scale;
        if (l4 < nlam) {
            const float x0 = acc[g].x * inv, x1 = acc[g].y * inv, x2 = acc[g].z * inv, x3 = acc[g].w * inv;
            const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1, h2 = (_Float16)x2, h3 = (_Float16)x3;
            f16x4 h = {h0, h1, h2, h3};
            f16x4 l = {(_Float16)(x0 - (float)h0), (_Float16)(x1 - (float)h1), (_Float16)(x2 - (float)h2), (_Float16)(x3 - (float)h3)};
            *reinterpret_cast<f16x4 *>(dst16 + off + l4) = h;
            *reinterpret_cast<f16x4 *>(dst16 + plane + off + l4) = l;
        }
    }
}

// debugging: the block-scaled fp16 operand back as fp32 [NP][K]
__global__ __launch_bounds__(TPB) void dequant_f16x2_kernel(const unsigned short *__restrict__ src16, long plane, const float *__restrict__ bscale,
                                                            float *__restrict__ dst, int NP, long K, int LinP, int nchunk) {
    const long i = (long)blockIdx.x * TPB + threadIdx.x;
    if (i >= (long)NP * K) return;
    const long row = i / K, k = i % K;
    const float sc = bscale[((k / LinP) * nchunk + (k % LinP) / 1024) * (long)NP + row];
    const _Float16 h = reinterpret_cast<const _Float16 *>(src16)[i], l = reinterpret_cast<const _Float16 *>(src16)[plane + i];
    dst[i] = ((float)h + (float)l) * sc;
}

// [L][na][nb] planes l0.. of a wavelength-major cube -> [nb][nap][LP] wavelength innermost (32x32 LDS tile transpose)
__global__ __launch_bounds__(256) void cube_to_lam_inner_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                                int l0, int L, int na, int nb, int nap, int LP) {
    __shared__ float tile[32][33];
    const int a = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int l = blockIdx.y * 32 + i, b = blockIdx.x * 32 + tx;
        tile[i][tx] = (l < L && b < nb) ? src[((long)(l0 + l) * na + a) * nb + b] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int b = blockIdx.x * 32 + i, l = blockIdx.y * 32 + tx;
        if (b < nb && l < L) dst[((long)b * nap + a) * LP + l] = tile[tx][i];
    }
}

__global__ __launch_bounds__(256) void cube_from_lam_inner_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                                  int l0, int L, int na, int nb, int nap, int LP) {
    __shared__ float tile[32][33];
    const int a = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int b = blockIdx.x * 32 + i, l = blockIdx.y * 32 + tx;
        tile[i][tx] = (b < nb && l < L) ? src[((long)b * nap + a) * LP + l] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int l = blockIdx.y * 32 + i, b = blockIdx.x * 32 + tx;
        if (l < L && b < nb) dst[((long)(l0 + l) * na + a) * nb + b] = tile[tx][i];
    }
}

// ---------------------------------------------------------------------------------------------
// layout helpers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void pad_planes_kernel(const float *__restrict__ src, float *__restrict__ dst, int na,
                                                         int nb, int nap, int nbp) {
    const int j = blockIdx.x * TPB + threadIdx.x;
    const int i = blockIdx.y;
    const long b = blockIdx.z;
    if (j < nb) dst[(b * nap + i) * nbp + j] = src[(b * na + i) * nb + j];
}

__global__ __launch_bounds__(TPB) void unpad_planes_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                           int na, int nb, int nap, int nbp) {
    const int j = blockIdx.x * TPB + threadIdx.x;
    const int i = blockIdx.y;
    const long b = blockIdx.z;
    if (j < nb) dst[(b * na + i) * nb + j] = src[(b * nap + i) * nbp + j];
}

// thread index runs over l fastest so the slab reads are coalesced
__global__ __launch_bounds__(TPB) void y_from_cpart_kernel(const float *__restrict__ cpart, long slab, int nsplit,
                                                           float *__restrict__ y, int PS, int Ldet, int aout,
                                                           int LdetP) {
    const int l = (blockIdx.x * TPB + threadIdx.x) * 4;          // four detector wavelengths per thread: 16-byte slab reads
    const int n = blockIdx.y;           // ps*aout + a
    if (l >= Ldet) return;
    const int ps = n / aout, a = n % aout;
    const long src = (long)n * LdetP + l;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < nsplit; ++k) {
        const float4 v = *reinterpret_cast<const float4 *>(cpart + k * slab + src);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float *o = y + ((long)ps * Ldet + l) * aout + a;
    o[0] = s.x;
    if (l + 1 < Ldet) o[aout] = s.y;
    if (l + 2 < Ldet) o[2 * aout] = s.z;
    if (l + 3 < Ldet) o[3 * aout] = s.w;
}

__global__ __launch_bounds__(TPB) void ymat_from_y_kernel(const float *__restrict__ y, float *__restrict__ ymat, int PS,
                                                          int Ldet, int aout, int LdetP, unsigned *amax) {
    const int l = blockIdx.x * TPB + threadIdx.x;
    const int n = blockIdx.y;
    float v = 0.f;
    if (l < Ldet) {
        const int ps = n / aout, a = n % aout;
        v = y[((long)ps * Ldet + l) * aout + a];
        ymat[(long)n * LdetP + l] = v;
    }
    if (amax) wave_amax_store(fabsf(v), amax, (blockIdx.y * gridDim.x + blockIdx.x) * (TPB / 64) + (threadIdx.x >> 6));
}

// Normal operator A^T A d: the forward model's output y is only the hand-over to the adjoint, whose GEMM operand
// ymat[n][l] = y[ps][l][a] (n = ps * aout + a) is again the slab sum in the slabs' own layout.  One workgroup per operand row:
// sum of the K slabs (same order as y_from_cpart_kernel), the row's maximum, the two fp16 pieces of the row at its
// power-of-two scale -- the bits y_from_cpart -> ymat_from_y -> rowmax -> split_rows2h produce, in one pass over the slabs.
__device__ __forceinline__ float f16x2_row_scale(float amax) {         // f16x2_scale_of of gemm_cc16.hip
    if (!(amax > 0.f)) return 1.f;
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127;
    int sc = e - 13;
    sc = sc < -126 ? -126 : (sc > 127 ? 127 : sc);
    return __uint_as_float((unsigned)(sc + 127) << 23);
}
template <int NV>      // float4 per thread: LdetP <= NV * 1024
__global__ __launch_bounds__(TPB) void ymat16_from_cpart_kernel(const float *__restrict__ cpart, long slab, int nsplit, unsigned short *__restrict__ dst16,
                                                                long plane, unsigned *__restrict__ rowmax, int nrows, int Ldet, int LdetP) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int n = blockIdx.x;
    float4 v[NV];
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int l = (j * TPB + threadIdx.x) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < nrows && l < Ldet) {
            const long src = (long)n * LdetP + l;
            for (int k = 0; k < nsplit; ++k) {
                const float4 x = *reinterpret_cast<const float4 *>(cpart + k * slab + src);
                s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
            }
            if (l + 1 >= Ldet) s.y = 0.f;        // columns beyond Ldet hold the products of the zero rows of W: not part of y
            if (l + 2 >= Ldet) s.z = 0.f;
            if (l + 3 >= Ldet) s.w = 0.f;
        }
        v[j] = s;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(s.x), fabsf(s.y)), fmaxf(fabsf(s.z), fabsf(s.w))));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float sm[TPB / 64];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    if (threadIdx.x == 0) rowmax[n] = __float_as_uint(m);
    const float inv = 1.f / f16x2_row_scale(m);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int l = (j * TPB + threadIdx.x) * 4;
        if (l >= LdetP) continue;
        const float x0 = v[j].x * inv, x1 = v[j].y * inv, x2 = v[j].z * inv, x3 = v[j].w * inv;
        const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1, h2 = (_Float16)x2, h3 = (_Float16)x3;
        f16x4 h = {h0, h1, h2, h3};
        f16x4 lo = {(_Float16)(x0 - (float)h0), (_Float16)(x1 - (float)h1), (_Float16)(x2 - (float)h2), (_Float16)(x3 - (float)h3)};
        const long o = (long)n * LdetP + l;
        *reinterpret_cast<f16x4 *>(dst16 + o) = h;
        *reinterpret_cast<f16x4 *>(dst16 + plane + o) = lo;
    }
}

// dst[plane][ka][kb] = src * (f_self if kb = 0 or 2 kb = Nb, else f_pair): spectra <-> the solver's Parseval-scaled form
__global__ __launch_bounds__(TPB) void spec_scale_kernel(const float *__restrict__ src, float *__restrict__ dst, long PL, int KBP, int Nb, float f_self,
                                                         float f_pair) {
    const long i = (long)blockIdx.x * TPB + threadIdx.x;
    if (i >= PL) return;
    const int kb = (int)(i % KBP);
    const long o = (long)blockIdx.y * PL + i;
    dst[o] = src[o] * ((kb == 0 || 2 * kb == Nb) ? f_self : f_pair);
}
// q += mu_reg (Dr^T Dr + Dc^T Dc) d on half spectra: circular first differences are diagonal in the Fourier domain
__global__ __launch_bounds__(TPB) void spec_prior_add_kernel(const float *__restrict__ d, float *__restrict__ q, int Na, int Nb, long PL, int KBP,
                                                             float mu_reg) {
    const long i = (long)blockIdx.x * TPB + threadIdx.x;
    if (i >= PL) return;
    const int kb = (int)(i % KBP), ka = (int)(i / KBP);
    if (ka >= Na || kb > Nb / 2) return;
    const long o = (long)blockIdx.y * PL + i;
    q[o] += mu_reg * (4.f - 2.f * cospif(2.f * (float)ka / (float)Na) - 2.f * cospif(2.f * (float)kb / (float)Nb)) * d[o];
}

__global__ __launch_bounds__(TPB) void fill_zero_kernel(float *p, long n) {
    long i = (long)blockIdx.x * TPB + threadIdx.x;
    const long stride = (long)gridDim.x * TPB;
    for (; i < n; i += stride) p[i] = 0.f;      // 1.05 GB cube in 0.16 ms = 6.4 TB/s: at the roofline as it is
}

// ---------------------------------------------------------------------------------------------
// CG vector kernels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void prior_add_kernel(const float *__restrict__ d, float *__restrict__ q, int na,
                                                        int nb, float mu) {
    const int j = blockIdx.x * TPB + threadIdx.x;
    const int i = blockIdx.y;
    const long t = blockIdx.z;
    if (j >= nb) return;
    const float *p = d + t * na * nb;
    const int im = (i == 0) ? na - 1 : i - 1, ip = (i == na - 1) ? 0 : i + 1;
    const int jm = (j == 0) ? nb - 1 : j - 1, jp = (j == nb - 1) ? 0 : j + 1;
    const float c = p[(long)i * nb + j];
    const float lap = (2.f * c - p[(long)im * nb + j] - p[(long)ip * nb + j]) +
                      (2.f * c - p[(long)i * nb + jm] - p[(long)i * nb + jp]);
    q[t * na * nb + (long)i * nb + j] += mu * lap;
}

// joint prior: q += mu * L^T L d with L the circular convolution by udft's 3 x 3 Laplacian [[0,-1,0],[-1,4,-1],[0,-1,0]]
// centred on the pixel (Difference_Operator_Joint.DtD, fusion_CT.py:45-62: |ir2fr(laplacian(2))|^2 in the Fourier domain).
// L is symmetric, L^T L is the 13-tap stencil 20 / -8 (axis neighbours) / 2 (diagonals) / 1 (axis distance 2).
__global__ __launch_bounds__(TPB) void prior_joint_add_kernel(const float *__restrict__ d, float *__restrict__ q, int na, int nb, float mu) {
    const int j = blockIdx.x * TPB + threadIdx.x;
    const int i = blockIdx.y;
    const long t = blockIdx.z;
    if (j >= nb) return;
    const float *p = d + t * na * nb;
    auto wi = [&](int v) { return ((v % na) + na) % na; };
    auto wj = [&](int v) { return ((v % nb) + nb) % nb; };
    auto at = [&](int a, int b) { return p[(long)wi(a) * nb + wj(b)]; };
    const float v = 20.f * at(i, j) - 8.f * (at(i - 1, j) + at(i + 1, j) + at(i, j - 1) + at(i, j + 1)) +
                    2.f * (at(i - 1, j - 1) + at(i - 1, j + 1) + at(i + 1, j - 1) + at(i + 1, j + 1)) +
                    (at(i - 2, j) + at(i + 2, j) + at(i, j - 2) + at(i, j + 2));
    q[t * na * nb + (long)i * nb + j] += mu * v;
}

__global__ __launch_bounds__(TPB) void scale_kernel(float *x, long n, float a) {
    long i = (long)blockIdx.x * TPB + threadIdx.x;
    const long stride = (long)gridDim.x * TPB;
    for (; i < n; i += stride) x[i] *= a;
}

__device__ inline double block_sum(double v) {
    __shared__ double sm[TPB / 64];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int k = 0; k < TPB / 64; ++k) s += sm[k];
    return s;   // valid in thread 0
}

__global__ __launch_bounds__(TPB) void dot_partial_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          long n, double *__restrict__ scratch) {
    double s = 0.0;
    const long stride = (long)gridDim.x * TPB;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) s += (double)a[i] * (double)b[i];
    s = block_sum(s);
    if (threadIdx.x == 0) scratch[blockIdx.x] = s;
}

__global__ __launch_bounds__(TPB) void reduce_final_kernel(const double *__restrict__ scratch, int nparts,
                                                           double *__restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += TPB) s += scratch[i];
    s = block_sum(s);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(TPB) void cg_step_kernel(float *__restrict__ x, float *__restrict__ r,
                                                      const float *__restrict__ d, const float *__restrict__ q,
                                                      long n, const double *__restrict__ rr,
                                                      const double *__restrict__ dq, double *__restrict__ scratch) {
    const float step = (float)(rr[0] / dq[0]);
    double s = 0.0;
    const long stride = (long)gridDim.x * TPB;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        x[i] += step * d[i];
        const float rn = r[i] - step * q[i];
        r[i] = rn;
        s += (double)rn * (double)rn;
    }
    s = block_sum(s);
    if (threadIdx.x == 0) scratch[blockIdx.x] = s;
}

__global__ __launch_bounds__(TPB) void cg_xupdate_kernel(float *__restrict__ x, const float *__restrict__ d, long n,
                                                         const double *__restrict__ rr,
                                                         const double *__restrict__ dq) {
    const float step = (float)(rr[0] / dq[0]);
    const long stride = (long)gridDim.x * TPB;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) x[i] += step * d[i];
}

__global__ __launch_bounds__(TPB) void cg_dir_kernel(float *__restrict__ d, const float *__restrict__ r, long n,
                                                     const double *__restrict__ rr_new,
                                                     const double *__restrict__ rr_old) {
    const float beta = (float)(rr_new[0] / rr_old[0]);
    const long stride = (long)gridDim.x * TPB;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) d[i] = r[i] + beta * d[i];
}

// ---- CG iteration with device-resident scalars in three launches (dot partials, step, direction) and no copies: a launch sums
// the previous launch's per-block partial sums itself, every block for itself and in reduce_final_kernel's order (same bits),
// instead of waiting for a one-block reduction launch in between.
__device__ inline double parts_sum(const double *__restrict__ scratch, int nparts) {      // valid in every thread
    __shared__ double smp[TPB / 64];
    __shared__ double tot;
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += TPB) s += scratch[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) smp[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < TPB / 64; ++k) t += smp[k];
        tot = t;
    }
    __syncthreads();
    return tot;
}

// x += (rr / d.q) d;  r -= (rr / d.q) q;  partial sums of r.r;  d.q = the sum of `dq_parts`
__global__ __launch_bounds__(TPB) void cg_step_parts_kernel(float *__restrict__ x, float *__restrict__ r, const float *__restrict__ d,
                                                            const float *__restrict__ q, long n, const double *__restrict__ rr,
                                                            const double *__restrict__ dq_parts, int nparts, double *__restrict__ dq_out,
                                                            double *__restrict__ scratch) {
    const double dq = parts_sum(dq_parts, nparts);
    if (blockIdx.x == 0 && threadIdx.x == 0) dq_out[0] = dq;
    const float step = (float)(rr[0] / dq);
    double s = 0.0;
    const long stride = (long)gridDim.x * TPB;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        x[i] += step * d[i];
        const float rn = r[i] - step * q[i];
        r[i] = rn;
        s += (double)rn * (double)rn;
    }
    s = block_sum(s);
    if (threadIdx.x == 0) scratch[blockIdx.x] = s;
}

// rr' = the sum of `rr_parts` -> rr_out (a slot nobody reads in this launch);  d = r + (rr' / rr_old) d
__global__ __launch_bounds__(TPB) void cg_dir_parts_kernel(float *__restrict__ d, const float *__restrict__ r, long n,
                                                           const double *__restrict__ rr_parts, int nparts,
                                                           const double *__restrict__ rr_old, double *__restrict__ rr_out) {
    const double rrn = parts_sum(rr_parts, nparts);
    if (blockIdx.x == 0 && threadIdx.x == 0) rr_out[0] = rrn;
    const float beta = (float)(rrn / rr_old[0]);
    const long stride = (long)gridDim.x * TPB;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) d[i] = r[i] + beta * d[i];
}

__global__ __launch_bounds__(TPB) void residual_kernel(float *__restrict__ r, const float *__restrict__ b,
                                                       const float *__restrict__ q, long n) {
    const long stride = (long)gridDim.x * TPB;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) r[i] = b[i] - q[i];
}

// ---- memory-gradient (3MG) vector kernels.  The subspace span{r, m} is handled in the basis [d, m] with d = r + beta m
// Q-orthogonal to the previous move m; the operator images are carried by linearity.
__global__ __launch_bounds__(TPB) void lincomb_kernel(float *__restrict__ out, const float *__restrict__ a,
                                                      const float *__restrict__ b, long n, float beta) {
    const long stride = (long)gridDim.x * TPB;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) out[i] = a[i] + beta * b[i];
}

__global__ __launch_bounds__(TPB) void mmmg_update_kernel(float *__restrict__ x, float *__restrict__ r, const float *__restrict__ d,
                                                          float *__restrict__ m, float *__restrict__ qm,
                                                          const float *__restrict__ qd, long n, float s0, float s1, int update_r) {
    const long stride = (long)gridDim.x * TPB;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        const float mv = s0 * d[i] + s1 * m[i];
        const float qv = s0 * qd[i] + s1 * qm[i];
        x[i] += mv;
        m[i] = mv;
        qm[i] = qv;
        if (update_r) r[i] -= qv;
    }
}

// ---- CG on a batch of independent planes (the 2-D deconvolution path, one problem per wavelength): every plane has
// its own step and direction scalars.  One workgroup per plane; fp64 sums; a plane whose residual is already zero
// (no data, or converged exactly) keeps still.
__global__ __launch_bounds__(TPB) void dot_planes_kernel(const float *__restrict__ a, const float *__restrict__ b, long npix,
                                                         double *__restrict__ out) {
    const long off = (long)blockIdx.x * npix;
    double s = 0.0;
    for (long i = threadIdx.x; i < npix; i += TPB) s += (double)a[off + i] * (double)b[off + i];
    s = block_sum(s);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// x += s d ; (update_r) r -= s q, rr' = r.r      with s = rr / dq of this plane
__global__ __launch_bounds__(TPB) void cg_step_planes_kernel(float *__restrict__ x, float *__restrict__ r, const float *__restrict__ d,
                                                             const float *__restrict__ q, long npix, const double *__restrict__ rr,
                                                             const double *__restrict__ dq, double *__restrict__ rrn, int update_r) {
    const long off = (long)blockIdx.x * npix;
    const double den = dq[blockIdx.x];
    const float step = den != 0.0 ? (float)(rr[blockIdx.x] / den) : 0.f;
    double s = 0.0;
    for (long i = threadIdx.x; i < npix; i += TPB) {
        x[off + i] += step * d[off + i];
        if (update_r) {
            const float rn = r[off + i] - step * q[off + i];
            r[off + i] = rn;
            s += (double)rn * (double)rn;
        }
    }
    if (update_r) {
        s = block_sum(s);
        if (threadIdx.x == 0) rrn[blockIdx.x] = s;
    }
}

// d = r + (rr' / rr) d per plane, then rr <- rr'
__global__ __launch_bounds__(TPB) void cg_dir_planes_kernel(float *__restrict__ d, const float *__restrict__ r, long npix,
                                                            const double *__restrict__ rrn, double *__restrict__ rr) {
    const long off = (long)blockIdx.x * npix;
    const double old = rr[blockIdx.x];
    const float beta = old != 0.0 ? (float)(rrn[blockIdx.x] / old) : 0.f;
    for (long i = threadIdx.x; i < npix; i += TPB) d[off + i] = r[off + i] + beta * d[off + i];
    __syncthreads();
    if (threadIdx.x == 0) rr[blockIdx.x] = rrn[blockIdx.x];
}

// ---- the same CG blocks on WAVELENGTH-INNERMOST arrays [NB rows beta][NAP][LP] (the plan's cube layout): the plane-wise solver keeps its
// vectors in the layout the transforms read and write, so an iteration holds no layout transpose.  A thread owns one wavelength
// and walks a slice of the pixels; the per-wavelength sums are formed in two deterministic stages (PN_SPLIT partial sums each).
constexpr int PN_SPLIT = 128;
__device__ __forceinline__ void pn_partial(double s, double *__restrict__ part, long LP, int l) {
    __shared__ double sm[TPB / 64][64];
    sm[threadIdx.x >> 6][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        double t = 0.0;
        for (int w = 0; w < TPB / 64; ++w) t += sm[w][threadIdx.x];
        part[(long)blockIdx.y * LP + l] = t;
    }
}
// q += mu_reg (Dr^T Dr + Dc^T Dc) d (circular first differences, fusion_CT.py:16-43 / criterion_2D.py:16-43), part = d . q per wavelength
__global__ __launch_bounds__(TPB) void pn_prior_dot_kernel(const float *__restrict__ d, float *__restrict__ q, int Na, int Nb, int NAP, long LP,
                                                           float mu_reg, double *__restrict__ part) {
    const int l = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const long npix = (long)Na * Nb, p0 = npix * blockIdx.y / gridDim.y, p1 = npix * (blockIdx.y + 1) / gridDim.y;
    double s = 0.0;
    for (long pix = p0 + sub; pix < p1; pix += TPB / 64) {
        const int b = (int)(pix / Na), a = (int)(pix % Na);
        const long idx = ((long)b * NAP + a) * LP + l;
        const float dv = d[idx];
        float qv = q[idx];
        if (mu_reg != 0.f) {
            const int am = a ? a - 1 : Na - 1, ap = a + 1 < Na ? a + 1 : 0, bm = b ? b - 1 : Nb - 1, bp = b + 1 < Nb ? b + 1 : 0;
            const float lap = (2.f * dv - d[((long)b * NAP + am) * LP + l] - d[((long)b * NAP + ap) * LP + l]) +
                              (2.f * dv - d[((long)bm * NAP + a) * LP + l] - d[((long)bp * NAP + a) * LP + l]);
            qv += mu_reg * lap;
            q[idx] = qv;
        }
        s += (double)dv * (double)qv;
    }
    pn_partial(s, part, LP, l);
}
__global__ __launch_bounds__(TPB) void pn_dot_kernel(const float *__restrict__ x, const float *__restrict__ y, int Na, int Nb, int NAP, long LP,
                                                     double *__restrict__ part) {
    const int l = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const long npix = (long)Na * Nb, p0 = npix * blockIdx.y / gridDim.y, p1 = npix * (blockIdx.y + 1) / gridDim.y;
    double s = 0.0;
    for (long pix = p0 + sub; pix < p1; pix += TPB / 64) {
        const long idx = ((long)(pix / Na) * NAP + pix % Na) * LP + l;
        s += (double)x[idx] * (double)y[idx];
    }
    pn_partial(s, part, LP, l);
}
__global__ __launch_bounds__(TPB) void pn_reduce_kernel(const double *__restrict__ part, int S, long LP, double *__restrict__ out) {
    const long l = (long)blockIdx.x * TPB + threadIdx.x;
    if (l >= LP) return;
    double t = 0.0;
    for (int k = 0; k < S; ++k) t += part[(long)k * LP + l];
    out[l] = t;
}
// x += s d ; (update_r) r -= s q, part = r . r      with s = rr / dq of the wavelength
__global__ __launch_bounds__(TPB) void pn_step_kernel(float *__restrict__ x, float *__restrict__ r, const float *__restrict__ d, const float *__restrict__ q,
                                                      int Na, int Nb, int NAP, long LP, const double *__restrict__ rr, const double *__restrict__ dq,
                                                      double *__restrict__ part, int update_r) {
    const int l = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const long npix = (long)Na * Nb, p0 = npix * blockIdx.y / gridDim.y, p1 = npix * (blockIdx.y + 1) / gridDim.y;
    const double den = dq[l];
    const float step = den != 0.0 ? (float)(rr[l] / den) : 0.f;
    double s = 0.0;
    for (long pix = p0 + sub; pix < p1; pix += TPB / 64) {
        const long idx = ((long)(pix / Na) * NAP + pix % Na) * LP + l;
        x[idx] += step * d[idx];
        if (update_r) {
            const float rn = r[idx] - step * q[idx];
            r[idx] = rn;
            s += (double)rn * (double)rn;
        }
    }
    if (update_r) pn_partial(s, part, LP, l);
}
// d = r + (rr' / rr) d per wavelength
__global__ __launch_bounds__(TPB) void pn_dir_kernel(float *__restrict__ d, const float *__restrict__ r, int Na, int Nb, int NAP, long LP,
                                                     const double *__restrict__ rrn, const double *__restrict__ rr) {
    const int l = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const long npix = (long)Na * Nb, p0 = npix * blockIdx.y / gridDim.y, p1 = npix * (blockIdx.y + 1) / gridDim.y;
    const double old = rr[l];
    const float beta = old != 0.0 ? (float)(rrn[l] / old) : 0.f;
    for (long pix = p0 + sub; pix < p1; pix += TPB / 64) {
        const long idx = ((long)(pix / Na) * NAP + pix % Na) * LP + l;
        d[idx] = r[idx] + beta * d[idx];
    }
}

// ---- 3MG on a batch of independent planes (the 2-D deconvolution drivers select it: deconvolution_mrs_noRotation.py:199-212).
// Same two-step scheme as surfh_mmmg, every plane with its own scalars, one workgroup per plane.
template <int N>
__device__ inline void block_sum_bcast(double (&v)[N]) {   // sums over the workgroup, result in every thread
    __shared__ double sm[N][TPB / 64];
    __shared__ double tot[N];
    __syncthreads();                                       // previous use of the buffers is over
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double t = v[k];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
        if ((threadIdx.x & 63) == 0) sm[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        double t = 0.0;
        for (int w = 0; w < TPB / 64; ++w) t += sm[threadIdx.x][w];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = tot[k];
}

// rr = r.r (trace), mQm kept for the step; d = r + beta m with beta = -(r.Qm)/(m.Qm)
__global__ __launch_bounds__(TPB) void mmmg_dir_planes_kernel(float *__restrict__ d, const float *__restrict__ r,
                                                              const float *__restrict__ m, const float *__restrict__ qm, long npix,
                                                              double *__restrict__ rr, double *__restrict__ mqm) {
    const long off = (long)blockIdx.x * npix;
    double v[3] = {0.0, 0.0, 0.0};
    for (long i = threadIdx.x; i < npix; i += TPB) {
        const double ri = r[off + i], qi = qm[off + i];
        v[0] += ri * ri;
        v[1] += ri * qi;
        v[2] += (double)m[off + i] * qi;
    }
    block_sum_bcast<3>(v);
    const float beta = v[2] > 0.0 ? (float)(-v[1] / v[2]) : 0.f;
    for (long i = threadIdx.x; i < npix; i += TPB) d[off + i] = r[off + i] + beta * m[off + i];
    if (threadIdx.x == 0) {
        rr[blockIdx.x] = v[0];
        mqm[blockIdx.x] = v[2];
    }
}

// 2x2 subspace step of every plane in the basis [d, m]; a plane without curvature (no data, or converged) keeps still
__global__ __launch_bounds__(TPB) void mmmg_step_planes_kernel(float *__restrict__ x, float *__restrict__ r, const float *__restrict__ d,
                                                               float *__restrict__ m, float *__restrict__ qm,
                                                               const float *__restrict__ qd, long npix,
                                                               const double *__restrict__ mqm, int update_r) {
    const long off = (long)blockIdx.x * npix;
    double v[4] = {0.0, 0.0, 0.0, 0.0};                    // d.Qd, d.Qm, d.r, m.r
    for (long i = threadIdx.x; i < npix; i += TPB) {
        const double di = d[off + i], ri = r[off + i];
        v[0] += di * (double)qd[off + i];
        v[1] += di * (double)qm[off + i];
        v[2] += di * ri;
        v[3] += (double)m[off + i] * ri;
    }
    block_sum_bcast<4>(v);
    const double dQd = v[0], dQm = v[1], dr = v[2], mr = v[3], mQm = mqm[blockIdx.x];
    double s0 = dQd > 0.0 ? dr / dQd : 0.0, s1 = 0.0;
    if (dQd > 0.0 && mQm > 0.0) {
        const double sc = sqrt(dQd * mQm), c = dQm / sc, det = 1.0 - c * c;
        if (det > 1e-12) {
            s0 = (dr / dQd - c * mr / sc) / det;
            s1 = (mr / mQm - c * dr / sc) / det;
        }
    }
    const float f0 = (float)s0, f1 = (float)s1;
    for (long i = threadIdx.x; i < npix; i += TPB) {
        const float mv = f0 * d[off + i] + f1 * m[off + i];
        const float qv = f0 * qd[off + i] + f1 * qm[off + i];
        x[off + i] += mv;
        m[off + i] = mv;
        qm[off + i] = qv;
        if (update_r) r[off + i] -= qv;
    }
}

inline int nblocks(long n, int cap = 2048) {
    long b = (n + TPB - 1) / TPB;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}
constexpr int DOT_BLOCKS = 512;

}  // namespace

int launch_specmix_fwd(hipStream_t s, const float *mhat, const float *sotf, const float *tpl, float *spec, int T,
                       long PL, int LP, int ilv) {
    if (T > MAXT) return (int)hipErrorInvalidValue;
    if (ilv) {
        dim3 grid((LP / 2 + TPB - 1) / TPB, (unsigned)PL);
        hipLaunchKernelGGL(specmix_fwd_ilv_kernel, grid, dim3(TPB), 0, s, mhat, sotf, tpl, spec, T, PL, LP);
        return (int)hipGetLastError();
    }
    dim3 grid((LP / 4 + TPB - 1) / TPB, (unsigned)PL);
    hipLaunchKernelGGL(specmix_fwd_kernel, grid, dim3(TPB), 0, s, mhat, sotf, tpl, spec, T, PL, LP, 0);
    return (int)hipGetLastError();
}

int launch_specmix_adj(hipStream_t s, const float *spec, const float *sotf, const float *tpl, float *madj, int T,
                       long PL, int LP, bool f64, int ilv, const SpecmixAdjOpt *opt) {
    if (T > MAXT) return (int)hipErrorInvalidValue;
    const SpecmixAdjOpt o = opt ? *opt : SpecmixAdjOpt();
    if (opt && T == 0 && (!ilv || o.lim || (o.prior_src && (o.prior_src != madj || o.KBP < 1 || o.Na < 1 || o.Nb < 1)))) return (int)hipErrorInvalidValue;
    if (opt && T != 0 && (!ilv || (o.lim && LP % 128) || ((o.lim || o.Nb) && (o.KBP < 1 || o.Na < 1)))) return (int)hipErrorInvalidValue;
    if (ilv) {
        if (T == 0) {
            dim3 grid((LP / 2 + TPB - 1) / TPB, (unsigned)PL);
            hipLaunchKernelGGL(specmix_adj_plane_ilv_kernel, grid, dim3(TPB), 0, s, spec, sotf, madj, PL, LP, o.out_self, o.prior_src ? o.prior_mu : 0.f,
                               o.Na, o.Nb, o.KBP);
        } else if (f64) hipLaunchKernelGGL(specmix_adj_ilv_kernel<double>, dim3((unsigned)PL), dim3(TPB), 0, s, spec, sotf, tpl, madj, T, PL, LP, o);
        else hipLaunchKernelGGL(specmix_adj_ilv_kernel<float>, dim3((unsigned)PL), dim3(TPB), 0, s, spec, sotf, tpl, madj, T, PL, LP, o);
        return (int)hipGetLastError();
    }
    if (T == 0) {
        dim3 grid((LP / 4 + TPB - 1) / TPB, (unsigned)PL);
        hipLaunchKernelGGL(specmix_adj_plane_kernel, grid, dim3(TPB), 0, s, spec, sotf, madj, PL, LP, 0);
    } else {
        if (f64) hipLaunchKernelGGL(specmix_adj_kernel<double>, dim3((unsigned)PL), dim3(TPB), 0, s, spec, sotf, tpl, madj, T, PL, LP, 0);
        else hipLaunchKernelGGL(specmix_adj_kernel<float>, dim3((unsigned)PL), dim3(TPB), 0, s, spec, sotf, tpl, madj, T, PL, LP, 0);
    }
    return (int)hipGetLastError();
}

int launch_wct_hessian(hipStream_t s, const float *sotf, const float *tpl, float *hth, int T, long PL, int LP, int ilv) {
    if (T < 1 || T > MAXT) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(wct_hessian_kernel, dim3((unsigned)PL), dim3(TPB), 0, s, sotf, tpl, hth, T, PL, LP, ilv);
    return (int)hipGetLastError();
}

int launch_wct_hess_apply(hipStream_t s, const float *hth, const float *in, float *out, int T, long PL) {
    hipLaunchKernelGGL(wct_hess_apply_kernel, dim3((unsigned)((PL + TPB - 1) / TPB)), dim3(TPB), 0, s, hth, in, out, T, PL);
    return (int)hipGetLastError();
}

namespace {

// ---------------------------------------------------------------------------------------------
// linear mixing model on plane-major arrays (driver utilities, not on the iteration path)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void lmm_maps2cube_kernel(const float *__restrict__ maps, const float *__restrict__ tpl,
                                                            float *__restrict__ cube, int T, int L, long npix) {
    const long i = (long)blockIdx.x * TPB + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= npix) return;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += tpl[(long)t * L + l] * maps[(long)t * npix + i];
    cube[(long)l * npix + i] = acc;
}

// one thread per pixel walks the planes; T <= 8 running sums in registers, every plane read is coalesced
__global__ __launch_bounds__(TPB) void lmm_cube2maps_kernel(const float *__restrict__ cube, const float *__restrict__ tpl,
                                                            float *__restrict__ maps, int T, int L, long npix) {
    const long i = (long)blockIdx.x * TPB + threadIdx.x;
    if (i >= npix) return;
    for (int t0 = 0; t0 < T; t0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < L; ++l) {
            const float v = cube[(long)l * npix + i];
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t0 + t < T) acc[t] += tpl[(long)(t0 + t) * L + l] * v;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t0 + t < T) maps[(long)(t0 + t) * npix + i] = acc[t];
    }
}

}  // namespace

long ymat_from_y_waves(int PS, int Ldet, int aout) { return (long)((Ldet + TPB - 1) / TPB) * PS * aout * (TPB / 64); }

int launch_spmm_rows(hipStream_t s, const EllTable &t, const float *src, float *dst, int nlam, int accumulate) {
    if (t.R == 0 || nlam <= 0) return 0;
    if (nlam % 4) return (int)hipErrorInvalidValue;
    dim3 grid((t.R + 7) / 8 * 8, (nlam / 4 + TPB - 1) / TPB);
    hipLaunchKernelGGL(spmm_rows_kernel, grid, dim3(TPB), 0, s, t, src, dst, nlam, accumulate);
    return (int)hipGetLastError();
}

int launch_spmm_rows_f64acc(hipStream_t s, const EllTable &t, const float *src, float *dst, int nlam, int accumulate) {
    if (t.R == 0 || nlam <= 0) return 0;
    dim3 grid((unsigned)t.R, (nlam + TPB - 1) / TPB);
    hipLaunchKernelGGL(spmm_rows_f64acc_kernel, grid, dim3(TPB), 0, s, t, src, dst, nlam, accumulate);
    return (int)hipGetLastError();
}

int launch_spmm_group_scatter(hipStream_t s, const GroupTable &t, const float *src, float *dst, int nlam) {
    if (t.NG == 0 || nlam <= 0) return 0;
    if (nlam % 4 || (nlam / 4 + TPB - 1) / TPB > 32) return (int)hipErrorInvalidValue;
    dim3 grid((t.NG + 7) / 8 * 8, (nlam / 4 + TPB - 1) / TPB);
    if (t.G == 4) hipLaunchKernelGGL(spmm_group_scatter_kernel<4>, grid, dim3(TPB), 0, s, t, src, dst, nlam);
    else if (t.G == 8) hipLaunchKernelGGL(spmm_group_scatter_kernel<8>, grid, dim3(TPB), 0, s, t, src, dst, nlam);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

int launch_spmm_rows_f16(hipStream_t s, const EllTable &t, const float *src, unsigned short *dst16, long plane, int nlam, float *bscale,
                         int NP, long K, int LinP) {
    if (t.R == 0 || nlam <= 0) return 0;
    if (nlam % 4 || plane % 4 || K % LinP || LinP % 32 || !bscale) return (int)hipErrorInvalidValue;
    dim3 grid((t.R + 7) / 8 * 8, (nlam / 4 + TPB - 1) / TPB);
    hipLaunchKernelGGL(spmm_rows_f16_kernel, grid, dim3(TPB), 0, s, t, src, dst16, plane, nlam, bscale, NP, K, LinP, (LinP + 1023) / 1024);
    return (int)hipGetLastError();
}

int launch_spmm_group_gather_f16(hipStream_t s, const GroupTable &t, const float *src, unsigned short *dst16, long plane, int nlam,
                                 float *bscale, int NP, long K, int LinP) {
    if (t.NG == 0 || nlam <= 0) return 0;
    if (nlam % 4 || plane % 4 || K % LinP || LinP % 32 || !bscale) return (int)hipErrorInvalidValue;
    dim3 grid((t.NG + 7) / 8 * 8, (nlam / 4 + TPB - 1) / TPB);
    if (t.G == 4)
        hipLaunchKernelGGL(spmm_group_gather_f16_kernel<4>, grid, dim3(TPB), 0, s, t, src, dst16, plane, nlam, bscale, NP, K, LinP, (LinP + 1023) / 1024);
    else if (t.G == 8)
        hipLaunchKernelGGL(spmm_group_gather_f16_kernel<8>, grid, dim3(TPB), 0, s, t, src, dst16, plane, nlam, bscale, NP, K, LinP, (LinP + 1023) / 1024);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

int launch_dequant_f16x2(hipStream_t s, const unsigned short *src16, long plane, const float *bscale, float *dst, int NP, long K, int LinP,
                         int nchunk) {
    const long n = (long)NP * K;
    hipLaunchKernelGGL(dequant_f16x2_kernel, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, s, src16, plane, bscale, dst, NP, K, LinP,
                       nchunk);
    return (int)hipGetLastError();
}

int launch_cube_to_lam_inner(hipStream_t s, const float *src, float *dst, int l0, int L, int na, int nb, int nap, int LP) {
    dim3 grid((nb + 31) / 32, (L + 31) / 32, na);
    hipLaunchKernelGGL(cube_to_lam_inner_kernel, grid, dim3(256), 0, s, src, dst, l0, L, na, nb, nap, LP);
    return (int)hipGetLastError();
}

int launch_cube_from_lam_inner(hipStream_t s, const float *src, float *dst, int l0, int L, int na, int nb, int nap, int LP) {
    dim3 grid((nb + 31) / 32, (L + 31) / 32, na);
    hipLaunchKernelGGL(cube_from_lam_inner_kernel, grid, dim3(256), 0, s, src, dst, l0, L, na, nb, nap, LP);
    return (int)hipGetLastError();
}

int launch_pad_planes(hipStream_t s, const float *src, float *dst, int B, int na, int nb, int nap, int nbp) {
    dim3 grid((nb + TPB - 1) / TPB, na, B);
    hipLaunchKernelGGL(pad_planes_kernel, grid, dim3(TPB), 0, s, src, dst, na, nb, nap, nbp);
    return (int)hipGetLastError();
}

int launch_unpad_planes(hipStream_t s, const float *src, float *dst, int B, int na, int nb, int nap, int nbp) {
    dim3 grid((nb + TPB - 1) / TPB, na, B);
    hipLaunchKernelGGL(unpad_planes_kernel, grid, dim3(TPB), 0, s, src, dst, na, nb, nap, nbp);
    return (int)hipGetLastError();
}

int launch_y_from_cpart(hipStream_t s, const float *cpart, long slab, int nsplit, float *y, int PS, int Ldet,
                        int aout, int LdetP) {
    if (LdetP % 4 || slab % 4) return (int)hipErrorInvalidValue;
    dim3 grid(((Ldet + 3) / 4 + TPB - 1) / TPB, PS * aout);
    hipLaunchKernelGGL(y_from_cpart_kernel, grid, dim3(TPB), 0, s, cpart, slab, nsplit, y, PS, Ldet, aout, LdetP);
    return (int)hipGetLastError();
}

int launch_ymat_from_y(hipStream_t s, const float *y, float *ymat, int PS, int Ldet, int aout, int LdetP, unsigned *pmax,
                       unsigned *rowmax, int NP) {
    if (pmax && (!rowmax || NP < PS * aout)) return (int)hipErrorInvalidValue;
    dim3 grid((Ldet + TPB - 1) / TPB, PS * aout);
    hipLaunchKernelGGL(ymat_from_y_kernel, grid, dim3(TPB), 0, s, y, ymat, PS, Ldet, aout, LdetP, pmax);
    if (pmax)
        hipLaunchKernelGGL(rowmax_contiguous_kernel, dim3((NP + TPB - 1) / TPB), dim3(TPB), 0, s, pmax, (int)grid.x * (TPB / 64), PS * aout,
                           NP, rowmax);
    return (int)hipGetLastError();
}

int launch_ymat16_from_cpart(hipStream_t s, const float *cpart, long slab, int nsplit, unsigned short *dst16, long plane, unsigned *rowmax,
                             int NP, int nrows, int Ldet, int LdetP) {
    if (LdetP % 4 || slab % 4 || plane % 4 || LdetP > 4 * 4 * TPB || nrows > NP) return (int)hipErrorInvalidValue;
    const int nv = (LdetP + 4 * TPB - 1) / (4 * TPB);
#define SURFH_Y16(NV_) hipLaunchKernelGGL(ymat16_from_cpart_kernel<NV_>, dim3(NP), dim3(TPB), 0, s, cpart, slab, nsplit, dst16, plane, rowmax, nrows, Ldet, LdetP)
    if (nv <= 1) SURFH_Y16(1);
    else if (nv == 2) SURFH_Y16(2);
    else if (nv == 3) SURFH_Y16(3);
    else SURFH_Y16(4);
#undef SURFH_Y16
    return (int)hipGetLastError();
}

int launch_spec_scale(hipStream_t s, const float *src, float *dst, int planes, long PL, long KBP, int Nb, float f_self, float f_pair) {
    hipLaunchKernelGGL(spec_scale_kernel, dim3((unsigned)((PL + TPB - 1) / TPB), planes), dim3(TPB), 0, s, src, dst, PL, (int)KBP, Nb, f_self, f_pair);
    return (int)hipGetLastError();
}
int launch_spec_prior_add(hipStream_t s, const float *d, float *q, int planes, int Na, int Nb, long PL, long KBP, float mu_reg) {
    hipLaunchKernelGGL(spec_prior_add_kernel, dim3((unsigned)((PL + TPB - 1) / TPB), planes), dim3(TPB), 0, s, d, q, Na, Nb, PL, (int)KBP, mu_reg);
    return (int)hipGetLastError();
}

int launch_fill_zero(hipStream_t s, float *p, long n) {
    hipLaunchKernelGGL(fill_zero_kernel, dim3(nblocks(n, 4096)), dim3(TPB), 0, s, p, n);
    return (int)hipGetLastError();
}

int launch_prior_add(hipStream_t s, const float *d, float *q, int T, int na, int nb, float mu_reg) {
    dim3 grid((nb + TPB - 1) / TPB, na, T);
    hipLaunchKernelGGL(prior_add_kernel, grid, dim3(TPB), 0, s, d, q, na, nb, mu_reg);
    return (int)hipGetLastError();
}

int launch_prior_joint_add(hipStream_t s, const float *d, float *q, int T, int na, int nb, float mu_reg) {
    if (na < 3 || nb < 3) return (int)hipErrorInvalidValue;
    dim3 grid((nb + TPB - 1) / TPB, na, T);
    hipLaunchKernelGGL(prior_joint_add_kernel, grid, dim3(TPB), 0, s, d, q, na, nb, mu_reg);
    return (int)hipGetLastError();
}

int launch_scale(hipStream_t s, float *x, long n, float a) {
    hipLaunchKernelGGL(scale_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, x, n, a);
    return (int)hipGetLastError();
}

int launch_dot(hipStream_t s, const float *a, const float *b, long n, double *scratch, double *out) {
    const int nb = nblocks(n, DOT_BLOCKS);
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(TPB), 0, s, a, b, n, scratch);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(TPB), 0, s, scratch, nb, out);
    return (int)hipGetLastError();
}

int launch_cg_step(hipStream_t s, float *x, float *r, const float *d, const float *q, long n, const double *rr,
                   const double *dq, double *scratch, double *out_rr) {
    const int nb = nblocks(n, DOT_BLOCKS);
    hipLaunchKernelGGL(cg_step_kernel, dim3(nb), dim3(TPB), 0, s, x, r, d, q, n, rr, dq, scratch);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(TPB), 0, s, scratch, nb, out_rr);
    return (int)hipGetLastError();
}

// scratch: 2 * DOT_BLOCKS doubles.  dot(a, b) partials only (no reduction launch); the consumers below sum them
int launch_dot_parts(hipStream_t s, const float *a, const float *b, long n, double *parts) {
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nblocks(n, DOT_BLOCKS)), dim3(TPB), 0, s, a, b, n, parts);
    return (int)hipGetLastError();
}
int launch_cg_step_parts(hipStream_t s, float *x, float *r, const float *d, const float *q, long n, const double *rr, const double *dq_parts,
                         double *dq_out, double *rr_parts) {
    const int nb = nblocks(n, DOT_BLOCKS);
    hipLaunchKernelGGL(cg_step_parts_kernel, dim3(nb), dim3(TPB), 0, s, x, r, d, q, n, rr, dq_parts, nb, dq_out, rr_parts);
    return (int)hipGetLastError();
}
int launch_cg_dir_parts(hipStream_t s, float *d, const float *r, long n, const double *rr_parts, const double *rr_old, double *rr_out) {
    hipLaunchKernelGGL(cg_dir_parts_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, d, r, n, rr_parts, nblocks(n, DOT_BLOCKS), rr_old, rr_out);
    return (int)hipGetLastError();
}
int dot_parts_stride() { return DOT_BLOCKS; }

int launch_cg_xupdate(hipStream_t s, float *x, const float *d, long n, const double *rr, const double *dq) {
    hipLaunchKernelGGL(cg_xupdate_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, x, d, n, rr, dq);
    return (int)hipGetLastError();
}

int launch_cg_dir(hipStream_t s, float *d, const float *r, long n, const double *rr_new, const double *rr_old) {
    hipLaunchKernelGGL(cg_dir_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, d, r, n, rr_new, rr_old);
    return (int)hipGetLastError();
}

int launch_lincomb(hipStream_t s, float *out, const float *a, const float *b, long n, double beta) {
    hipLaunchKernelGGL(lincomb_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, out, a, b, n, (float)beta);
    return (int)hipGetLastError();
}

int launch_mmmg_update(hipStream_t s, float *x, float *r, const float *d, float *m, float *qm, const float *qd, long n, double s0,
                       double s1, int update_r) {
    hipLaunchKernelGGL(mmmg_update_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, x, r, d, m, qm, qd, n, (float)s0, (float)s1, update_r);
    return (int)hipGetLastError();
}

int launch_residual(hipStream_t s, float *r, const float *b, const float *q, long n) {
    hipLaunchKernelGGL(residual_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, r, b, q, n);
    return (int)hipGetLastError();
}

int launch_lmm_maps2cube(hipStream_t s, const float *maps, const float *tpl, float *cube, int T, int L, long npix) {
    dim3 grid((unsigned)((npix + TPB - 1) / TPB), (unsigned)L);
    hipLaunchKernelGGL(lmm_maps2cube_kernel, grid, dim3(TPB), 0, s, maps, tpl, cube, T, L, npix);
    return (int)hipGetLastError();
}

int launch_lmm_cube2maps(hipStream_t s, const float *cube, const float *tpl, float *maps, int T, int L, long npix) {
    hipLaunchKernelGGL(lmm_cube2maps_kernel, dim3((unsigned)((npix + TPB - 1) / TPB)), dim3(TPB), 0, s, cube, tpl, maps, T, L, npix);
    return (int)hipGetLastError();
}

int launch_wct_solve(hipStream_t s, const float *hth, const float *reg, const double *mu, const float *in, float *out, int T,
                     long PL, int *flag) {
    hipLaunchKernelGGL(wct_solve_kernel, dim3((unsigned)((PL + TPB - 1) / TPB)), dim3(TPB), 0, s, hth, reg, mu, in, out, T, PL, flag);
    return (int)hipGetLastError();
}

int launch_dot_planes(hipStream_t s, const float *a, const float *b, int nplanes, long npix, double *out) {
    hipLaunchKernelGGL(dot_planes_kernel, dim3(nplanes), dim3(TPB), 0, s, a, b, npix, out);
    return (int)hipGetLastError();
}

int launch_cg_step_planes(hipStream_t s, float *x, float *r, const float *d, const float *q, int nplanes, long npix, const double *rr,
                          const double *dq, double *rrn, int update_r) {
    hipLaunchKernelGGL(cg_step_planes_kernel, dim3(nplanes), dim3(TPB), 0, s, x, r, d, q, npix, rr, dq, rrn, update_r);
    return (int)hipGetLastError();
}

int launch_mmmg_dir_planes(hipStream_t s, float *d, const float *r, const float *m, const float *qm, int nplanes, long npix,
                           double *rr, double *mqm) {
    hipLaunchKernelGGL(mmmg_dir_planes_kernel, dim3(nplanes), dim3(TPB), 0, s, d, r, m, qm, npix, rr, mqm);
    return (int)hipGetLastError();
}

int launch_mmmg_step_planes(hipStream_t s, float *x, float *r, const float *d, float *m, float *qm, const float *qd, int nplanes,
                            long npix, const double *mqm, int update_r) {
    hipLaunchKernelGGL(mmmg_step_planes_kernel, dim3(nplanes), dim3(TPB), 0, s, x, r, d, m, qm, qd, npix, mqm, update_r);
    return (int)hipGetLastError();
}

// the wavelength-innermost forms: part = work buffer of PN_SPLIT * LP doubles, out / rr / dq / rrn = [LP] doubles
int launch_pn_prior_dot(hipStream_t s, const float *d, float *q, int Na, int Nb, int NAP, long LP, float mu_reg, double *part, double *out) {
    if (LP % 64) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pn_prior_dot_kernel, dim3((unsigned)(LP / 64), PN_SPLIT), dim3(TPB), 0, s, d, q, Na, Nb, NAP, LP, mu_reg, part);
    hipLaunchKernelGGL(pn_reduce_kernel, dim3((unsigned)((LP + TPB - 1) / TPB)), dim3(TPB), 0, s, part, PN_SPLIT, LP, out);
    return (int)hipGetLastError();
}
int launch_pn_dot(hipStream_t s, const float *x, const float *y, int Na, int Nb, int NAP, long LP, double *part, double *out) {
    if (LP % 64) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pn_dot_kernel, dim3((unsigned)(LP / 64), PN_SPLIT), dim3(TPB), 0, s, x, y, Na, Nb, NAP, LP, part);
    hipLaunchKernelGGL(pn_reduce_kernel, dim3((unsigned)((LP + TPB - 1) / TPB)), dim3(TPB), 0, s, part, PN_SPLIT, LP, out);
    return (int)hipGetLastError();
}
int launch_pn_step(hipStream_t s, float *x, float *r, const float *d, const float *q, int Na, int Nb, int NAP, long LP, const double *rr,
                   const double *dq, double *part, double *rrn, int update_r) {
    if (LP % 64) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pn_step_kernel, dim3((unsigned)(LP / 64), PN_SPLIT), dim3(TPB), 0, s, x, r, d, q, Na, Nb, NAP, LP, rr, dq, part, update_r);
    if (update_r) hipLaunchKernelGGL(pn_reduce_kernel, dim3((unsigned)((LP + TPB - 1) / TPB)), dim3(TPB), 0, s, part, PN_SPLIT, LP, rrn);
    return (int)hipGetLastError();
}
int launch_pn_dir(hipStream_t s, float *d, const float *r, int Na, int Nb, int NAP, long LP, const double *rrn, const double *rr) {
    if (LP % 64) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pn_dir_kernel, dim3((unsigned)(LP / 64), PN_SPLIT), dim3(TPB), 0, s, d, r, Na, Nb, NAP, LP, rrn, rr);
    return (int)hipGetLastError();
}
size_t pn_part_doubles(long LP) { return (size_t)PN_SPLIT * (size_t)LP; }

int launch_cg_dir_planes(hipStream_t s, float *d, const float *r, int nplanes, long npix, const double *rrn, double *rr) {
    hipLaunchKernelGGL(cg_dir_planes_kernel, dim3(nplanes), dim3(TPB), 0, s, d, r, npix, rrn, rr);
    return (int)hipGetLastError();
}
