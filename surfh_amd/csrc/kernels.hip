// HBM-bound kernels of the surfh hot path for gfx950 (see kernels.h for the contracts).
#include "kernels.h"

namespace {

constexpr int TPB = 256;

// ---------------------------------------------------------------------------------------------
// spectral mix x OTF
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void specmix_fwd_kernel(const float *__restrict__ mhat,
                                                          const float *__restrict__ sotf,
                                                          const float *__restrict__ tpl, float *__restrict__ spec,
                                                          int T, int L, long PL) {
    const long k4 = (long)blockIdx.x * TPB + threadIdx.x;
    if (k4 * 4 >= PL) return;
    const int l = blockIdx.y;
    float4 sr = make_float4(0.f, 0.f, 0.f, 0.f), si = sr;
    if (T > 0) {
        for (int t = 0; t < T; ++t) {
            const float w = tpl[(long)t * L + l];
            const float4 mr = *reinterpret_cast<const float4 *>(mhat + ((long)t * 2 + 0) * PL + k4 * 4);
            const float4 mi = *reinterpret_cast<const float4 *>(mhat + ((long)t * 2 + 1) * PL + k4 * 4);
            sr.x += w * mr.x; sr.y += w * mr.y; sr.z += w * mr.z; sr.w += w * mr.w;
            si.x += w * mi.x; si.y += w * mi.y; si.z += w * mi.z; si.w += w * mi.w;
        }
    } else {
        sr = *reinterpret_cast<const float4 *>(mhat + ((long)l * 2 + 0) * PL + k4 * 4);
        si = *reinterpret_cast<const float4 *>(mhat + ((long)l * 2 + 1) * PL + k4 * 4);
    }
    const float4 hr = *reinterpret_cast<const float4 *>(sotf + ((long)l * 2 + 0) * PL + k4 * 4);
    const float4 hi = *reinterpret_cast<const float4 *>(sotf + ((long)l * 2 + 1) * PL + k4 * 4);
    float4 xr, xi;
    xr.x = hr.x * sr.x - hi.x * si.x; xi.x = hr.x * si.x + hi.x * sr.x;
    xr.y = hr.y * sr.y - hi.y * si.y; xi.y = hr.y * si.y + hi.y * sr.y;
    xr.z = hr.z * sr.z - hi.z * si.z; xi.z = hr.z * si.z + hi.z * sr.z;
    xr.w = hr.w * sr.w - hi.w * si.w; xi.w = hr.w * si.w + hi.w * sr.w;
    *reinterpret_cast<float4 *>(spec + ((long)l * 2 + 0) * PL + k4 * 4) = xr;
    *reinterpret_cast<float4 *>(spec + ((long)l * 2 + 1) * PL + k4 * 4) = xi;
}

// partial[chunk][t][c][PL] = sum_{l in chunk} tpl[t,l] conj(H[l]) Y[l], templates t0..t0+3
__global__ __launch_bounds__(TPB) void specmix_adj_partial_kernel(const float *__restrict__ spec,
                                                                  const float *__restrict__ sotf,
                                                                  const float *__restrict__ tpl,
                                                                  float *__restrict__ partial, int T, int t0, int L,
                                                                  long PL, int nchunk) {
    const long k4 = (long)blockIdx.x * TPB + threadIdx.x;
    if (k4 * 4 >= PL) return;
    const int ch = blockIdx.y;
    const int per = (L + nchunk - 1) / nchunk;
    const int l0 = ch * per, l1 = min(L, l0 + per);
    float4 ar[4], ai[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ar[g] = ai[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = l0; l < l1; ++l) {
        const float4 hr = *reinterpret_cast<const float4 *>(sotf + ((long)l * 2 + 0) * PL + k4 * 4);
        const float4 hi = *reinterpret_cast<const float4 *>(sotf + ((long)l * 2 + 1) * PL + k4 * 4);
        const float4 yr = *reinterpret_cast<const float4 *>(spec + ((long)l * 2 + 0) * PL + k4 * 4);
        const float4 yi = *reinterpret_cast<const float4 *>(spec + ((long)l * 2 + 1) * PL + k4 * 4);
        float4 pr, pi;
        pr.x = hr.x * yr.x + hi.x * yi.x; pi.x = hr.x * yi.x - hi.x * yr.x;
        pr.y = hr.y * yr.y + hi.y * yi.y; pi.y = hr.y * yi.y - hi.y * yr.y;
        pr.z = hr.z * yr.z + hi.z * yi.z; pi.z = hr.z * yi.z - hi.z * yr.z;
        pr.w = hr.w * yr.w + hi.w * yi.w; pi.w = hr.w * yi.w - hi.w * yr.w;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float w = (t0 + g < T) ? tpl[(long)(t0 + g) * L + l] : 0.f;
            ar[g].x += w * pr.x; ar[g].y += w * pr.y; ar[g].z += w * pr.z; ar[g].w += w * pr.w;
            ai[g].x += w * pi.x; ai[g].y += w * pi.y; ai[g].z += w * pi.z; ai[g].w += w * pi.w;
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (t0 + g < T) {
            float *p = partial + (((long)ch * T + (t0 + g)) * 2) * PL + k4 * 4;
            *reinterpret_cast<float4 *>(p) = ar[g];
            *reinterpret_cast<float4 *>(p + PL) = ai[g];
        }
}

// madj[i] = sum_ch partial[ch][i],  i over T*2*PL
__global__ __launch_bounds__(TPB) void chunk_reduce_kernel(const float *__restrict__ partial, float *__restrict__ out,
                                                           long n, int nchunk) {
    const long i4 = (long)blockIdx.x * TPB + threadIdx.x;
    if (i4 * 4 >= n) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < nchunk; ++c) {
        const float4 v = *reinterpret_cast<const float4 *>(partial + (long)c * n + i4 * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4 *>(out + i4 * 4) = a;
}

// no-LMM adjoint: out[l] = conj(H[l]) Y[l]
__global__ __launch_bounds__(TPB) void specmix_adj_plane_kernel(const float *__restrict__ spec,
                                                                const float *__restrict__ sotf,
                                                                float *__restrict__ out, long PL) {
    const long k4 = (long)blockIdx.x * TPB + threadIdx.x;
    if (k4 * 4 >= PL) return;
    const long l = blockIdx.y;
    const float4 hr = *reinterpret_cast<const float4 *>(sotf + (l * 2 + 0) * PL + k4 * 4);
    const float4 hi = *reinterpret_cast<const float4 *>(sotf + (l * 2 + 1) * PL + k4 * 4);
    const float4 yr = *reinterpret_cast<const float4 *>(spec + (l * 2 + 0) * PL + k4 * 4);
    const float4 yi = *reinterpret_cast<const float4 *>(spec + (l * 2 + 1) * PL + k4 * 4);
    float4 pr, pi;
    pr.x = hr.x * yr.x + hi.x * yi.x; pi.x = hr.x * yi.x - hi.x * yr.x;
    pr.y = hr.y * yr.y + hi.y * yi.y; pi.y = hr.y * yi.y - hi.y * yr.y;
    pr.z = hr.z * yr.z + hi.z * yi.z; pi.z = hr.z * yi.z - hi.z * yr.z;
    pr.w = hr.w * yr.w + hi.w * yi.w; pi.w = hr.w * yi.w - hi.w * yr.w;
    *reinterpret_cast<float4 *>(out + (l * 2 + 0) * PL + k4 * 4) = pr;
    *reinterpret_cast<float4 *>(out + (l * 2 + 1) * PL + k4 * 4) = pi;
}

// ---------------------------------------------------------------------------------------------
// ELL sparse gather, LB lambda planes per thread so one table read serves LB planes
// ---------------------------------------------------------------------------------------------
template <int LB>
__global__ __launch_bounds__(TPB) void spmm_ell_kernel(EllTable t, const float *__restrict__ src, long srcStride,
                                                       float *__restrict__ dst, long dstStride, int nblk,
                                                       int accumulate) {
    const int r = blockIdx.x * TPB + threadIdx.x;
    if (r >= t.R) return;
    const int b0 = blockIdx.y * LB;
    const float *sp[LB];
#pragma unroll
    for (int l = 0; l < LB; ++l) sp[l] = src + (long)min(b0 + l, nblk - 1) * srcStride;
    float acc[LB];
#pragma unroll
    for (int l = 0; l < LB; ++l) acc[l] = 0.f;
    const int n = t.cnt[r];
    for (int e = 0; e < n; ++e) {
        const int c = t.col[(long)e * t.R + r];
        const float v = t.val[(long)e * t.R + r];
#pragma unroll
        for (int l = 0; l < LB; ++l) acc[l] += v * sp[l][c];
    }
    const long off = t.dst_off[r];
#pragma unroll
    for (int l = 0; l < LB; ++l)
        if (b0 + l < nblk) {
            float *p = dst + (long)(b0 + l) * dstStride + off;
            if (accumulate)
                *p += acc[l];
            else
                *p = acc[l];
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

__global__ __launch_bounds__(TPB) void y_from_cpart_kernel(const float *__restrict__ cpart, long slab, int nsplit,
                                                           float *__restrict__ y, int PS, int Ldet, int aout, int NP) {
    const long i = (long)blockIdx.x * TPB + threadIdx.x;
    const long n = (long)PS * Ldet * aout;
    if (i >= n) return;
    const int a = i % aout;
    const int l = (i / aout) % Ldet;
    const int ps = i / ((long)aout * Ldet);
    const long src = (long)l * NP + ps * aout + a;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += cpart[k * slab + src];
    y[i] = s;
}

__global__ __launch_bounds__(TPB) void ymat_from_y_kernel(const float *__restrict__ y, float *__restrict__ ymat, int PS,
                                                          int Ldet, int aout, int NP) {
    const int n = blockIdx.x * TPB + threadIdx.x;
    const int l = blockIdx.y;
    if (n >= PS * aout) return;
    const int ps = n / aout, a = n % aout;
    ymat[(long)l * NP + n] = y[((long)ps * Ldet + l) * aout + a];
}

__global__ __launch_bounds__(TPB) void fill_zero_kernel(float *p, long n) {
    long i = (long)blockIdx.x * TPB + threadIdx.x;
    const long stride = (long)gridDim.x * TPB;
    for (; i < n; i += stride) p[i] = 0.f;
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

__global__ __launch_bounds__(TPB) void residual_kernel(float *__restrict__ r, const float *__restrict__ b,
                                                       const float *__restrict__ q, long n) {
    const long stride = (long)gridDim.x * TPB;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) r[i] = b[i] - q[i];
}

inline int nblocks(long n, int cap = 2048) {
    long b = (n + TPB - 1) / TPB;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}
constexpr int DOT_BLOCKS = 512;

}  // namespace

int launch_specmix_fwd(hipStream_t s, const float *mhat, const float *sotf, const float *tpl, float *spec, int T,
                       int L, long PL) {
    dim3 grid((unsigned)((PL / 4 + TPB - 1) / TPB), L);
    hipLaunchKernelGGL(specmix_fwd_kernel, grid, dim3(TPB), 0, s, mhat, sotf, tpl, spec, T, L, PL);
    return (int)hipGetLastError();
}

int launch_specmix_adj(hipStream_t s, const float *spec, const float *sotf, const float *tpl, float *partial,
                       float *madj, int T, int L, long PL, int nchunk) {
    if (T == 0) {
        dim3 grid((unsigned)((PL / 4 + TPB - 1) / TPB), L);
        hipLaunchKernelGGL(specmix_adj_plane_kernel, grid, dim3(TPB), 0, s, spec, sotf, madj, PL);
        return (int)hipGetLastError();
    }
    dim3 grid((unsigned)((PL / 4 + TPB - 1) / TPB), nchunk);
    for (int t0 = 0; t0 < T; t0 += 4)
        hipLaunchKernelGGL(specmix_adj_partial_kernel, grid, dim3(TPB), 0, s, spec, sotf, tpl, partial, T, t0, L, PL,
                           nchunk);
    const long n = (long)T * 2 * PL;
    hipLaunchKernelGGL(chunk_reduce_kernel, dim3((unsigned)((n / 4 + TPB - 1) / TPB)), dim3(TPB), 0, s, partial, madj,
                       n, nchunk);
    return (int)hipGetLastError();
}

int launch_spmm_ell(hipStream_t s, const EllTable &t, const float *src, long srcStride, float *dst, long dstStride,
                    int nblk, int accumulate) {
    if (t.R == 0 || nblk == 0) return 0;
    constexpr int LB = 4;
    dim3 grid((t.R + TPB - 1) / TPB, (nblk + LB - 1) / LB);
    hipLaunchKernelGGL((spmm_ell_kernel<LB>), grid, dim3(TPB), 0, s, t, src, srcStride, dst, dstStride, nblk,
                       accumulate);
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
                        int aout, int NP) {
    const long n = (long)PS * Ldet * aout;
    hipLaunchKernelGGL(y_from_cpart_kernel, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, s, cpart, slab,
                       nsplit, y, PS, Ldet, aout, NP);
    return (int)hipGetLastError();
}

int launch_ymat_from_y(hipStream_t s, const float *y, float *ymat, int PS, int Ldet, int aout, int NP) {
    dim3 grid((PS * aout + TPB - 1) / TPB, Ldet);
    hipLaunchKernelGGL(ymat_from_y_kernel, grid, dim3(TPB), 0, s, y, ymat, PS, Ldet, aout, NP);
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

int launch_cg_xupdate(hipStream_t s, float *x, const float *d, long n, const double *rr, const double *dq) {
    hipLaunchKernelGGL(cg_xupdate_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, x, d, n, rr, dq);
    return (int)hipGetLastError();
}

int launch_cg_dir(hipStream_t s, float *d, const float *r, long n, const double *rr_new, const double *rr_old) {
    hipLaunchKernelGGL(cg_dir_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, d, r, n, rr_new, rr_old);
    return (int)hipGetLastError();
}

int launch_residual(hipStream_t s, float *r, const float *b, const float *q, long n) {
    hipLaunchKernelGGL(residual_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, r, b, q, n);
    return (int)hipGetLastError();
}
