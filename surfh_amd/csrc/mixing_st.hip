// Masked linear mixing model ("MixingST", surfh/Models/mixing.py:276-337 with the kernels of
// surfh/ToolsDir/cythons_files.pyx:370-463): the cube exists only on a list of selected voxels.
//
//   forward : cube[l][i][j] = sum_m maps[m][i][j] templates[m][l]     for (l, i, j) in the list, 0 elsewhere
//   adjoint : maps[m][i][j] = sum_{l : (l,i,j) in the list} cube[l][i][j] templates[m][l]
//   fwadj   : out[m][i][j]  = sum_m' TST[m][m'][i][j] maps[m'][i][j],  TST = sum_l templates[m'][l] templates[m][l] S[l][i][j]
//
// The voxel list is regrouped by pixel at construction (stable: the per-pixel wavelength order of the caller's list is
// kept), so the adjoint is a deterministic per-pixel loop instead of the scattered += of the reference.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/surfh_amd.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXT = 8;

thread_local std::string g_tst_err;
int tfail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_tst_err = buf;
    return 1;
}
#define T_OK(x)                                                                                   \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) return tfail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// one thread per listed voxel; a voxel listed twice is counted twice, like the reference's +=
__global__ __launch_bounds__(TPB) void tst_forward_kernel(const int *__restrict__ vox, long n, const float *__restrict__ maps,
                                                          const float *__restrict__ tpl, float *__restrict__ cube, int T, int L, long npix,
                                                          int nb) {
    const long v = (long)blockIdx.x * TPB + threadIdx.x;
    if (v >= n) return;
    const int l = vox[3 * v], i = vox[3 * v + 1], j = vox[3 * v + 2];
    const long px = (long)i * nb + j;
    float s = 0.f;
    for (int m = 0; m < T; ++m) s += maps[(long)m * npix + px] * tpl[(long)m * L + l];
    atomicAdd(cube + (long)l * npix + px, s);
}

// one thread per pixel walks that pixel's wavelengths in list order
__global__ __launch_bounds__(TPB) void tst_adjoint_kernel(const long *__restrict__ off, const int *__restrict__ lam,
                                                          const float *__restrict__ cube, const float *__restrict__ tpl,
                                                          float *__restrict__ maps, int T, int L, long npix) {
    const long px = (long)blockIdx.x * TPB + threadIdx.x;
    if (px >= npix) return;
    float acc[MAXT];
#pragma unroll
    for (int m = 0; m < MAXT; ++m) acc[m] = 0.f;
    for (long e = off[px]; e < off[px + 1]; ++e) {
        const int l = lam[e];
        const float c = cube[(long)l * npix + px];
#pragma unroll
        for (int m = 0; m < MAXT; ++m)
            if (m < T) acc[m] += c * tpl[(long)m * L + l];
    }
#pragma unroll
    for (int m = 0; m < MAXT; ++m)
        if (m < T) maps[(long)m * npix + px] = acc[m];
}

// TST[m][mp][px] = sum_l tpl[mp][l] tpl[m][l] S[l][px]   (cythons_files.pyx:374-392, same summation order)
__global__ __launch_bounds__(TPB) void tst_precompute_kernel(const float *__restrict__ S, const float *__restrict__ tpl,
                                                             float *__restrict__ TST, int T, int L, long npix) {
    const long px = (long)blockIdx.x * TPB + threadIdx.x;
    const int m = blockIdx.y / T, mp = blockIdx.y % T;
    if (px >= npix) return;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += tpl[(long)mp * L + l] * tpl[(long)m * L + l] * S[(long)l * npix + px];
    TST[((long)m * T + mp) * npix + px] = acc;
}

__global__ __launch_bounds__(TPB) void tst_fwadj_kernel(const float *__restrict__ TST, const float *__restrict__ maps,
                                                        float *__restrict__ out, int T, long npix) {
    const long px = (long)blockIdx.x * TPB + threadIdx.x;
    const int m = blockIdx.y;
    if (px >= npix) return;
    float acc = 0.f;
    for (int mp = 0; mp < T; ++mp) acc += TST[((long)m * T + mp) * npix + px] * maps[(long)mp * npix + px];
    out[(long)m * npix + px] = acc;
}

}  // namespace

struct surfh_tst {
    int dev = 0, T = 0, L = 0, na = 0, nb = 0;
    long npix = 0, nvox = 0;
    int *vox = nullptr, *lam = nullptr;
    long *off = nullptr;
    float *tpl = nullptr, *TST = nullptr, *maps = nullptr, *cube = nullptr, *out = nullptr;
};

extern "C" {

const char *surfh_tst_last_error(void) { return g_tst_err.c_str(); }

int surfh_tst_destroy(surfh_tst *t) {
    if (!t) return 0;
    hipSetDevice(t->dev);
    hipFree(t->vox); hipFree(t->lam); hipFree(t->off); hipFree(t->tpl); hipFree(t->TST); hipFree(t->maps); hipFree(t->cube); hipFree(t->out);
    delete t;
    return 0;
}

int surfh_tst_create(int32_t n_alpha, int32_t n_beta, int32_t n_lambda, int32_t n_templates, const double *templates,
                     const int32_t *voxels, int64_t n_voxels, const float *S, int32_t device, surfh_tst **out) {
    if (!templates || !out || (n_voxels > 0 && !voxels)) return tfail("null argument");
    if (n_alpha < 1 || n_beta < 1 || n_lambda < 1 || n_templates < 1 || n_templates > MAXT || n_voxels < 0)
        return tfail("bad shape (templates: 1..%d)", MAXT);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return tfail("no HIP device: the masked mixing model has no CPU fallback");
    T_OK(hipSetDevice(device));
    surfh_tst *t = new surfh_tst;
    t->dev = device; t->T = n_templates; t->L = n_lambda; t->na = n_alpha; t->nb = n_beta;
    t->npix = (long)n_alpha * n_beta; t->nvox = n_voxels;
    auto bail = [&](int rc) { surfh_tst_destroy(t); return rc; };
    // voxel list checked and regrouped by pixel, keeping the caller's order inside a pixel
    std::vector<long> cnt((size_t)t->npix + 1, 0);
    for (int64_t v = 0; v < n_voxels; ++v) {
        const int l = voxels[3 * v], i = voxels[3 * v + 1], j = voxels[3 * v + 2];
        if (l < 0 || l >= n_lambda || i < 0 || i >= n_alpha || j < 0 || j >= n_beta) return bail(tfail("voxel %lld = (%d,%d,%d) outside the cube", (long long)v, l, i, j));
        ++cnt[(size_t)i * n_beta + j + 1];
    }
    for (long p = 0; p < t->npix; ++p) cnt[p + 1] += cnt[p];
    std::vector<int> lam((size_t)std::max<int64_t>(n_voxels, 1));
    {
        std::vector<long> pos(cnt.begin(), cnt.end() - 1);
        for (int64_t v = 0; v < n_voxels; ++v) lam[pos[(size_t)voxels[3 * v + 1] * n_beta + voxels[3 * v + 2]]++] = voxels[3 * v];
    }
    std::vector<float> tp((size_t)n_templates * n_lambda);
    for (size_t k = 0; k < tp.size(); ++k) tp[k] = (float)templates[k];   // the reference casts to float32 (mixing.py:307)
    const size_t nv = (size_t)std::max<int64_t>(n_voxels, 1);
    if (hipMalloc((void **)&t->vox, nv * 3 * sizeof(int)) != hipSuccess || hipMalloc((void **)&t->lam, nv * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&t->off, cnt.size() * sizeof(long)) != hipSuccess || hipMalloc((void **)&t->tpl, tp.size() * 4) != hipSuccess ||
        hipMalloc((void **)&t->maps, (size_t)n_templates * t->npix * 4) != hipSuccess ||
        hipMalloc((void **)&t->out, (size_t)n_templates * t->npix * 4) != hipSuccess ||
        hipMalloc((void **)&t->cube, (size_t)n_lambda * t->npix * 4) != hipSuccess)
        return bail(tfail("device allocation failed"));
    if (n_voxels) {
        if (hipMemcpy(t->vox, voxels, (size_t)n_voxels * 3 * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(t->lam, lam.data(), (size_t)n_voxels * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            return bail(tfail("copy failed"));
    }
    if (hipMemcpy(t->off, cnt.data(), cnt.size() * sizeof(long), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t->tpl, tp.data(), tp.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        return bail(tfail("copy failed"));
    if (S) {   // TST from the caller's mask (fast_precompute_TST, mixing.py:319-327)
        if (hipMalloc((void **)&t->TST, (size_t)n_templates * n_templates * t->npix * 4) != hipSuccess) return bail(tfail("device allocation failed"));
        if (hipMemcpy(t->cube, S, (size_t)n_lambda * t->npix * 4, hipMemcpyHostToDevice) != hipSuccess) return bail(tfail("copy failed"));
        dim3 grid((unsigned)((t->npix + TPB - 1) / TPB), (unsigned)(n_templates * n_templates));
        hipLaunchKernelGGL(tst_precompute_kernel, grid, dim3(TPB), 0, 0, t->cube, t->tpl, t->TST, n_templates, n_lambda, t->npix);
        if (hipDeviceSynchronize() != hipSuccess) return bail(tfail("TST precompute failed"));
    }
    *out = t;
    return 0;
}

int surfh_tst_forward(surfh_tst *t, const float *maps, float *cube) {
    if (!t || !maps || !cube) return tfail("null argument");
    T_OK(hipSetDevice(t->dev));
    T_OK(hipMemcpy(t->maps, maps, (size_t)t->T * t->npix * 4, hipMemcpyHostToDevice));
    T_OK(hipMemset(t->cube, 0, (size_t)t->L * t->npix * 4));
    if (t->nvox)
        hipLaunchKernelGGL(tst_forward_kernel, dim3((unsigned)((t->nvox + TPB - 1) / TPB)), dim3(TPB), 0, 0, t->vox, t->nvox, t->maps, t->tpl,
                           t->cube, t->T, t->L, t->npix, t->nb);
    T_OK(hipGetLastError());
    T_OK(hipMemcpy(cube, t->cube, (size_t)t->L * t->npix * 4, hipMemcpyDeviceToHost));
    return 0;
}

int surfh_tst_adjoint(surfh_tst *t, const float *cube, float *maps) {
    if (!t || !maps || !cube) return tfail("null argument");
    T_OK(hipSetDevice(t->dev));
    T_OK(hipMemcpy(t->cube, cube, (size_t)t->L * t->npix * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(tst_adjoint_kernel, dim3((unsigned)((t->npix + TPB - 1) / TPB)), dim3(TPB), 0, 0, t->off, t->lam, t->cube, t->tpl, t->maps,
                       t->T, t->L, t->npix);
    T_OK(hipGetLastError());
    T_OK(hipMemcpy(maps, t->maps, (size_t)t->T * t->npix * 4, hipMemcpyDeviceToHost));
    return 0;
}

int surfh_tst_fwadj(surfh_tst *t, const float *maps, float *out) {
    if (!t || !maps || !out) return tfail("null argument");
    if (!t->TST) return tfail("fwadj needs the mask S at creation (MixingST.fast_precompute_TST)");
    T_OK(hipSetDevice(t->dev));
    T_OK(hipMemcpy(t->maps, maps, (size_t)t->T * t->npix * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(tst_fwadj_kernel, dim3((unsigned)((t->npix + TPB - 1) / TPB), (unsigned)t->T), dim3(TPB), 0, 0, t->TST, t->maps, t->out, t->T,
                       t->npix);
    T_OK(hipGetLastError());
    T_OK(hipMemcpy(out, t->out, (size_t)t->T * t->npix * 4, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
