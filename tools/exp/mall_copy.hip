// Does a chunked two-step copy through a small, reused intermediate run at the speed of ONE copy?  (i.e. does the
// intermediate live in the 256 MB Infinity Cache: written by step 1, read back by step 2, never paid for in HBM time)
//   baseline : src (1 GiB) -> dst (1 GiB)                      one launch
//   two-pass : src -> tmp (1 GiB) -> dst                       two launches over the whole arrays (what the DFT passes do)
//   chunked  : per chunk of C MB: src[c] -> tmp (C MB), tmp -> dst[c]    one stream, and two streams with two tmps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void copy_k(const float4 *__restrict__ a, float4 *__restrict__ b, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i];
}
int main() {
    const long n = (long)1 << 26;   // float4 elements = 1 GiB
    float4 *a, *b, *t;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&t, n * 16));
    CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 0, n * 16)); CK(hipMemset(t, 0, n * 16));
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    hipEvent_t e0, e1, j; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
    const int grid = 2048, reps = 10;
    auto timeit = [&](const char *name, auto fn) {
        fn(); fn();
        hipDeviceSynchronize();
        hipEventRecord(e0, s0);
        for (int i = 0; i < reps; ++i) fn();
        hipEventRecord(e1, s0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("%-34s %.4f ms   (1 GiB in + 1 GiB out = %.2f TB/s)\n", name, ms, 2.0 * n * 16 / ms * 1e-9);
    };
    timeit("baseline one copy", [&]() { hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, s0, a, b, n); });
    timeit("two full passes via 1 GiB tmp", [&]() {
        hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, s0, a, t, n);
        hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, s0, t, b, n);
    });
    for (long mb : {16, 32, 64, 96, 128}) {
        const long c = mb * 1024 * 1024 / 16;
        char nm[64];
        snprintf(nm, sizeof nm, "chunked %3ld MB, one stream", mb);
        timeit(nm, [&]() {
            for (long o = 0; o < n; o += c) {
                const long m = o + c <= n ? c : n - o;
                hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, s0, a + o, t, m);
                hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, s0, t, b + o, m);
            }
        });
        snprintf(nm, sizeof nm, "chunked %3ld MB, two streams", mb);
        timeit(nm, [&]() {
            hipEventRecord(j, s0); hipStreamWaitEvent(s1, j, 0);
            int k = 0;
            for (long o = 0; o < n; o += c, ++k) {
                const long m = o + c <= n ? c : n - o;
                hipStream_t s = (k & 1) ? s1 : s0;
                float4 *tt = t + (k & 1) * c;
                hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, s, a + o, tt, m);
                hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, s, tt, b + o, m);
            }
            hipEventRecord(j, s1); hipStreamWaitEvent(s0, j, 0);
        });
    }
    return 0;
}
