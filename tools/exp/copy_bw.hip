// achievable HBM rates on this chip for the access mix of the transform passes: pure read, pure write, copy
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void copy_k(const float4 *__restrict__ a, float4 *__restrict__ b, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void read_k(const float4 *__restrict__ a, float *out, long n) {
    float4 s = make_float4(0, 0, 0, 0);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { float4 v = a[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (s.x + s.y + s.z + s.w == 123.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void write_k(float4 *__restrict__ b, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = make_float4(1, 2, 3, 4);
}
int main() {
    const long n = (long)1 << 26;   // float4 elements = 1 GiB
    float4 *a, *b; float *o;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {1024, 4096, 16384}) {
        for (int mode = 0; mode < 3; ++mode) {
            auto run = [&]() {
                if (mode == 0) hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n);
                else if (mode == 1) hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, o, n);
                else hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, b, n);
            };
            run(); run();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) run();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            const double bytes = (mode == 0 ? 2.0 : 1.0) * n * 16;
            printf("grid %5d %-5s %.4f ms  %.2f TB/s\n", grid, mode == 0 ? "copy" : mode == 1 ? "read" : "write", ms, bytes / ms * 1e-9);
        }
    }
    return 0;
}
