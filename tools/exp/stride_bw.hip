// What does the access pattern of the transform passes cost?  Copy 1 GiB -> 1 GiB as tiles of 256 rows x 512 B (128 floats),
// one dword per lane like the register-direct loads of the pass kernels, for several row pitches:
//   pitch 512 B  : a tile is one contiguous 128 KB block
//   pitch 64 KB .. 4 MB : 512-byte pieces at that stride (4 MB = the cube's [beta][alpha][4096 lambda] layout)
// The footprint is 1 GiB in every case; consecutive tiles are adjacent columns of the same row group.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <bool RD, bool WR>
__global__ __launch_bounds__(256, 2) void tile_copy(const float *__restrict__ src, float *__restrict__ dst, long pitchF, long ntile, float *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tpg = pitchF / 128;                       // tiles per row group
    float acc = 0.f;
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long grp = t / tpg, col = t % tpg;
        const long base = grp * 256 * pitchF + col * 128 + wave * 32 + (lane & 31);
        for (int r0 = 0; r0 < 256; r0 += 32) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long o = base + (long)(r0 + 2 * i + (lane >> 5)) * pitchF;
                v[i] = RD ? src[o] : (float)i;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long o = base + (long)(r0 + 2 * i + (lane >> 5)) * pitchF;
                if (WR) dst[o] = v[i]; else acc += v[i];
            }
        }
    }
    if (!WR && acc == 12345.f) sink[0] = acc;
}
int main() {
    const long n = (long)1 << 28;   // floats = 1 GiB
    float *a, *b, *sink;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long ntile = n / (256 * 128);
    for (long pitchB : {512L, 65536L, 131072L, 2097152L, 4194304L}) {
        const long pitchF = pitchB / 4;
        for (int mode = 0; mode < 3; ++mode) {
            auto run = [&]() {
                if (mode == 0) hipLaunchKernelGGL((tile_copy<true, true>), dim3(512), dim3(256), 0, 0, a, b, pitchF, ntile, sink);
                else if (mode == 1) hipLaunchKernelGGL((tile_copy<true, false>), dim3(512), dim3(256), 0, 0, a, b, pitchF, ntile, sink);
                else hipLaunchKernelGGL((tile_copy<false, true>), dim3(512), dim3(256), 0, 0, a, b, pitchF, ntile, sink);
            };
            run(); run();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) run();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            const double bytes = (mode == 0 ? 2.0 : 1.0) * n * 4;
            printf("pitch %8ld B  %-5s %.4f ms  %.2f TB/s\n", pitchB, mode == 0 ? "copy" : mode == 1 ? "read" : "write", ms, bytes / ms * 1e-9);
        }
    }
    return 0;
}
