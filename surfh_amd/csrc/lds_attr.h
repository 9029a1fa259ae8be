// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised -- once per
// kernel AND per device (a process may hold plans on several devices).
#pragma once
#include <hip/hip_runtime.h>

template <typename K>
inline int ensure_dynamic_lds(K kernel, size_t bytes, unsigned long long &done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0 && dev < 64 && ((done_mask >> dev) & 1ull)) return 0;
    e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0 && dev < 64) done_mask |= 1ull << dev;
    return 0;
}
