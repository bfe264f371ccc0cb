// Host-side helpers shared by the translation units of libfarnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <map>
#include <utility>
#include "common.hip.h"

namespace farnn {

inline int select_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(FARNN_ENODEV, "no HIP device visible%s%s");
    if (device < 0 || device >= n) return fail(FARNN_EINVAL, "device index out of range%s%s");
    FARNN_HIP_TRY(hipSetDevice(device));
    return FARNN_OK;
}

inline int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Raise a kernel's dynamic-LDS limit (needed above 48 KiB).  The attribute is sticky, so it is set
// only when a kernel needs more than it was last given (a host API call per launch otherwise).
template <typename KernelT>
inline int raise_lds_limit(KernelT kern, size_t bytes) {
    static std::map<std::pair<int, const void *>, size_t> granted;    // (device, kernel) -> bytes
    if (bytes > 160 * 1024) return fail(FARNN_ERANGE, "kernel needs more than 160 KiB of LDS%s%s");
    if (bytes <= 48 * 1024) return FARNN_OK;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void *fn = reinterpret_cast<const void *>(kern);
    size_t &g = granted[{dev, fn}];
    if (bytes > g) {
        FARNN_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        g = bytes;
    }
    return FARNN_OK;
}

}  // namespace farnn
