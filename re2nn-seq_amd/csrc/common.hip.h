// Shared helpers for the FA-RNN tagging kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "../../include/farnn.h"

namespace farnn {

constexpr int WAVE = 64;   // CDNA wavefront width; hard-coded on purpose (gfx950 only)

// ---- error plumbing: no exceptions cross the C ABI -------------------------------------------
extern thread_local char g_err[512];

inline int fail(int code, const char *fmt, const char *a = "", const char *b = "") {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}

#define FARNN_HIP_TRY(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            snprintf(farnn::g_err, sizeof(farnn::g_err), "%s failed: %s (%s:%d)", #expr,      \
                     hipGetErrorString(_e), __FILE__, __LINE__);                              \
            return (_e == hipErrorOutOfMemory) ? FARNN_ENOMEM : FARNN_EIO;                    \
        }                                                                                     \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline size_t round_up_sz(size_t v, size_t m) { return (v + m - 1) / m * m; }

// ---- device helpers ---------------------------------------------------------------------------
// update_nonlinear dispatch (reference model_onehot.py:379-386).  `nl` is wave-uniform.
__device__ __forceinline__ float apply_nl(float v, int nl) {
    switch (nl) {
        case FARNN_NL_RELU:     return fmaxf(v, 0.0f);
        case FARNN_NL_TANH:     return tanhf(v);
        case FARNN_NL_RELUTANH: return tanhf(fmaxf(v, 0.0f));
        case FARNN_NL_SIGMOID:  return 1.0f / (1.0f + expf(-v));
        default:                return v;
    }
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// (value desc, index asc) combine for first-index argmax, matching torch.max tie-breaking.
__device__ __forceinline__ void argmax_combine(float &v, int &i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__device__ __forceinline__ void wave_argmax(float &v, int &i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ov = __shfl_xor(v, off, WAVE);
        int oi = __shfl_xor(i, off, WAVE);
        argmax_combine(v, i, ov, oi);
    }
}

}  // namespace farnn
