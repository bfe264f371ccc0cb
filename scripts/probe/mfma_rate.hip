// Measures the issue rate of the f32 MFMA shapes on one SIMD and on the whole chip (probe, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(float *out, int iters, long long *clk) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0;
    f32x16 c0 = {0}, c1 = c0;
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
        } else if (MODE == 1) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c1, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a4, 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float s = a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + c0[0] + c1[5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}

template <int MODE>
static void run(const char *name, int per_iter, double flop_per, int blocks) {
    float *out; long long *clk, h;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<blocks, 256>>>(out, 100, clk);
    hipEventRecord(e0);
    rate_kernel<MODE><<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * per_iter;
    printf("%-12s blocks=%5d  %.1f clock64-ticks/MFMA/wave  %.2f ns/MFMA/wave  chip %.1f TFLOP/s\n", name, blocks,
           (double)h / n, ms * 1e6 / n, n * flop_per * blocks * 4 / (ms * 1e-3) * 1e-12);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int blocks : {1, 256, 512, 1024}) {
        run<0>("16x16x4f32", 5, 2048.0, blocks);
        run<1>("32x32x2f32", 2, 4096.0, blocks);
        run<2>("4x4x1f32", 5, 512.0, blocks);
    }
    return 0;
}
