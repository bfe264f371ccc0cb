// LDS read cost by access shape (probe): one workgroup of `threads` threads per CU, loops like the ones in
// train_crf_kernel / matvec2_partial.  Prints cycles per LDS read instruction as seen by one wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024) lds_probe(float *out, long long *clk, int iters, int n) {
    extern __shared__ float smem[];
    float *v = smem;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) v[i] = i * 0.5f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {                       // broadcast scalar reads, one dependent max chain (unroll 8)
#pragma unroll 8
            for (int i = 0; i < n; i++) a0 = fmaxf(a0, v[i] + it);
        } else if (MODE == 1) {                // broadcast scalar reads, 4 independent chains, unroll 16
#pragma unroll 16
            for (int i = 0; i < n; i += 4) { a0 += v[i] * it; a1 += v[i + 1] * it; a2 += v[i + 2] * it; a3 += v[i + 3] * it; }
        } else if (MODE == 2) {                // per-lane consecutive addresses (conflict-free), 4 chains
#pragma unroll 16
            for (int i = 0; i < n; i += 4) { a0 += v[i * 64 + lane] * it; a1 += v[(i + 1) * 64 + lane] * it; a2 += v[(i + 2) * 64 + lane] * it; a3 += v[(i + 3) * 64 + lane] * it; }
        } else if (MODE == 3) {                // broadcast 16-byte reads, 4 chains
#pragma unroll 4
            for (int i = 0; i < n; i += 4) { const v4f x = *(const v4f *)(v + i); a0 += x[0] * it; a1 += x[1] * it; a2 += x[2] * it; a3 += x[3] * it; }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}

template <int MODE>
static void run(const char *name, int threads, int n, int reads_per_pass) {
    float *out; long long *clk, h;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 8);
    const int iters = 200;
    hipFuncSetAttribute((const void *)lds_probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    lds_probe<MODE><<<256, threads, 65536>>>(out, clk, iters, n);
    hipDeviceSynchronize();
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-44s %4d threads: %.1f cycles per LDS read instruction, %.1f per element\n", name, threads,
           (double)h / (iters * reads_per_pass), (double)h / (iters * n));
    hipFree(out); hipFree(clk);
}

int main() {
    for (int threads : {64, 256, 512, 1024}) {
        run<0>("broadcast b32, one max chain, unroll 8", threads, 72, 72);
        run<1>("broadcast b32, 4 chains, unroll 16", threads, 64, 64);
        run<2>("lane-consecutive b32, 4 chains, unroll 16", threads, 64, 64);
        run<3>("broadcast b128, 4 chains", threads, 64, 16);
    }
    return 0;
}
