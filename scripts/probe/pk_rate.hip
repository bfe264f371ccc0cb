// Probe (not product code): issue cost of INDEPENDENT f32 vector instructions on one SIMD -- scalar add / fma against the packed
// forms -- with one, two and four wavefronts per SIMD.  The Viterbi forward step is two adds and a max per (source, tag) pair:
// if v_pk_fma_f32(a, 1.0, b) (= a + b, one rounding: the same bits as an add) issues faster than v_pk_add_f32, the step's adds
// are cheaper as FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, long long *clk, int iters) {
    const int lane = threadIdx.x;
    v2f a[8];
    for (int i = 0; i < 8; i++) a[i] = v2f{lane * 0.001f + i, lane * 0.002f - i};
    v2f b = v2f{1.0f + lane * 1e-6f, 0.5f}, one = v2f{1.0f, 1.0f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y)); }
            if (MODE == 1) { asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(a[i].x) : "v"(b.x), "v"(one.x)); asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(a[i].y) : "v"(b.y), "v"(one.x)); }
            if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %2, %1" : "+v"(a[i]) : "v"(b), "v"(one));
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(one));
            if (MODE == 5) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y)); }
            if (MODE == 6) { asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(b.y)); asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(b.y), "v"(b.x)); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int per_iter_scalar_ops) {
    float *out; long long *clk, h;
    (void)hipMalloc(&out, 1024 * 4); (void)hipMalloc(&clk, 8);
    const int iters = 2000;
    printf("%-44s", name);
    for (int waves : {4, 8, 16}) {                       // 1, 2, 4 wavefronts per SIMD
        k<MODE><<<1, waves * 64>>>(out, clk, 10);
        k<MODE><<<1, waves * 64>>>(out, clk, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
        printf("  %d/SIMD: %5.2f cycles per f32 op per SIMD", waves / 4, (double)h / ((double)iters * per_iter_scalar_ops * (waves / 4)));
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    printf("(cycles of one wavefront's loop / (f32 lane-operations per lane x wavefronts per SIMD): 4.0 = one 64-lane op per 4 cycles)\n");
    run<0>("v_add_f32 x2", 16);
    run<1>("v_fma_f32 x2 (a * 1 + b)", 16);
    run<2>("v_pk_add_f32", 16);
    run<3>("v_pk_fma_f32 (a * 1 + b)", 16);
    run<4>("v_pk_mul_f32", 16);
    run<5>("v_max_f32 x2", 16);
    run<6>("v_max3_f32 x2", 16);
    return 0;
}
