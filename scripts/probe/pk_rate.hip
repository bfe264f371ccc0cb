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
    if ((threadIdx.x & 63) == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; }   // every wavefront: the oldest one alone never waits
}

template <int MODE>
static void run(const char *name, int per_iter_instrs) {
    float *out; long long *clk, h[32];
    (void)hipMalloc(&out, 1024 * 4); (void)hipMalloc(&clk, 32 * 8);
    const int iters = 2000;
    printf("%-44s", name);
    for (int waves : {4, 8, 16}) {                       // 1, 2, 4 wavefronts per SIMD
        k<MODE><<<1, waves * 64>>>(out, clk, 10);
        k<MODE><<<1, waves * 64>>>(out, clk, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, clk, waves * 16, hipMemcpyDeviceToHost);
        long long lo = h[0], hi = h[1];
        for (int w = 0; w < waves; w++) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; }
        // first start to last end of all wavefronts, per instruction a SIMD issued (per_iter_instrs per wavefront and iteration)
        printf("  %d/SIMD: %5.2f cycles per instruction per SIMD", waves / 4, (double)(hi - lo) / ((double)iters * per_iter_instrs * (waves / 4)));
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    printf("(first start to last end over all wavefronts / instructions one SIMD issued; the x2 rows issue 16 instructions per iteration, the packed rows 8)\n");
    run<0>("v_add_f32 x2", 16);
    run<1>("v_fma_f32 x2 (a * 1 + b)", 16);
    run<2>("v_pk_add_f32", 8);
    run<3>("v_pk_fma_f32 (a * 1 + b)", 8);
    run<4>("v_pk_mul_f32", 8);
    run<5>("v_max_f32 x2", 16);
    run<6>("v_max3_f32 x2", 16);
    return 0;
}
