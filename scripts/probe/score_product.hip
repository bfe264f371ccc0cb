// Probe (not product code): the fused Viterbi kernel's score product -- [16 tokens x S] . [S x 16 tags] units on the f32 matrix
// cores, both operands in LDS, reads issued a state group ahead -- timed per unit with s_memtime for 1..16 wavefronts per
// workgroup and a few variants of the loop, to find what made 28 matrix-core instructions cost 2 000 cycles in the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ void __launch_bounds__(1024) k(float *out, long long *clk, int c16, int SP, int SPq, int units_per_wave) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    float *ab = smem;                       // [64][SPq]
    float *img = smem + 64 * SPq;           // [5][c16][256]
    for (int i = tid; i < 64 * SPq + 5 * c16 * 256; i += blockDim.x) smem[i] = 0.001f * (i % 97);
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    f32x4 total = {0, 0, 0, 0};
    for (int u = 0; u < units_per_wave; u++) {
        const int unit = (w + u * 16) % 20, tb = unit / 5, cb = unit % 5;
        const unsigned a_lane = (unsigned)(size_t)(ab + (tb * 16 + lr) * SPq);
        const unsigned b_lane = (unsigned)(size_t)(img + (size_t)cb * c16 * 256 + lane * 4);
        f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
        auto issue = [&](int g, float (&a)[4], f32x4 &bf) {
            const int gc = g < c16 ? g : c16 - 1;
            unsigned aa[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int s_ = 16 * gc + 4 * e + lk;
                aa[e] = a_lane + 4u * (unsigned)(VARIANT == 1 ? s_ : (s_ < SP ? s_ : SP - 1));
            }
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7"
                         : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]) : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=&v"(bf) : "v"(b_lane + 1024u * (unsigned)gc) : "memory");
        };
        auto landed = [&](float (&a)[4], f32x4 &bf, auto cnt) {
            asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(bf) : "n"(decltype(cnt)::value));
        };
        auto mfma4 = [&](const float (&a)[4], const f32x4 &bf) {
            if (VARIANT == 2) {             // two accumulators (different bits: a throughput experiment only)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], bf.x, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], bf.y, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], bf.z, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], bf.w, acc2, 0, 0, 0);
            } else if (VARIANT == 3) {      // no matrix-core work at all: what the loads, waits and address arithmetic cost
                acc.x += a[0] * bf.x + a[1] * bf.y + a[2] * bf.z + a[3] * bf.w;
            } else {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], bf.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], bf.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], bf.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], bf.w, acc, 0, 0, 0);
            }
        };
        if (VARIANT == 4) {
            // rows padded with zeros to 16 c16 columns: no clamping; ONE address register per operand, advanced once per group; the
            // four A entries of a group are 16 bytes apart -> offset immediates
            unsigned pa = a_lane + 4u * (unsigned)lk, pb = b_lane;
            auto issue4 = [&](float (&a)[4], f32x4 &bf) {
                asm volatile("ds_read_b32 %0, %5\n\tds_read_b32 %1, %5 offset:16\n\tds_read_b32 %2, %5 offset:32\n\tds_read_b32 %3, %5 offset:48\n\t"
                             "ds_read_b128 %4, %6"
                             : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]) , "=&v"(bf) : "v"(pa), "v"(pb) : "memory");
                pa += 64u; pb += 1024u;
            };
            float ae[4], ao[4];
            f32x4 be, bo;
            issue4(ae, be);
#pragma unroll 1
            for (int g = 0; g < c16; g += 2) {
                issue4(ao, bo);
                landed(ae, be, std::integral_constant<int, 5>{});
                mfma4(ae, be);
                issue4(ae, be);
                landed(ao, bo, std::integral_constant<int, 5>{});
                if (g + 1 < c16) mfma4(ao, bo);
            }
            landed(ae, be, std::integral_constant<int, 0>{});
            total += acc + acc2;
            continue;
        }
        float ae[4], ao[4];
        f32x4 be, bo;
        issue(0, ae, be);
#pragma unroll 1
        for (int g = 0; g < c16; g += 2) {
            issue(g + 1, ao, bo);
            landed(ae, be, std::integral_constant<int, 5>{});
            mfma4(ae, be);
            issue(g + 2, ae, be);
            landed(ao, bo, std::integral_constant<int, 5>{});
            if (g + 1 < c16) mfma4(ao, bo);
        }
        landed(ae, be, std::integral_constant<int, 0>{});
        total += acc + acc2;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = total.x + total.y + total.z + total.w;
    if (blockIdx.x == 0 && lane == 0) clk[w] = t1 - t0;
}

template <int VARIANT>
static void run(const char *name, int waves, int upw) {
    float *out; long long *clk, h[16];
    hipMalloc(&out, 1024 * 4); hipMalloc(&clk, 16 * 8);
    const int c16 = 7, SP = 104, SPq = VARIANT == 4 ? 116 : 108;
    const size_t lds = (64 * SPq + 5 * c16 * 256 + 4096) * 4;
    hipFuncSetAttribute((const void *)k<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<VARIANT><<<1, waves * 64, lds>>>(out, clk, c16, SP, SPq, upw);
    k<VARIANT><<<1, waves * 64, lds>>>(out, clk, c16, SP, SPq, upw);
    hipDeviceSynchronize();
    hipMemcpy(h, clk, 16 * 8, hipMemcpyDeviceToHost);
    long long mx = 0, mn = 1ll << 60;
    for (int i = 0; i < waves; i++) { mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; }
    printf("%-44s waves %2d x %d units of 28 MFMA: %6lld .. %6lld cycles per wavefront = %5.1f cycles per MFMA per SIMD (%d wavefronts per SIMD)\n", name, waves, upw,
           mn, mx, (double)mx / (28.0 * upw * ((waves + 3) / 4)), (waves + 3) / 4);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int waves : {1, 4, 8, 16}) {
        run<0>("as in the kernel (clamped A addresses)", waves, 4);
        run<1>("A addresses unclamped", waves, 4);
        run<2>("two accumulators", waves, 4);
        run<3>("no matrix-core instructions", waves, 4);
        run<4>("one address per operand + offset immediates", waves, 4);
    }
    return 0;
}
