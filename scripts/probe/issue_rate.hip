// What ONE wavefront pays per instruction when nothing else runs on its SIMD -- the regime of every serial chain in this library
// (K1d's compute wavefronts, K1t's chain wavefronts): cycles per instruction (s_memtime) of fixed inline-asm sequences, by the
// distance between an instruction and the one that consumes its result, for the instruction kinds the chains are made of.
//
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
//
// One workgroup per compute unit; `waves` wavefronts per workgroup: wavefront 0 runs the timed sequence, the others sleep-poll an LDS
// word the way the scoring / tagging wavefronts do (does their presence cost the chain anything?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define R2(x) x x
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
constexpr int NI = 256;     // instructions per timed block (R64 of a 4-instruction group)

enum {
    T_VADD_DEP = 0, T_VADD_D2, T_VADD_D4, T_VADD_D8, T_FMA_DEP, T_AND_BCNT_DEP, T_AND_BCNT_ILP, T_SALU_DEP, T_VCMP_SALU_VALU,
    T_READLANE_ROUND, T_DPP_DEP, T_DPP_D2, T_PKFMA_DEP, T_CVT_MUL_DEP, T_BRANCH_TAKEN, T_LDS_ROUND, T_LDS_WRITE_READ, T_MULLO_DEP, T_MULLO_D4, T_EXP_RCP, T_COUNT
};
static const char *names[T_COUNT] = {
    "v_add_f32, each consumes the previous result",
    "v_add_f32, dependency distance 2",
    "v_add_f32, dependency distance 4",
    "v_add_f32, dependency distance 8",
    "v_fmac_f32, each consumes the previous result",
    "v_and_b32 -> v_bcnt_u32_b32 accumulate, one chain (as the compiler emits the popcounts)",
    "v_and_b32 x4 then v_bcnt x4, four chains interleaved",
    "s_add_u32, each consumes the previous result",
    "v_cmp (ballot) -> s_and_b64 -> v_cndmask round (3 instructions)",
    "s_ff1 -> v_readlane -> v_mov round (3 instructions)",
    "v_add_f32 row_shr:1 DPP, each consumes the previous result",
    "v_add_f32 row_shr:1 DPP, dependency distance 2",
    "v_pk_fma_f32, each consumes the previous result",
    "v_cvt_f32_u32 -> v_mul_f32 round (2 instructions)",
    "s_branch to the next instruction (taken branch)",
    "ds_read_b32 -> address of the next (dependent LDS round trip)",
    "ds_write_b32 + ds_read_b32 of it + use (store / load / consume round)",
    "v_mul_lo_u32, each consumes the previous result",
    "v_mul_lo_u32, dependency distance 4",
    "v_exp_f32 -> v_rcp_f32 round (2 instructions)",
};

__global__ void __launch_bounds__(512) probe(long long *out, int test, int reps) {
    __shared__ int word[64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 64) word[threadIdx.x] = threadIdx.x == 63 ? 0 : (int)(unsigned)(size_t)(void *)word + 4 * ((threadIdx.x + 1) & 15);
    __syncthreads();
    if (w != 0) {             // the bystanders: poll an LDS word with s_sleep between the reads until wavefront 0 is done
        while (__hip_atomic_load(&word[63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(8);
        return;
    }
    float v0 = lane, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f, v8 = 8.f;
    unsigned u0 = lane * 2654435761u, u1 = 0, u2 = 0, u3 = 0, u4 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0;
    unsigned long long m = 0x5555aaaa3333ccccull;
    int sidx = 3;
    long long best = 1ll << 60;
    for (int r = 0; r < reps; r++) {
        __builtin_amdgcn_s_setprio(2);
        const long long t0 = (long long)__builtin_amdgcn_s_memtime();
        asm volatile("s_nop 0" ::: "memory");
        switch (test) {
        case T_VADD_DEP:
            asm volatile(R64("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n") : "+v"(v0) : "v"(v1)); break;
        case T_VADD_D2:
            asm volatile(R64("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n") : "+v"(v0), "+v"(v2) : "v"(v1)); break;
        case T_VADD_D4:
            asm volatile(R64("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") : "+v"(v0), "+v"(v2), "+v"(v3), "+v"(v4) : "v"(v1)); break;
        case T_VADD_D8:
            asm volatile(R16(R2("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                                "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"))
                         : "+v"(v0), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8) : "v"(v1));
            break;
        case T_FMA_DEP:
            asm volatile(R64("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n") : "+v"(v0) : "v"(v1), "v"(v2)); break;
        case T_AND_BCNT_DEP:
            asm volatile(R64("v_and_b32 %1, %2, %3\n v_bcnt_u32_b32 %0, %1, %0\n v_and_b32 %1, %2, %4\n v_bcnt_u32_b32 %0, %1, %0\n") : "+v"(u1), "+v"(u2) : "v"(u0), "s"((unsigned)m), "s"((unsigned)(m >> 32))); break;
        case T_AND_BCNT_ILP:
            asm volatile(R16(R2("v_and_b32 %4, %8, %9\n v_and_b32 %5, %8, %10\n v_and_b32 %6, %8, %9\n v_and_b32 %7, %8, %10\n"
                                "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3\n"))
                         : "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(u0), "s"((unsigned)m), "s"((unsigned)(m >> 32)));
            u1 += t4 + t5 + t6 + t7;
            break;
        case T_SALU_DEP:
            asm volatile(R64("s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n") : "+s"(sidx) :: "scc"); break;
        case T_VCMP_SALU_VALU:
            asm volatile(R64("v_cmp_neq_f32 vcc, 0, %0\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %1, %0, vcc\n s_nop 0\n") : "+v"(v0) : "v"(v1) : "vcc"); break;
        case T_READLANE_ROUND:
            asm volatile(R64("s_ff1_i32_b64 s20, %2\n v_readlane_b32 s21, %0, s20\n v_mov_b32 %0, s21\n s_nop 0\n") : "+v"(v0), "+s"(sidx) : "s"(m) : "s20", "s21"); break;
        case T_DPP_DEP:
            asm volatile(R64("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(v0)); break;
        case T_DPP_D2:
            asm volatile(R64("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(v0), "+v"(v2)); break;
        case T_PKFMA_DEP: {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a = {v0, v1}, b = {v2, v3};
            asm volatile(R64("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n") : "+v"(a) : "v"(b));
            v0 = a.x + a.y; break; }
        case T_CVT_MUL_DEP:
            asm volatile(R64("v_cvt_f32_u32 %0, %1\n v_mul_f32 %0, %0, %2\n v_cvt_u32_f32 %1, %0\n v_and_b32 %1, 7, %1\n") : "+v"(v0), "+v"(u1) : "v"(v1)); break;
        case T_BRANCH_TAKEN:
            asm volatile(R64("s_branch 1f\n1:\n s_branch 2f\n2:\n s_branch 3f\n3:\n s_branch 4f\n4:\n") ::: "memory"); break;
        case T_LDS_ROUND: {
            unsigned a = (unsigned)(size_t)(void *)word;
            asm volatile(R64("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a) :: "memory");
            u1 = a; break; }
        case T_MULLO_DEP:
            asm volatile(R64("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n") : "+v"(u1) : "v"(u0)); break;
        case T_MULLO_D4:
            asm volatile(R64("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n") : "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4) : "v"(u0)); break;
        case T_EXP_RCP:
            asm volatile(R64("v_exp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_exp_f32 %0, %0\n v_rcp_f32 %0, %0\n") : "+v"(v0)); break;
        case T_LDS_WRITE_READ: {
            unsigned a = (unsigned)(size_t)(void *)word + 4 * (lane & 31) + 64;
            asm volatile(R64("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %2\n") : "+v"(v0) : "v"(a), "v"(v1) : "memory");
            break; }
        }
        asm volatile("s_nop 0" ::: "memory");
        const long long t1 = (long long)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_setprio(0);
        best = t1 - t0 < best ? t1 - t0 : best;
    }
    if (lane == 0) {
        out[blockIdx.x] = best;
        __hip_atomic_store(&word[63], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // keep the results alive
    if (v0 + v2 + v3 + v4 + v5 + v6 + v7 + v8 == 12345.678f && u1 + u2 + u3 + u4 + sidx == 77) out[blockIdx.x] = 0;
}

int main() {
    long long *d;
    const int nb = 256;
    if (hipMalloc(&d, nb * sizeof(long long)) != hipSuccess) { printf("no device\n"); return 1; }
    std::vector<long long> h(nb);
    // the empty region's cost (two s_memtime + the s_nops) is subtracted
    printf("%-92s %10s %10s %10s\n", "cycles per instruction (lone wavefront; min over 20 runs, median over 256 compute units)", "1 wave", "4 waves", "8 waves");
    for (int t = 0; t < T_COUNT; t++) {
        double res[3];
        int k = 0;
        for (int waves : {1, 4, 8}) {
            probe<<<nb, 64 * waves>>>(d, t, 20);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            hipMemcpy(h.data(), d, nb * sizeof(long long), hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            res[k++] = (double)h[nb / 2] / NI;
        }
        printf("%-92s %10.2f %10.2f %10.2f\n", names[t], res[0], res[1], res[2]);
    }
    return 0;
}
