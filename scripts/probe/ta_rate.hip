// What the vector-memory path of ONE compute unit delivers to registers when every request hits its L1: cycles per
// global_load instruction by width (dword, x2, x3, x4) and by the set of active lanes (all 64; the first 48 / 36 / 32 / 16: whole
// quads switched off; every other lane; three of every four) -- is the cost of a load its instruction, its active lanes or its
// active QUADS?  The dense recurrence kernel's step (K1d) is bound by this path (scripts/debug/pool_probe.py: the step does not care
// whether its block sits in the L1, the L2 of the XCD or the other XCDs' -- only beyond the L2s it slows down).
//
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe/ta_rate.hip -o /tmp/ta_rate && /tmp/ta_rate
//
// One workgroup of `waves` wavefronts per compute unit; every wavefront issues NL loads per round from a 4 KiB (per wavefront)
// region it has touched before, waits for all of them, NR rounds; cycles by s_memtime over the whole loop of wavefront 0 after a
// workgroup barrier (all wavefronts run the same loop: the compute unit's path is shared).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int NL = 16, NR = 64;

template <int W>
__global__ void __launch_bounds__(512) probe(const char *base, long long *out, unsigned long long mask, int stride) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // lane l reads `W` dwords at offset l * stride of its wavefront's 4 KiB (stride 16: the lanes tile 1 KiB contiguously)
    const char *p = base + ((size_t)blockIdx.x * 8 + w) * 4096;
    const unsigned voff = (unsigned)lane * (unsigned)stride;
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    v4 acc = {0, 0, 0, 0};
    // warm the L1
    for (int i = 0; i < 4; i++) acc += *reinterpret_cast<const v4 *>(p + (size_t)lane * 16 + 1024 * i);
    __syncthreads();
    const unsigned long long act = mask;
    long long t0 = 0, t1 = 0;
    if (w == 0) t0 = (long long)__builtin_amdgcn_s_memtime();
    if ((act >> lane) & 1ull) {
        for (int r = 0; r < NR; r++) {
            v4 d[NL];
#pragma unroll
            for (int i = 0; i < NL; i++) {
                const unsigned o = voff + 1024u * (i & 3);
                if constexpr (W == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d[i]) : "v"(o), "s"(p));
                else if constexpr (W == 3) { typedef unsigned v3 __attribute__((ext_vector_type(3))); v3 t; asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(t) : "v"(o), "s"(p)); d[i] = v4{t.x, t.y, t.z, 0}; }
                else if constexpr (W == 2) { typedef unsigned v2 __attribute__((ext_vector_type(2))); v2 t; asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(t) : "v"(o), "s"(p)); d[i] = v4{t.x, t.y, 0, 0}; }
                else { unsigned t; asm volatile("global_load_dword %0, %1, %2" : "=v"(t) : "v"(o), "s"(p)); d[i] = v4{t, 0, 0, 0}; }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NL; i++) acc += d[i];
        }
    }
    __syncthreads();
    if (w == 0) t1 = (long long)__builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc.x + acc.y + acc.z + acc.w == 0x12345u) out[blockIdx.x] = 0;
}

int main() {
    const int nb = 256;
    char *buf; long long *d;
    if (hipMalloc(&buf, (size_t)nb * 8 * 4096 + 8192) != hipSuccess || hipMalloc(&d, nb * sizeof(long long)) != hipSuccess) { printf("no device\n"); return 1; }
    (void)hipMemset(buf, 1, (size_t)nb * 8 * 4096 + 8192);
    std::vector<long long> h(nb);
    struct M { const char *name; unsigned long long m; } masks[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-47 (12 quads)", (1ull << 48) - 1}, {"lanes 0-35 (9 quads)", (1ull << 36) - 1},
        {"lanes 0-31 (8 quads)", (1ull << 32) - 1}, {"lanes 0-15 (4 quads)", (1ull << 16) - 1},
        {"every other lane (32 lanes, 16 quads)", 0x5555555555555555ull}, {"three of four lanes (48 lanes, 16 quads)", 0x7777777777777777ull},
        {"one of four lanes (16 lanes, 16 quads)", 0x1111111111111111ull}};
    printf("cycles per load instruction and wavefront-load bytes per clock of one compute unit (L1 hits; median over 256 compute units)\n");
    for (int waves : {8, 2}) {
        for (int W : {4, 3, 2, 1}) {
            for (auto &mk : masks) {
                for (int stride : {16}) {
                    if (W == 4) probe<4><<<nb, 64 * waves>>>(buf, d, mk.m, stride);
                    else if (W == 3) probe<3><<<nb, 64 * waves>>>(buf, d, mk.m, stride);
                    else if (W == 2) probe<2><<<nb, 64 * waves>>>(buf, d, mk.m, stride);
                    else probe<1><<<nb, 64 * waves>>>(buf, d, mk.m, stride);
                    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                    (void)hipMemcpy(h.data(), d, nb * sizeof(long long), hipMemcpyDeviceToHost);
                    std::sort(h.begin(), h.end());
                    const double cyc = (double)h[nb / 2];
                    const double ninstr = (double)waves * NL * NR;
                    const int lanes = __builtin_popcountll(mk.m);
                    printf("%d waves  %-12s %-42s %7.1f cycles per instruction  %6.1f B/clk\n", waves,
                           W == 4 ? "dwordx4" : W == 3 ? "dwordx3" : W == 2 ? "dwordx2" : "dword", mk.name, cyc / ninstr, ninstr * lanes * W * 4 / cyc);
                }
            }
        }
    }
    return 0;
}
