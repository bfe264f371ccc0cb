// Shared helpers for the FA-RNN tagging kernels (gfx950 / CDNA4 only).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
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

// In-kernel phase probes (s_memtime stamps, device printf) exist in the profiling build only (build.py --probes: -DFARNN_PROBES);
// in the production library the predicate is a compile-time false and the probe code is gone.
#if defined(FARNN_PROBES)
#define FARNN_PROBE_ON(expr) (expr)
#else
#define FARNN_PROBE_ON(expr) false
#endif

// ---- device helpers ---------------------------------------------------------------------------
// tanh of a SMALL argument by its odd series (|x| < 1/16: x (1 - x^2/3 + 2 x^4/15), exact to 4e-9 relative).  The kernels' fast form
// (1 - e) / (1 + e), e = exp(-2|x|), is good to ~1e-7 ABSOLUTE, but 1 - e cancels as |x| -> 0: 2e-4 relative at |x| = 1e-4.  State
// vectors are mostly small entries, a matrix row sums a hundred of their errors, and a locally sensitive gated recurrence amplified that
// to 8e-4 in one sequence of tests/soak_rows_rounds.py (the float32 oracle with exactly that tanh reproduces the deviation; with the
// series below the threshold -- 1/4, 1/16 or 1/32 alike -- it is back at 3e-7).  With the threshold at 1/16 the combined function is
// within 9e-7 relative everywhere (worst just above it).  Branch-free use: compute both, select on |x| < 1/16 (six instructions).
constexpr float TANH_SERIES_BELOW = 0.0625f;
__device__ __forceinline__ float tanh_series(float x) {
    const float s = x * x;
    return x * fmaf(fmaf(s, 2.0f / 15.0f, -1.0f / 3.0f), s, 1.0f);
}
// The update non-linearity's MODE as two scalars (FARNN_NL_NONE 0, RELU 1, TANH 2, RELUTANH 3: bit 0 = relu first, bit 1 = tanh):
// `lo` = 0 or -inf -- y = max(x, lo) is the relu or nothing, ONE instruction -- and `keep` = all ones or zero -- the result is
// (tanh(y) & keep) | (y & ~keep), one v_bfi_b32.  As selects on the mode's bits the same took a canonicalising max, a max, a select
// and a second select on a 64-bit mask the register-tight kernels re-read from spilled scalars: ten instructions of a step's critical
// tail.  (A NaN input comes out as `lo` -- max returns its number operand; the selects handed it through.  Nothing finite changes.)
struct NlMode { float lo; unsigned keep; };
__device__ __forceinline__ NlMode nl_mode_of(int nl) {
    NlMode m;
    m.lo = (nl & 1) ? 0.0f : -__builtin_inff();
    m.keep = (nl & 2) ? 0xffffffffu : 0u;
    asm volatile("" : "+s"(m.keep));                    // (opaque: seen as 0 / ~0 the compiler turns the insert back into a select on a 64-bit mask)
    return m;
}
__device__ __forceinline__ float nl_floor(float x, NlMode m) {
    float y;
    asm("v_max_f32 %0, %1, %2" : "=v"(y) : "s"(m.lo), "v"(x));
    return y;
}
__device__ __forceinline__ float nl_pick(float th, float y, NlMode m) {
    return __uint_as_float((__float_as_uint(th) & m.keep) | (__float_as_uint(y) & ~m.keep));
}

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

// 16-byte loads through pointers of an EXPLICIT address space: where one code path may read LDS or
// global memory, two loops over typed pointers keep ds_read / global_load; a select between generic
// pointers compiles to flat_load with a full vmcnt+lgkmcnt drain.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef __attribute__((address_space(3))) const v4f lds_cv4f;
typedef __attribute__((address_space(1))) const v4f glb_cv4f;

// LDS reads / counted waits as explicit instructions, for step loops that issue ALL of a phase's reads up front (the forward step
// of viterbi_hist_kernel, the phases of decomp_regs8_kernel): left to the compiler they were issued piecemeal through one or two
// recycled register quads -- an exposed LDS round trip per chunk.  The registers a wait releases are its "+v" operands, so no
// use of them can move in front of it; LDS returns in order, so lgkmcnt(N) = "all but the N youngest have landed".
template <int OFF> __device__ __forceinline__ void lds_read16_at(v4f &d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void lds_wait_for(v4f &d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }
// f(integral_constant<int, I>) for I = I0 .. I1 - 1, unrolled at compile time (asm immediates, register arrays)
template <int I0, int I1, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I0 < I1) { f(std::integral_constant<int, I0>{}); static_for<I0 + 1, I1>(f); }
}

// a sequence length as the kernels use it: the API says 1..L; anything else is clamped, never trusted
__device__ __forceinline__ int clamp_len(long long v, int L) { return v < 0 ? 0 : (v > L ? L : (int)v); }

// a token id as the kernels use it: ids outside [0, V) are treated as the pad word V-1 (zero block), never
// used to index the weights
__device__ __forceinline__ int clamp_tok(long long v, int V) { return (v < 0 || v >= V) ? V - 1 : (int)v; }

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// Agent-scope (write-through / L1-bypassing) 8-byte accesses for data handed from one workgroup to another inside a
// launch: relaxed agent-scope atomics lower to `global_store_dwordx2 ... sc1` / `global_load_dwordx2 ... sc1`
// (MI355X_MICROARCH.md, inter-workgroup visibility).  `p` must be 8-byte aligned.
__device__ __forceinline__ void st2_agent(float *p, float a, float b) {
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld4_agent(const float *p) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                       __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}

// 16-byte agent-scope load in ONE instruction (the atomic form above is two 8-byte loads).  The compiler does not count
// inline-asm loads: the caller issues its loads and then passes every destination through wait_sc1_loads().
__device__ __forceinline__ void ld4_agent_issue(v4f &dst, const float *p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void wait_sc1_loads(v4f &a, v4f &b, v4f &c, v4f &d) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}

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

// One LDS-DMA piece: 64 lanes x 16 bytes from (base + voff) land at LDS byte address `lds`
// (wave-uniform) + lane*16.  M0 carries the LDS base and is compiler-reserved, so it is saved and
// restored inside the statement (cdna_hip_programming.md 5.7).  The leading s_nop covers the
// SGPR-write -> VMEM-read wait states of operands fresh from v_readfirstlane / SALU.
__device__ __forceinline__ void lds_dma16(unsigned voff, const char *base, unsigned lds) {
    unsigned keep;
    asm volatile("s_nop 4\n\t"
                 "s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds)
                 : "memory");
}

// Four consecutive pieces (LDS destinations lds, lds+1K, lds+2K, lds+3K) in one statement: M0 is
// saved/restored once and advanced with s_add between the loads, which trims the scalar overhead
// the loaders pay per piece (the DMA issue itself costs ~60-100 cycles).
__device__ __forceinline__ void lds_dma16x4(unsigned v0, unsigned v1, unsigned v2, unsigned v3,
                                            const char *base, unsigned lds) {
    unsigned keep;
    asm volatile("s_nop 4\n\t"
                 "s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %4, %5\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(base), "s"(lds)
                 : "memory", "scc");
}

// Workgroup barrier that orders LDS traffic only (lgkmcnt), never the vector-memory counter.
__device__ __forceinline__ void wg_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define FARNN_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n < 0 ? 0 : (n > 63 ? 63 : n)) {
        FARNN_VMC(0) FARNN_VMC(1) FARNN_VMC(2) FARNN_VMC(3) FARNN_VMC(4) FARNN_VMC(5) FARNN_VMC(6) FARNN_VMC(7)
        FARNN_VMC(8) FARNN_VMC(9) FARNN_VMC(10) FARNN_VMC(11) FARNN_VMC(12) FARNN_VMC(13) FARNN_VMC(14)
        FARNN_VMC(15) FARNN_VMC(16) FARNN_VMC(17) FARNN_VMC(18) FARNN_VMC(19) FARNN_VMC(20) FARNN_VMC(21)
        FARNN_VMC(22) FARNN_VMC(23) FARNN_VMC(24) FARNN_VMC(25) FARNN_VMC(26) FARNN_VMC(27) FARNN_VMC(28)
        FARNN_VMC(29) FARNN_VMC(30) FARNN_VMC(31) FARNN_VMC(32) FARNN_VMC(33) FARNN_VMC(34) FARNN_VMC(35)
        FARNN_VMC(36) FARNN_VMC(37) FARNN_VMC(38) FARNN_VMC(39) FARNN_VMC(40) FARNN_VMC(41) FARNN_VMC(42)
        FARNN_VMC(43) FARNN_VMC(44) FARNN_VMC(45) FARNN_VMC(46) FARNN_VMC(47) FARNN_VMC(48) FARNN_VMC(49)
        FARNN_VMC(50) FARNN_VMC(51) FARNN_VMC(52) FARNN_VMC(53) FARNN_VMC(54) FARNN_VMC(55) FARNN_VMC(56)
        FARNN_VMC(57) FARNN_VMC(58) FARNN_VMC(59) FARNN_VMC(60) FARNN_VMC(61) FARNN_VMC(62)
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
#undef FARNN_VMC
}


// ---- wave-level first-index argmax on the DPP network (no LDS round trips) ---------------------
// torch.max tie-breaking = (value desc, index asc).  Both are folded into one 64-bit key:
// hi = order-preserving u32 image of the float, lo = ~index (smaller index -> larger key), and the
// key is max-scanned with row_shr 1/2/4/8 + row_bcast 15/31; lane 63 ends up with the wave total.
__device__ __forceinline__ unsigned float_order_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_max_step(unsigned &hi, unsigned &lo) {
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xf, false);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xf, false);
    const bool take = ohi > hi || (ohi == hi && olo > lo);
    hi = take ? ohi : hi;
    lo = take ? olo : lo;
}

// returns the winning index in every lane (wave-uniform); `idx` must be < 0x7fffffff
__device__ __forceinline__ int wave_argmax_dpp(float v, int idx) {
    unsigned hi = float_order_key(v), lo = ~(unsigned)idx;
    dpp_max_step<0x111, 0xf>(hi, lo);      // row_shr:1
    dpp_max_step<0x112, 0xf>(hi, lo);      // row_shr:2
    dpp_max_step<0x114, 0xf>(hi, lo);      // row_shr:4
    dpp_max_step<0x118, 0xf>(hi, lo);      // row_shr:8
    dpp_max_step<0x142, 0xa>(hi, lo);      // row_bcast:15 into rows 1 and 3
    dpp_max_step<0x143, 0xc>(hi, lo);      // row_bcast:31 into rows 2 and 3
    return (int)~(unsigned)__builtin_amdgcn_readlane((int)lo, 63);
}

// wave maximum on the DPP network (wave-uniform result): one v_max with a DPP operand per level -- lanes without a
// source lane (bound_ctrl off) keep their value.  Inline asm: the builtin form costs five instructions per level
// (copy, DPP move, two canonicalising maxes, max).
__device__ __forceinline__ float wave_max_dpp(float v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// wave sum on the DPP network (wave-uniform result), the same ladder as wave_max_dpp
__device__ __forceinline__ float wave_sum_dpp(float v) {
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// first index (torch.max's rule) of the maximum of up to 256 candidates held as v[r] = candidate r*64 + lane (lanes
// beyond the last candidate hold -inf): the wave maximum, then per row of 64 the lanes that equal it -- the lowest set
// bit of the first non-empty row is the answer.  Wave-uniform result; 0 when nothing compares equal (all NaN).
template <int NR>
__device__ __forceinline__ int wave_first_argmax(const float (&v)[NR]) {
    float best = v[0];
#pragma unroll
    for (int r = 1; r < NR; r++) best = fmaxf(best, v[r]);
    const float m = wave_max_dpp(best);
    int idx = 0;
    bool found = false;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const unsigned long long hit = __ballot(v[r] == m);
        if (!found && hit) { idx = r * 64 + __builtin_ctzll(hit); found = true; }
    }
    return idx;
}

// the same answer when the maximum m is already known (bit-identical to one of the candidates): no reduction at all
template <int NR>
__device__ __forceinline__ int wave_first_equal(const float (&v)[NR], const float m) {
    int idx = 0;
    bool found = false;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const unsigned long long hit = __ballot(v[r] == m);
        if (!found && hit) { idx = r * 64 + __builtin_ctzll(hit); found = true; }
    }
    return idx;
}

}  // namespace farnn
