// K1c -- the onehot i-FST recurrence on a COMPACT automaton: bit-packed transition blocks and a walk over the ACTIVE
// states only (SURVEY.md 8f2 / VERDICT r1 "missing #1").
//
// Reference: FARNN_S_O_I_S.forward_score time loop (model_onehot.py:372-403), the same recurrence as K1 (chain.hip.h):
//     fwd:  a[k+1] = nl( (a[k] . (T[x_k] + W)) * o )            bwd:  b[k+1] = nl( (T[x'_k] + W) . (b[k] * o) )
// The reference's loaders write T and W as dense float64 tensors (wfa/fsa_to_tensor.py:546-615), but with
// --rand_constant 0 (the only setting main.py allows for --method onehot, :175-176) every entry is 0 or 1 and a word's
// S x S block holds a handful of edges (~0.2 % dense at ATIS size).  K1 streams the dense fp32 blocks: 2 S^2 4 bytes per
// token (40 KB at S = 71, 2 MB at S = 512).  Here a block is S rows of S BITS:
//     bitsF[w][j] = { i : T[w][i][j] = 1 }   (sources of destination j: the forward chain's gather)
//     bitsB[w][i] = { j : T[w][i][j] = 1 }   (destinations of source i: the backward chain's gather)
// 2 S NS 8 bytes per token (NS = ceil(S/64) 64-bit words per row: 2.3 KB at S = 71, 64 KB at S = 512; the whole 21 GB
// tensor of BASELINE's largest config becomes 0.66 GB), plus the two bitmaps of W, which are the same for every token
// and live in registers.  A state vector of an automaton has few non-zero entries, so a step does not visit S sources
// per destination: the wavefront ballots its non-zero state entries and walks THAT set; for each active source i every
// lane tests bit i of its destinations' rows and accumulates a[i] * (T bit + W bit) -- the same fmaf(a[i], T+W, acc) the
// dense kernel performs, over the non-zero terms only (a zero term adds +0.0).  Integer-valued states (0/1 automata
// with `none` / `relu`) make every order of the additions exact: bit-identical to K1.  With tanh the sums differ from
// K1's by rounding order (within the 1e-4 bar).
//
// One wavefront per (sequence, direction); lane l owns the state entries l, l+64, ... (NS of them).  No barrier:
// tokens sit in LDS (read with a wave-uniform address), the active sources are broadcast with v_readlane, the next
// step's bitmap rows are fetched one step ahead.  Bound: the serial step chain (one L2 round trip is hidden by the
// prefetch; ~25 instructions per active source).  The stash it writes is the dense kernel's, so scores / decode follow
// unchanged (score_tile_kernel / Viterbi).  Contract numbers stay fp32-dense; bench.py reports this
// path separately as `compact`.
#pragma once
#include "common.hip.h"

namespace farnn {

typedef unsigned long long u64;

struct CompactParams {
    const u64 *bitsF, *bitsB;     // [V][S][NS]
    const u64 *wF, *wB;           // [S][NS]  bitmaps of W (sources of j / destinations of i)
    const u64 *mF, *mB;           // [V][S][NS] T | W merged (compact_tag.hip.h; S <= 128), 4 KiB of slack behind the last block
    const u64 *xF, *xB;           // [V][S][NS] T & W: the edges a word's block and the wildcard block share (almost always none)
    const unsigned *tokoff;       // [V] byte offset of word v's merged block | 1 when its T & W plane is not empty
    const float *o;               // [SP] output-sum vector or nullptr
    const float *h0, *hT;         // [S]
    const int64_t *x, *len;
    const int *order;             // launch order (batch_prep) or nullptr
    float *A, *Bk;                // stash [B][L+1][SP]
    int B, L, S, SP, V, nl, full;
    int dbg;                      // FARNN_DBG & 4096: chain 0 / 1 print their cycle counts (diagnostic)
};

constexpr int CC_WAVES = 4;       // chains per workgroup

// update non-linearity on the hardware exponential and reciprocal (|error| ~2e-7 against the 1e-4 bar of the float paths;
// `none` / `relu` stay exact)
__device__ __forceinline__ float cc_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));
    const float big = copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
    return fabsf(x) < TANH_SERIES_BELOW ? tanh_series(x) : big;    // (common.hip.h: 1 - e cancels for small |x|)
}
__device__ __forceinline__ float cc_nl(float v, int nl) {
    switch (nl) {
        case FARNN_NL_RELU:     return fmaxf(v, 0.0f);
        case FARNN_NL_TANH:     return cc_tanh(v);
        case FARNN_NL_RELUTANH: return cc_tanh(fmaxf(v, 0.0f));
        default:                return v;
    }
}

template <int NS>
__global__ void __launch_bounds__(CC_WAVES * 64, 1)
compact_chain_kernel(const CompactParams p) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * CC_WAVES + w;
    if (item >= 2 * p.B) return;
    const int dir = item & 1;
    const int b = p.order ? p.order[item >> 1] : (item >> 1);
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP;
    const u64 *bits = dir == 0 ? p.bitsF : p.bitsB;
    const u64 *wb = dir == 0 ? p.wF : p.wB;
    const float *hinit = dir == 0 ? p.h0 : p.hT;
    float *stash = (dir == 0 ? p.A : p.Bk) + (long long)b * (p.L + 1) * SP;

    // this lane's state entries j_k = lane + 64 k, their output-sum factors and their rows of the wildcard bitmap
    float a[NS], ov[NS];
    u64 ww[NS][NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {
        const int j = lane + 64 * k;
        const bool ok = j < S;
        ov[k] = (ok && p.o) ? p.o[j] : 1.0f;
        float v = ok ? hinit[j] : 0.0f;
        if (j < SP) stash[j] = v;                              // state 0 (pad columns zero)
        if (dir == 1) v *= ov[k];                              // backward input is pre-scaled (:393)
        a[k] = v;
#pragma unroll
        for (int s = 0; s < NS; s++) ww[k][s] = ok ? wb[(long long)j * NS + s] : 0ull;
    }
    if (nsteps <= 0) return;
    // tokens in consumption order: this wavefront's slice of the (only) LDS array; read back with a wave-uniform address
    extern __shared__ int cc_tok[];
    int *tok = cc_tok + w * p.L;
    for (int q = lane; q < nsteps; q += WAVE) {
        const int idx = (dir == 0) ? q : (q < len ? len - 1 - q : q);
        tok[q] = clamp_tok(p.x[(long long)b * p.L + idx], p.V);
    }
    __builtin_amdgcn_wave_barrier();
    auto token = [&](int t) -> int { return __builtin_amdgcn_readfirstlane(tok[t]); };
    auto fetch = [&](int tk, u64 (&dst)[NS][NS]) {
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const int j = lane + 64 * k;
            const u64 *row = bits + ((long long)tk * S + (j < S ? j : S - 1)) * NS;
#pragma unroll
            for (int s = 0; s < NS; s++) dst[k][s] = row[s];
        }
    };
    // The bitmap rows of the next PF steps are in flight (a ring of register sets, the step loop unrolled by PF so that the
    // ring is indexed statically): a lone wavefront's step is shorter than an L2 round trip, so one step of lookahead
    // left most of that latency exposed (measured at ATIS size: 1.0 us per step with one stage).
    constexpr int PF = NS <= 2 ? 4 : (NS == 4 ? 2 : 1);
    u64 ring[PF][NS][NS];
#pragma unroll
    for (int u = 0; u < PF; u++) fetch(token(u < nsteps ? u : nsteps - 1), ring[u]);
    const int nl_mode = p.nl;
    auto step = [&](int t, const u64 (&cur)[NS][NS]) {
        float acc[NS];
#pragma unroll
        for (int k = 0; k < NS; k++) acc[k] = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            u64 m = __ballot(a[s] != 0.0f);                    // the active sources 64 s .. 64 s + 63
            while (m) {                                        // (uniform loop)
                const int i = __builtin_ctzll(m);
                m &= m - 1;
                const float ai = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[s]), i));
                const unsigned sh = (unsigned)i & 31u;
                // fmaf(a[i], T + W, acc): the dense kernel's term; the count T bit + W bit from two 32-bit bit-field
                // extracts of the half the source lives in (uniform branch; 64-bit variable shifts run at quarter rate)
                if (i < 32) {
#pragma unroll
                    for (int k = 0; k < NS; k++) {
                        const unsigned cnt = __builtin_amdgcn_ubfe((unsigned)cur[k][s], sh, 1u) + __builtin_amdgcn_ubfe((unsigned)ww[k][s], sh, 1u);
                        acc[k] = fmaf(ai, (float)cnt, acc[k]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NS; k++) {
                        const unsigned cnt = __builtin_amdgcn_ubfe((unsigned)(cur[k][s] >> 32), sh, 1u) +
                                             __builtin_amdgcn_ubfe((unsigned)(ww[k][s] >> 32), sh, 1u);
                        acc[k] = fmaf(ai, (float)cnt, acc[k]);
                    }
                }
            }
        }
        float *srow = stash + (long long)(t + 1) * SP;
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const int j = lane + 64 * k;
            float hn, hnext;
            if (dir == 0) { hn = cc_nl(acc[k] * ov[k], nl_mode); hnext = hn; }      // (:377-386)
            else          { hn = cc_nl(acc[k], nl_mode);         hnext = hn * ov[k]; } // (:393-402)
            if (j >= S) { hn = 0.0f; hnext = 0.0f; }
            if (j < SP) srow[j] = hn;
            a[k] = hnext;
        }
    };
    const long long tb0 = FARNN_PROBE_ON(p.dbg & 4096) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    // main part: whole groups of PF steps with NO branch around a fetch (a conditional fetch makes the compiler merge the
    // loaded registers with moves right behind the load, i.e. wait for it at once); the look-ahead index is clamped instead
    int t0 = 0;
    for (; t0 + PF <= nsteps; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int t = t0 + u;
            step(t, ring[u]);
            const int tn = t + PF < nsteps ? t + PF : nsteps - 1;
            fetch(token(tn), ring[u]);
        }
    }
    // tail (fewer than PF steps): their rows are already in the ring
#pragma unroll
    for (int u = 0; u < PF; u++)
        if (t0 + u < nsteps) step(t0 + u, ring[u]);
    if (FARNN_PROBE_ON(p.dbg & 4096) && item < 2 && lane == 0)
        printf("compact chain %d (dir %d): %d steps, %lld cycles per step\n", item, dir, nsteps,
               (long long)(__builtin_amdgcn_s_memtime() - tb0) / nsteps);
}

// ---- building the bitmaps (create time) -----------------------------------------------------------------------------
// from the dense tensors the reference's loader writes: T [V][S][S], W [S][S] (device).  Every entry must be 0 or 1;
// `bad` counts the others (the handle then has no compact form).  grid = (V + 1, ceil(S*S/256)): block column V is W
// (the vocabulary rides on grid.x: grid.y stops at 65 535).
__global__ void dense_to_bits_kernel(const float *T, const float *W, u64 *bitsF, u64 *bitsB, u64 *wF, u64 *wB,
                                     int V, int S, int NS, int *bad) {
    const long long v = blockIdx.x;
    const int idx = blockIdx.y * blockDim.x + threadIdx.x;
    if (idx >= S * S) return;
    const int i = idx / S, j = idx - i * S;
    const float val = v < V ? T[v * S * S + idx] : W[idx];
    if (val == 0.0f) return;
    if (val != 1.0f) { atomicAdd(bad, 1); return; }
    u64 *f = v < V ? bitsF + v * S * NS : wF, *bk = v < V ? bitsB + v * S * NS : wB;
    atomicOr(f + (long long)j * NS + (i >> 6), 1ull << (i & 63));      // sources of destination j
    atomicOr(bk + (long long)i * NS + (j >> 6), 1ull << (j & 63));     // destinations of source i
}

// from the automaton's edge list (farnn_edge_list): word >= 0: T bit, word == -1: W bit, word < -1: label only;
// weights other than 1 are counted in `bad`
__global__ void edges_to_bits_kernel(const int32_t *word, const int32_t *from, const int32_t *to, const float *val,
                                     long long n, u64 *bitsF, u64 *bitsB, u64 *wF, u64 *wB, int V, int S, int NS,
                                     int *bad) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int wd = word[e], i = from[e], j = to[e];
    if (wd < -1) return;
    if (wd >= V || i < 0 || i >= S || j < 0 || j >= S || (val && val[e] != 1.0f)) { atomicAdd(bad, 1); return; }
    u64 *f = wd >= 0 ? bitsF + (long long)wd * S * NS : wF, *bk = wd >= 0 ? bitsB + (long long)wd * S * NS : wB;
    atomicOr(f + (long long)j * NS + (i >> 6), 1ull << (i & 63));
    atomicOr(bk + (long long)i * NS + (j >> 6), 1ull << (j & 63));
}

// K1t's form of the blocks: the wildcard bitmap merged into every word's block (one popcount per row word instead of two), the
// shared edges -- entries 2 of T + W -- as a second plane, and the per-word table {block offset | second plane present}.
// One thread per (word, row, 64-bit word of the row).
__global__ void merge_planes_kernel(const u64 *bitsF, const u64 *bitsB, const u64 *wF, const u64 *wB, u64 *mF, u64 *mB, u64 *xF, u64 *xB,
                                    unsigned *tokoff, int V, int S, int NS) {
    const long long n = (long long)V * S * NS;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int rw = (int)(e % ((long long)S * NS));               // (row, word) inside the block
    const int v = (int)(e / ((long long)S * NS));
    const u64 tf = bitsF[e], tb = bitsB[e], wf = wF[rw], wbk = wB[rw];
    mF[e] = tf | wf; mB[e] = tb | wbk;
    xF[e] = tf & wf; xB[e] = tb & wbk;
    if ((tf & wf) | (tb & wbk)) atomicOr(tokoff + v, 1u);       // (tokoff[v] holds the offset already: set by the host)
}

inline int compact_ns(int S) { return S <= 64 ? 1 : S <= 128 ? 2 : S <= 256 ? 4 : S <= 512 ? 8 : 0; }   // 0: not built

inline int launch_compact_chain(const CompactParams &p, int NS, hipStream_t s) {
    const dim3 grid((2 * p.B + CC_WAVES - 1) / CC_WAVES), block(CC_WAVES * 64);
    const size_t lds = (size_t)CC_WAVES * p.L * sizeof(int);          // tokens (L <= 1024: 16 KiB)
    switch (NS) {
        case 1: compact_chain_kernel<1><<<grid, block, lds, s>>>(p); break;
        case 2: compact_chain_kernel<2><<<grid, block, lds, s>>>(p); break;
        case 4: compact_chain_kernel<4><<<grid, block, lds, s>>>(p); break;
        case 8: compact_chain_kernel<8><<<grid, block, lds, s>>>(p); break;
        default: return fail(FARNN_ERANGE, "compact chain: more than 512 states%s%s");
    }
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

}  // namespace farnn
