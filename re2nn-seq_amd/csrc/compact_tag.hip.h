// K1t -- the COMPACT tagging step in ONE launch (round 5): bit-packed transition blocks (compact.hip.h), both chains
// of a sequence, its label-map scores and the threshold/argmax decode in one workgroup; nothing but the tags leaves the chip.
//
// Reference: FARNN_S_O_I_S.forward_score / local_decode (model_onehot.py:372-403, :411-426, :162-180) on the 0/1 tensors the
// loader writes (wfa/fsa_to_tensor.py:546-615; main.py:175-176 allows no other for --method onehot).
//
// compact_chain_kernel (round 2) walked a chain with one wavefront and stored a stash row to HBM every step -- 830 cycles per
// forward step with ~1 active state, 1 730 backward (the stores and the bitmap ring share vmcnt) -- and a second launch scored the
// stash: 57.6 us per 256 x 64 batch against 34.9 dense, while moving 18 x fewer bytes.  Here:
//   * one workgroup of CT_WAVES wavefronts per sequence.  Wavefront 0 walks the forward chain, wavefront 1 the backward chain;
//     lane l owns the state entries l and l + 64 (S <= 128).  Both directions' state rows stay in LDS ([L + 1][SP] each: 37 KB at
//     L = 64, S = 71); the only vector-memory traffic of a step is the next bitmap rows (16 bytes per lane and state entry, a ring
//     of CT_PF steps loaded by counted inline-asm loads: chain_regs.hip.h says why).
//   * a step: the ballot of the non-zero state entries (two 64-bit masks), then ONE pass per DISTINCT value among them: the sources
//     that hold v add v * (popcount(T row_j & their mask) + popcount(W row_j & their mask)) to destination j -- mask arithmetic on
//     the bitmap rows, no walk over the sources.  A 0/1 automaton's state entries are small path counts (one to three distinct
//     values per step, whatever the number of active states), and integer values make every grouping of the sum exact:
//     bit-identical to the dense kernels on `none` / `relu`.
//     A lone wavefront pays ~8 cycles per DEPENDENT instruction, so the step is kept short: block addresses from a register window
//     (one v_readlane per step, chain_regs.hip.h), `none` / `relu` without a branch, one pass and no mask bookkeeping when all
//     active sources hold one value.
//   * the other six wavefronts tag the tokens WHILE the chains run, two per scan, from the two LDS histories (label_map.hip.h:
//     S multiply-adds per token): token i needs forward row i + 1 and backward row len - 1 - i, so the middle of the sequence is
//     ready when both chains have passed it and two tokens become ready per step from then on -- the pairs are dealt to the
//     wavefronts from the middle outwards, each waits for the two chains' progress words (LDS, stored behind the row) and then
//     scans.  When the chains end only the outermost pairs are left: ~1.5 k cycles behind the chain instead of a 5 k-cycle
//     scoring phase.  The same wavefronts first write the pad positions' tags (LOCAL mode) and find the sequence's flat offset.
// Applies when the output matrix is a label map (every loader-built i-FST), no priority layer, no score tensor asked for, no
// CRF, S <= 128 and the histories fit the LDS; everything else keeps compact_chain_kernel + the score kernels.
#pragma once
#include "common.hip.h"
#include "compact.hip.h"
#include "score_params.hip.h"
#include "label_map.hip.h"

namespace farnn {

constexpr int CT_WAVES = 8;
constexpr int CT_PF = 4;            // steps of bitmap rows in flight per chain

struct CompactTagLds { int tok, hA, hB, misc, total; };          // offsets in 4-byte words
__host__ __device__ inline CompactTagLds compact_tag_lds(int L, int SP) {
    CompactTagLds l;
    int at = 0;
    l.tok = at; at += (L + 3) & ~3;
    l.hA = at;  at += (L + 1) * SP;
    l.hB = at;  at += (L + 1) * SP;
    l.misc = at; at += 8;
    l.total = at;
    return l;
}

template <int NS> struct CtRow;
template <> struct CtRow<1> { typedef unsigned v __attribute__((ext_vector_type(2))); };
template <> struct CtRow<2> { typedef unsigned v __attribute__((ext_vector_type(4))); };

// NLX: tanh / relu-tanh between the steps (the none / relu instantiation has no branch in the step)
template <int NS, bool NLX>
__global__ void __launch_bounds__(CT_WAVES * 64, 1)
compact_tag_kernel(const CompactParams p, const ScoreParams sp) {
    static_assert(NS == 1 || NS == 2, "one or two 64-bit words per bitmap row (S <= 128)");
    typedef typename CtRow<NS>::v rowv;
    extern __shared__ __align__(16) float ct_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = blockIdx.x;
#if defined(FARNN_PROBES)
    const long long tr_start = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    const int b = p.order ? p.order[slot] : slot;
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, L = p.L;
    const CompactTagLds lds = compact_tag_lds(L, SP);
    unsigned *tok = reinterpret_cast<unsigned *>(ct_smem) + lds.tok;     // [nsteps] byte offset of token q's bitmap block
    float *hA = ct_smem + lds.hA, *hB = ct_smem + lds.hB;
    int *misc = reinterpret_cast<int *>(ct_smem) + lds.misc;          // [0] flat offset, [2] / [3] rows complete (forward / backward)
    const unsigned blk = (unsigned)S * NS * 8u;                       // bytes of a word's bitmap block (V * blk < 2^32: compact_tag_fits)

    // ---- set-up: the tokens (as block offsets), row 0 of both histories -------------------------------------------------
    for (int q = tid; q < nsteps; q += CT_WAVES * 64) tok[q] = (unsigned)clamp_tok(p.x[(long long)b * L + q], p.V) * blk;
    for (int j = tid; j < SP; j += CT_WAVES * 64) {
        hA[j] = j < S ? p.h0[j] : 0.0f;
        hB[j] = j < S ? p.hT[j] : 0.0f;
    }
    if (tid < 8) misc[tid] = 0;
    __syncthreads();
#if defined(FARNN_PROBES)
    const long long tk0 = (long long)__builtin_amdgcn_s_memtime(), tr0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif

    if (w < 2) {
        if (nsteps > 0) {
        // =================================================================================================================
        // a chain: wavefront 0 forward, wavefront 1 backward
        // =================================================================================================================
        const int dir = w;
        __builtin_amdgcn_s_setprio(2);
        const u64 *wb = dir == 0 ? p.wF : p.wB;
        const char *bits = reinterpret_cast<const char *>(dir == 0 ? p.bitsF : p.bitsB);
        float *hist = dir == 0 ? hA : hB;
        float a[NS], ov[NS], cpre[NS], cpost[NS];
        u64 ww[NS][NS];
        unsigned voff[NS];
        int hoff[NS], hstep[NS];                                   // (offsets into ct_smem, not pointers: a select between two
                                                                   //  LDS pointers decays to a generic one -- flat_store, seen in the ISA)
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const int j = lane + 64 * k;
            const bool ok = j < S;
            ov[k] = ok ? (p.o ? p.o[j] : 1.0f) : 0.0f;           // (a lane without a state: factor 0 -- what it hands to the next step is an exact zero)
            cpre[k] = dir == 0 ? ov[k] : (ok ? 1.0f : 0.0f);
            cpost[k] = dir == 0 ? (ok ? 1.0f : 0.0f) : ov[k];
            float v = ok ? hist[j] : 0.0f;
            if (dir == 1) v *= ov[k];                              // backward input is pre-scaled (:393)
            a[k] = v;
#pragma unroll
            for (int s = 0; s < NS; s++) ww[k][s] = ok ? wb[(long long)j * NS + s] : 0ull;
            voff[k] = (unsigned)(ok ? j : S - 1) * (unsigned)(NS * 8);
            // where this lane's new entries go: row t + 1 of the history (the pad columns S .. SP - 1 receive whatever the lane
            // computes: no state of the label map lives there), or a dump word
            hoff[k] = j < SP ? (dir == 0 ? lds.hA : lds.hB) + SP + j : lds.misc + 4 + k;
            hstep[k] = j < SP ? SP : 0;
        }
        int *prog = misc + 2 + dir;
        // block offsets: 64 steps' worth in a register (lane l: step window + l), one v_readlane per step (chain_regs.hip.h)
        unsigned tkw;
#define FARNN_CT_WINDOW(t_)                                                                    \
        do {                                                                                   \
            const int ti_ = min((t_) + lane, nsteps - 1);                                      \
            tkw = tok[dir == 0 ? ti_ : (ti_ < len ? len - 1 - ti_ : ti_)];                     \
        } while (0)
        // the ring: CT_PF steps x NS rows per lane, loaded by inline asm behind ONE counted wait statement per step
        rowv r[CT_PF][NS];
#define FARNN_CT_ISSUE(u_, off_)                                                               \
        do {                                                                                   \
            const char *bp_ = bits + (off_);                                                   \
            if constexpr (NS == 2)                                                             \
                asm volatile("s_nop 4\n\t"                                                     \
                             "global_load_dwordx4 %0, %2, %4\n\t"                              \
                             "global_load_dwordx4 %1, %3, %4"                                  \
                             : "=&v"(r[u_][0]), "=&v"(r[u_][NS - 1]) : "v"(voff[0]), "v"(voff[NS - 1]), "s"(bp_)); \
            else                                                                               \
                asm volatile("s_nop 4\n\t"                                                     \
                             "global_load_dwordx2 %0, %1, %2"                                  \
                             : "=&v"(r[u_][0]) : "v"(voff[0]), "s"(bp_));                      \
        } while (0)
#define FARNN_CT_WAITSTR                                                                       \
        "s_cmp_ge_i32 %[rem], %[dm1]\n\t"                                                      \
        "s_cbranch_scc1 1f\n\t"                                                                \
        "s_waitcnt vmcnt(0)\n\t"                                                               \
        "s_branch 2f\n"                                                                        \
        "1:\n\t"                                                                               \
        "s_waitcnt vmcnt(%[cnt])\n"                                                            \
        "2:"
#define FARNN_CT_WAIT(u_, rem_)                                                                \
        do {                                                                                   \
            if constexpr (NS == 2)                                                             \
                asm volatile(FARNN_CT_WAITSTR : "+v"(r[u_][0]), "+v"(r[u_][NS - 1])            \
                             : [rem] "s"(rem_), [dm1] "n"(CT_PF - 1), [cnt] "n"((CT_PF - 1) * NS) : "scc"); \
            else                                                                               \
                asm volatile(FARNN_CT_WAITSTR : "+v"(r[u_][0])                                 \
                             : [rem] "s"(rem_), [dm1] "n"(CT_PF - 1), [cnt] "n"((CT_PF - 1) * NS) : "scc"); \
        } while (0)
        FARNN_CT_WINDOW(0);
#pragma unroll
        for (int u = 0; u < CT_PF; u++) {
#pragma unroll
            for (int k = 0; k < NS; k++) r[u][k] = rowv(0u);
            if (u < nsteps) FARNN_CT_ISSUE(u, (unsigned)__builtin_amdgcn_readlane((int)tkw, u));
        }
        const int nl_mode = p.nl;
        const bool nl_relu = nl_mode == FARNN_NL_RELU;
#if defined(FARNN_PROBES)
        long long ph[4] = {0, 0, 0, 0}, pt = tk0;
        int npass = 0;
        pt = (long long)__builtin_amdgcn_s_memtime();
        const long long tb0 = pt;
#define FARNN_CT_PHASE(i) do { if (p.dbg & 8192) { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); ph[i] += n_ - pt; pt = n_; } } while (0)
#else
#define FARNN_CT_PHASE(i) do { } while (0)
#endif
        static_assert(CT_PF == 4, "the window reload below assumes a ring of four steps");
        for (int t0 = 0; t0 < nsteps; t0 += CT_PF) {
#pragma unroll
            for (int u = 0; u < CT_PF; u++) {
                const int t = t0 + u;
                if (t >= nsteps) break;
                // the active sources (wave-uniform masks), the first one's value, whether every active source holds it
                u64 rem[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) rem[s] = __ballot(a[s] != 0.0f);
                // the first active source's value (word 0 first; no active source at all: some lane's value, multiplied by a count of 0)
                float v1;
                {
                    const float va = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[0]), (int)__builtin_ctzll(rem[0] | (1ull << 63))));
                    float vb = va;
                    if constexpr (NS == 2) vb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[NS - 1]), (int)__builtin_ctzll(rem[NS - 1] | (1ull << 63))));
                    v1 = rem[0] ? va : vb;
                }
                bool uniform = true;
#pragma unroll
                for (int s = 0; s < NS; s++) uniform = uniform && __ballot(a[s] == v1) == rem[s];
                FARNN_CT_PHASE(0);
                FARNN_CT_WAIT(u, nsteps - 1 - t);
                FARNN_CT_PHASE(1);
                u64 cur[NS][NS];
#pragma unroll
                for (int k = 0; k < NS; k++)
#pragma unroll
                    for (int s = 0; s < NS; s++)
                        cur[k][s] = (u64)r[u][k][2 * s] | ((u64)r[u][k][2 * s + 1] << 32);
                float acc[NS];
#pragma unroll
                for (int k = 0; k < NS; k++) acc[k] = 0.0f;
                // The sources that hold the value v add v * (their edges into destination j, T and W counted apart) to it: one pass per
                // DISTINCT value among the active sources.  Integer values make any grouping of the sum exact.  The common case -- every
                // active source holds the first one's value -- is straight-line code: no loop, no mask bookkeeping.
                if (uniform) {
#pragma unroll
                    for (int k = 0; k < NS; k++) {
                        int cnt = 0;
#pragma unroll
                        for (int s = 0; s < NS; s++) cnt += __popcll(cur[k][s] & rem[s]) + __popcll(ww[k][s] & rem[s]);
                        acc[k] = v1 * (float)cnt;
                    }
#if defined(FARNN_PROBES)
                    npass++;
#endif
                } else
                for (;;) {
                    int s0 = -1;
#pragma unroll
                    for (int s = NS - 1; s >= 0; s--) if (rem[s]) s0 = s;
                    if (s0 < 0) break;
#if defined(FARNN_PROBES)
                    npass++;
#endif
                    u64 first = 0ull;
                    float v = 0.0f;
#pragma unroll
                    for (int s = 0; s < NS; s++)
                        if (s == s0) {
                            const int i = (int)__builtin_ctzll(rem[s]);
                            v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[s]), i));
                            first = 1ull << i;
                        }
                    u64 mv[NS];
#pragma unroll
                    for (int s = 0; s < NS; s++) {
                        mv[s] = __ballot(a[s] == v) & rem[s];
                        if (s == s0) mv[s] |= first;               // (a NaN equals nothing: its own bit still leaves the set)
                        rem[s] &= ~mv[s];
                    }
#pragma unroll
                    for (int k = 0; k < NS; k++) {
                        int cnt = 0;
#pragma unroll
                        for (int s = 0; s < NS; s++) cnt += __popcll(cur[k][s] & mv[s]) + __popcll(ww[k][s] & mv[s]);
                        acc[k] = fmaf(v, (float)cnt, acc[k]);
                    }
                }
                FARNN_CT_PHASE(2);
                // the block of step t + CT_PF into the slot that has just been read (its offset: lane (t + CT_PF) & 63 of the window,
                // which is reloaded when that step opens the next 64 -- only u = 0 can reach a multiple of 64)
                if (u == 0 && ((t + CT_PF) & 63) == 0 && t + CT_PF < nsteps) FARNN_CT_WINDOW(t + CT_PF);
                if (t + CT_PF < nsteps) FARNN_CT_ISSUE(u, (unsigned)__builtin_amdgcn_readlane((int)tkw, (t + CT_PF) & 63));
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    // (:377-386) / (:393-402): the forward chain scales by o before the non-linearity, the backward chain after it -- one
                    // multiply each by a per-lane constant (o, 1 or -- a lane without a state -- 0) instead of a branch on the direction
                    const float x = acc[k] * cpre[k];
                    const float hn = NLX ? cc_nl(x, nl_mode) : (nl_relu ? fmaxf(x, 0.0f) : x);
                    const float hnext = hn * cpost[k];
                    ct_smem[hoff[k]] = hn;
                    hoff[k] += hstep[k];
                    a[k] = hnext;
                }
                // rows 0 .. t + 1 of this direction are complete: the progress word, behind the row in this wavefront's LDS order
                if (lane == 0) __hip_atomic_store(prog, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                FARNN_CT_PHASE(3);
            }
        }
#if defined(FARNN_PROBES)
        if ((p.dbg & 4096) && nsteps == L && lane == 0)
            printf("compact tag seq %d dir %d: %d steps, %lld cycles per step\n", b, dir, nsteps, (long long)(__builtin_amdgcn_s_memtime() - tb0) / nsteps);
        if ((p.dbg & 8192) && nsteps == L && lane == 0)
            printf("compact tag seq %d dir %d phases, cycles per step: masks %lld, block wait %lld, value passes %lld (%d.%d per step), issue + epilogue %lld\n",
                   b, dir, ph[0] / nsteps, ph[1] / nsteps, ph[2] / nsteps, npass / nsteps, (10 * npass / nsteps) % 10, ph[3] / nsteps);
#endif
#undef FARNN_CT_PHASE
#undef FARNN_CT_WAIT
#undef FARNN_CT_WAITSTR
#undef FARNN_CT_ISSUE
#undef FARNN_CT_WINDOW
        __builtin_amdgcn_s_setprio(0);
        }
    } else {
        // =====================================================================================================================
        // the six tagging wavefronts
        // =====================================================================================================================
        constexpr int NTW = CT_WAVES - 2;
        const int tw = w - 2;
        long long foff = 0;
        if (sp.flat) {
            // the flat-output offset of the sequence (utils.py:153-164): the sum of the lengths in front of it (every wavefront its own
            // copy: ~b / 64 loads, under the chains' first steps)
            int partsum = 0;
            if (sp.offs) partsum = lane == 0 ? (int)sp.offs[b] : 0;
            else for (int j = lane; j < b; j += WAVE) partsum += clamp_len(sp.len[j], L);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) partsum += __shfl_xor(partsum, off, WAVE);
            foff = partsum;
        }
        if (tw == 0 && sp.tags)                                       // pad positions of LOCAL mode: tag -1
            for (int i = nsteps + lane; i < L; i += WAVE) sp.tags[(long long)b * L + i] = -1;
        LabelMapRegs lr;
        lm_load(sp.lm, lane, lr);
        const bool two = sp.lm.nq > 1;                              // (second register unused: products selected to zero, not multiplied)
        const int NP = (nsteps + 1) >> 1;                             // token pairs (2 p, 2 p + 1)
        const int mid = min(max((len >> 1) >> 1, 0), max(NP - 1, 0));
        const int *progA = misc + 2, *progB = misc + 3;
        auto rowB = [&](int i) { return (i + 1 <= len) ? len - (i + 1) : i + 1; };   // beta of token i (pads of FULL mode: row i + 1)
        for (int q = tw; q < 2 * NP + 2; q += NTW) {
            const int pp = mid + ((q & 1) ? -((q + 1) >> 1) : (q >> 1));          // from the middle outwards: the order the pairs become ready in
            if (pp < 0 || pp >= NP) continue;
            const int ia = 2 * pp;
            const bool hasb = ia + 1 < nsteps;
            const int ib = hasb ? ia + 1 : ia;
            const int needA = ib + 1, needB = max(rowB(ia), rowB(ib));
            for (;;) {
                const int pa = __hip_atomic_load(progA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int pb = __hip_atomic_load(progB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
                if (pa >= needA && pb >= needB) break;
                __builtin_amdgcn_s_sleep(8);
            }
            const float *fa = hA + (ia + 1) * SP, *fb = hA + (ib + 1) * SP;
            const float *ba = hB + rowB(ia) * SP, *bb = hB + rowB(ib) * SP;
            const float xa0 = fa[lr.st0] * ba[lr.st0], xa1 = two ? fa[lr.st1] * ba[lr.st1] : 0.0f;
            const float xb0 = fb[lr.st0] * bb[lr.st0], xb1 = two ? fb[lr.st1] * bb[lr.st1] : 0.0f;
            float ya0, ya1, yb0, yb1;
            lm_scan_scores2(lr, xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1);
            float ma = fmaxf(ya0, ya1), mb = fmaxf(yb0, yb1);
            wave_max_dpp2(ma, mb);
            const int taga = lm_tag_from_candidates(sp.lm, lr, ya0, ya1, ma, sp.K, sp.o_idx);
            const int tagb = lm_tag_from_candidates(sp.lm, lr, yb0, yb1, mb, sp.K, sp.o_idx);
            if (lane < (hasb ? 2 : 1)) {
                const int i = lane ? ib : ia;
                const int tag = lane ? tagb : taga;
                if (sp.tags) sp.tags[(long long)b * L + i] = tag;
                if (sp.flat && i < len) sp.flat[foff + i] = tag;
            }
        }
#if defined(FARNN_PROBES)
        if ((p.dbg & 4096) && nsteps == L && tw == 0 && lane == 0)
            printf("compact tag seq %d: tagging wavefront 0 done %lld cycles = %lld ns after the set-up, set-up %lld ns\n", b, (long long)__builtin_amdgcn_s_memtime() - tk0,
                   10 * ((long long)__builtin_amdgcn_s_memrealtime() - tr0), 10 * (tr0 - tr_start));
#endif
    }
}

// the geometries the one-launch form covers
inline bool compact_tag_fits(int V, int S, int SP, int L) {
    return S <= 128 && (size_t)compact_tag_lds(L, SP).total * 4 <= (size_t)150 * 1024 &&
           (unsigned long long)V * S * (S <= 64 ? 1 : 2) * 8ull < (1ull << 32);
}

}  // namespace farnn
