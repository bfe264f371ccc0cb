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
constexpr int CT_PF = 4;            // steps of bitmap rows in flight per chain (eight: no faster at 256 x 64, slower at B = 1024)

struct CompactTagLds { int tok, hA, hB, misc, total, HS; };      // offsets in 4-byte words; HS: floats per history row
__host__ __device__ inline CompactTagLds compact_tag_lds(int L, int NS) {
    CompactTagLds l;
    int at = 0;
    l.HS = 64 * NS;                    // every lane of a chain wavefront owns a slot of the row: the row goes out unconditionally
    l.tok = at; at += (L + 3) & ~3;
    l.hA = at;  at += (L + 1) * l.HS;
    l.hB = at;  at += (L + 1) * l.HS;
    l.misc = at; at += 8;
    l.total = at;
    return l;
}

#if defined(FARNN_PROBES)
constexpr int CT_STAMP_MAX = 4096;
__device__ long long g_ct_stamps[16 * CT_STAMP_MAX];     // FARNN_DBG & 2048: {seq, len, start, set-up done, xcc, se, cu, -, end of wavefront 0..7} (100 MHz clock)
#endif

template <int NS> struct CtRow;
template <> struct CtRow<1> { typedef unsigned v __attribute__((ext_vector_type(2))); };
template <> struct CtRow<2> { typedef unsigned v __attribute__((ext_vector_type(4))); };

// ---- a chain: one wavefront, DIR 0 forward / 1 backward.  NW: 32-bit words of a bitmap row in use (ceil(S / 32)); NLK: 0 none,
// 1 relu, 2 tanh / relu-tanh.  scripts/probe/issue_rate.hip: a lone wavefront pays ~5 cycles per instruction whatever the
// dependencies, ~9 per instruction around a ballot, ~21 per taken branch -- the step is written for its instruction count:
//   * T | W comes merged from the loader (compact.hip.h: merge_planes_kernel), one popcount per row word instead of two; the rare
//     word whose T and W share an edge carries a flag in its block offset and adds its second plane T & W;
//   * one loop over the DISTINCT values among the active sources and nothing in front of it: the first pass takes the first
//     active source's value, ballots who else holds it (compared as bit patterns: a NaN equals itself), counts, and leaves the
//     loop when nobody is left -- the common step runs it once without a taken branch;
//   * block addresses: a VGPR add of the word's offset (one v_readlane from the 64-step window) and the row offset, the second
//     row through the load's immediate offset; the steady-state steps wait with a constant count, the last CT_PF with zero;
//   * the history row goes out unconditionally (rows are 64 NS floats wide), the progress word from every lane.
template <int NW, int NLK, int DIR>
__device__ __forceinline__ void ct_chain(const CompactParams &p, float *ct_smem, const CompactTagLds &lds, int lane, int b, int len, int nsteps
#if defined(FARNN_PROBES)
                                         , long long tk0
#endif
                                         ) {
    constexpr int NS = (NW + 1) / 2;
    constexpr int HS = 64 * NS;
    typedef typename CtRow<NS>::v rowv;
    const int S = p.S;
    const char *bits = reinterpret_cast<const char *>(DIR == 0 ? p.mF : p.mB);
    const char *xbits = reinterpret_cast<const char *>(DIR == 0 ? p.xF : p.xB);
    const unsigned *tok = reinterpret_cast<const unsigned *>(ct_smem) + lds.tok;
    const int hbase = DIR == 0 ? lds.hA : lds.hB;
    float a[NS], ov[NS];
    bool ok[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {
        const int j = lane + 64 * k;
        ok[k] = j < S;
        ov[k] = ok[k] ? (p.o ? p.o[j] : 1.0f) : 0.0f;
        float v = ct_smem[hbase + j];                          // row 0 (zero beyond S)
        if (DIR == 1) v *= ov[k];                              // backward input is pre-scaled (:393)
        a[k] = v;
    }
    const unsigned voff = (unsigned)lane * (unsigned)(NS * 8);  // row `lane` of a block; row lane + 64: the load's immediate offset
    int hrow = hbase + HS + lane;                               // ct_smem index of this lane's slot in row 1
    const int progi = lds.misc + 2 + DIR;
    // block offsets (| 1: the word has a second plane): 64 steps' worth in a register (lane l: step window + l), the flags as a mask
    // (the window the REQUESTS read runs CT_PF steps ahead of the one the steps themselves read)
    unsigned tkw, ctkw;
    u64 nextflags, flagbits;            // flagbits bit 0: the current step's flag (shifted down every step)
#define FARNN_CT_WINDOW(t_)                                                                    \
    do {                                                                                       \
        const int ti_ = min((t_) + lane, nsteps - 1);                                          \
        const unsigned raw_ = tok[DIR == 0 ? ti_ : (ti_ < len ? len - 1 - ti_ : ti_)];         \
        nextflags = __ballot((raw_ & 1u) != 0u);                                               \
        tkw = raw_ & ~1u;                                                                      \
    } while (0)
    // The ring: CT_PF steps x NS rows per lane in FIXED registers the compiler never allocates (v[CT_RING0 ...]: the kernel is
    // compiled with amdgpu_num_vgpr(CT_RING0)).  A request names its destination registers only inside the instruction text; they
    // become values the compiler can see at the counted wait of their step and nowhere earlier.  (As "=v" outputs of the request
    // they were ordinary values from the moment of the request on, and the register allocator moved them between registers at
    // control-flow joins while the loads were in flight -- copies of stale data: 22 of 42 parity cases failed.)
    // (the clobber list: a row value of an earlier step still alive in these registers is consumed or copied out before the request)
#define FARNN_CT_ISSUE_R(off_, R0_, R1_, ...)                                                  \
    do {                                                                                       \
        const unsigned va_ = (off_) + voff;                                                    \
        if constexpr (NS == 2)                                                                 \
            asm volatile("global_load_dwordx4 " R0_ ", %0, %1\n\t"                             \
                         "global_load_dwordx4 " R1_ ", %0, %1 offset:1024" :: "v"(va_), "s"(bits) : "memory", __VA_ARGS__); \
        else                                                                                   \
            asm volatile("global_load_dwordx2 " R0_ ", %0, %1" :: "v"(va_), "s"(bits) : "memory", __VA_ARGS__); \
    } while (0)
#define FARNN_CT_ISSUE(u_, off_)                                                               \
    do {                                                                                       \
        if constexpr (NS == 2) {                                                               \
            if constexpr ((u_) == 0) FARNN_CT_ISSUE_R(off_, "v[128:131]", "v[132:135]", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135"); \
            else if constexpr ((u_) == 1) FARNN_CT_ISSUE_R(off_, "v[136:139]", "v[140:143]", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143"); \
            else if constexpr ((u_) == 2) FARNN_CT_ISSUE_R(off_, "v[144:147]", "v[148:151]", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151"); \
            else FARNN_CT_ISSUE_R(off_, "v[152:155]", "v[156:159]", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159"); \
        } else {                                                                               \
            if constexpr ((u_) == 0) FARNN_CT_ISSUE_R(off_, "v[128:129]", "", "v128", "v129"); \
            else if constexpr ((u_) == 1) FARNN_CT_ISSUE_R(off_, "v[136:137]", "", "v136", "v137"); \
            else if constexpr ((u_) == 2) FARNN_CT_ISSUE_R(off_, "v[144:145]", "", "v144", "v145"); \
            else FARNN_CT_ISSUE_R(off_, "v[152:153]", "", "v152", "v153"); \
        }                                                                                      \
    } while (0)
    // the counted wait of a step: its rows become values here
#define FARNN_CT_WAIT_R(cnt_, R0_, R1_)                                                        \
    do {                                                                                       \
        if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(%2)" : "={" R0_ "}"(cur[0]), "={" R1_ "}"(cur[NS - 1]) : "n"(cnt_) : "memory"); \
        else                   asm volatile("s_waitcnt vmcnt(%1)" : "={" R0_ "}"(cur[0]) : "n"(cnt_) : "memory"); \
    } while (0)
#define FARNN_CT_WAIT(u_, cnt_)                                                                \
    do {                                                                                       \
        if constexpr (NS == 2) {                                                               \
            if constexpr ((u_) == 0) FARNN_CT_WAIT_R(cnt_, "v[128:131]", "v[132:135]"); \
            else if constexpr ((u_) == 1) FARNN_CT_WAIT_R(cnt_, "v[136:139]", "v[140:143]"); \
            else if constexpr ((u_) == 2) FARNN_CT_WAIT_R(cnt_, "v[144:147]", "v[148:151]"); \
            else FARNN_CT_WAIT_R(cnt_, "v[152:155]", "v[156:159]"); \
        } else {                                                                               \
            if constexpr ((u_) == 0) FARNN_CT_WAIT_R(cnt_, "v[128:129]", "v[128:129]"); \
            else if constexpr ((u_) == 1) FARNN_CT_WAIT_R(cnt_, "v[136:137]", "v[136:137]"); \
            else if constexpr ((u_) == 2) FARNN_CT_WAIT_R(cnt_, "v[144:145]", "v[144:145]"); \
            else FARNN_CT_WAIT_R(cnt_, "v[152:153]", "v[152:153]"); \
        }                                                                                      \
    } while (0)
    FARNN_CT_WINDOW(0);
    ctkw = tkw; flagbits = nextflags;
#define FARNN_CT_FIRST(u_) if ((u_) < nsteps) FARNN_CT_ISSUE(u_, (unsigned)__builtin_amdgcn_readlane((int)tkw, u_));
    FARNN_CT_FIRST(0) FARNN_CT_FIRST(1) FARNN_CT_FIRST(2) FARNN_CT_FIRST(3)
#undef FARNN_CT_FIRST
    const int nl_mode = p.nl;
#if defined(FARNN_PROBES)
    int npass = 0;
    const long long tb0 = (long long)__builtin_amdgcn_s_memtime();
#endif
    // one step; U: the ring slot (t mod CT_PF), STEADY: the block of step t + CT_PF is requested behind it (t + CT_PF < nsteps)
    using std::integral_constant;
    int progb = 4 * progi;               // the progress word's byte offset in ct_smem, kept in a VGPR (else every step moves it there
    asm volatile("" : "+v"(progb));      //  from an SGPR; an offset, not a pointer: a laundered pointer is a generic one -- flat_store)
    auto step = [&](auto U_, auto STEADY_, const int t) {
        constexpr int U = decltype(U_)::value;
        constexpr bool STEADY = decltype(STEADY_)::value;
        u64 rem[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) rem[s] = __ballot(a[s] != 0.0f);
        rowv cur[NS];
        if constexpr (STEADY) FARNN_CT_WAIT(U, (CT_PF - 1) * NS); else FARNN_CT_WAIT(U, 0);
        float acc[NS];
#pragma unroll
        for (int k = 0; k < NS; k++) acc[k] = 0.0f;
        rowv xr[NS];
        // The sources that hold the value v add v * (their edges into destination j) to it: one pass per DISTINCT value among the
        // active sources (integer values make any grouping of the sum exact).  No active source: one pass over an empty set.
        // PLANE 0: the merged T | W rows of this step (cur); PLANE 1: the second plane's rows (xr), counted on top by the rare step whose
        // word has one -- a loop of its own behind the common one, so that the common step carries no trace of it
        auto one_pass = [&](auto PLANE_) {
            constexpr int PLANE = decltype(PLANE_)::value;
            int vi;
            {
                // (s_ff1 of an empty mask is -1; v_readlane takes the lane modulo 64: lane 63, whose value the select below drops or a
                //  count of zero multiplies.  The instruction by hand: __builtin_ctzll of 0 is undefined, its guarded forms cost four)
                int i0;
                asm("s_ff1_i32_b64 %0, %1" : "=s"(i0) : "s"(rem[0]));
                const int va = __builtin_amdgcn_readlane(__float_as_int(a[0]), i0 & 63);
                vi = va;
                if constexpr (NS == 2) {
                    int i1;
                    asm("s_ff1_i32_b64 %0, %1" : "=s"(i1) : "s"(rem[NS - 1]));
                    const int vb = __builtin_amdgcn_readlane(__float_as_int(a[NS - 1]), i1 & 63);
                    vi = rem[0] ? va : vb;
                }
            }
            u64 mv[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                mv[s] = __ballot(__float_as_int(a[s]) == vi) & rem[s];      // (bit patterns: a NaN is in its own group)
                rem[s] ^= mv[s];
            }
#pragma unroll
            for (int k = 0; k < NS; k++) {
                int cnt = 0;
#pragma unroll
                for (int w = 0; w < NW; w++)
                    cnt += __popc((PLANE == 0 ? cur[k][w] : xr[k][w]) & (unsigned)(mv[w >> 1] >> (32 * (w & 1))));
                acc[k] = fmaf(__int_as_float(vi), (float)cnt, acc[k]);
            }
#if defined(FARNN_PROBES)
            npass++;
#endif
        };
        auto left = [&]() { return NS == 2 ? (rem[0] | rem[NS - 1]) != 0ull : rem[0] != 0ull; };
        const bool second = (flagbits & 1ull) != 0ull;           // this word's T and W share an edge: its second plane counts too
        flagbits >>= 1;
        one_pass(integral_constant<int, 0>());                   // the common step: one value, no taken branch
        if (__builtin_expect(left(), 0))
            do one_pass(integral_constant<int, 0>()); while (left());
        if (__builtin_expect(second, 0)) {
            const unsigned vx = (unsigned)__builtin_amdgcn_readlane((int)ctkw, t & 63) + voff;
            if constexpr (NS == 2)
                asm volatile("global_load_dwordx4 %0, %2, %3\n\t"
                             "global_load_dwordx4 %1, %2, %3 offset:1024\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(xr[0]), "=&v"(xr[NS - 1]) : "v"(vx), "s"(xbits) : "memory");
            else
                asm volatile("global_load_dwordx2 %0, %1, %2\n\t"
                             "s_waitcnt vmcnt(0)" : "=&v"(xr[0]) : "v"(vx), "s"(xbits) : "memory");
#pragma unroll
            for (int s = 0; s < NS; s++) rem[s] = __ballot(a[s] != 0.0f);
            do one_pass(integral_constant<int, 1>()); while (left());
        }
        // the block of step t + CT_PF into the slot that has just been read (its offset: lane (t + CT_PF) & 63 of the window)
        if constexpr (STEADY) FARNN_CT_ISSUE(U, (unsigned)__builtin_amdgcn_readlane((int)tkw, (t + CT_PF) & 63));
#pragma unroll
        for (int k = 0; k < NS; k++) {
            // (:377-386) / (:393-402): the forward chain scales by o before the non-linearity, the backward chain after it; a lane
            // without a state hands an exact zero to the next step whatever its rows held
            const float x = DIR == 0 ? acc[k] * ov[k] : (ok[k] ? acc[k] : 0.0f);
            const float hn = NLK == 0 ? x : (NLK == 1 ? fmaxf(x, 0.0f) : cc_nl(x, nl_mode));
            ct_smem[hrow + 64 * k] = hn;
            a[k] = DIR == 0 ? (ok[k] ? hn : 0.0f) : hn * ov[k];
        }
        hrow += HS;
        // rows 0 .. t + 1 of this direction are complete: the progress word, behind the row in this wavefront's LDS order (every
        // lane stores the same word: no exec juggling)
        __hip_atomic_store(reinterpret_cast<int *>(reinterpret_cast<char *>(ct_smem) + progb), t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    static_assert(CT_PF == 4 && 64 % CT_PF == 0, "the step lists below are written out for a ring of four steps");
    int t0 = 0;
    // groups of CT_PF steps that all have a block to request: no per-step bounds
    for (; t0 + 2 * CT_PF <= nsteps; t0 += CT_PF) {
        if ((t0 & 63) == 0) { ctkw = tkw; flagbits = nextflags; }  // the steps enter the window their requests opened CT_PF steps ago
        if (((t0 + CT_PF) & 63) == 0) {                          // the next window opens with step t0 + CT_PF: its offsets now
            step(integral_constant<int, 0>(), integral_constant<bool, false>(), t0);     // (never taken at L <= 64)
            FARNN_CT_WINDOW(t0 + CT_PF);
            FARNN_CT_ISSUE(0, (unsigned)__builtin_amdgcn_readlane((int)tkw, 0));
        } else
            step(integral_constant<int, 0>(), integral_constant<bool, true>(), t0);
        step(integral_constant<int, 1>(), integral_constant<bool, true>(), t0 + 1);
        step(integral_constant<int, 2>(), integral_constant<bool, true>(), t0 + 2);
        step(integral_constant<int, 3>(), integral_constant<bool, true>(), t0 + 3);
    }
    // the rest: up to CT_PF - 1 steps that still request a block, then the last CT_PF (or fewer) that do not
    for (; t0 < nsteps; t0 += CT_PF) {
        if ((t0 & 63) == 0) { ctkw = tkw; flagbits = nextflags; }
#define FARNN_CT_REST(u_)                                                                      \
        if (t0 + (u_) < nsteps) {                                                              \
            if (t0 + (u_) + CT_PF < nsteps) {                                                  \
                if ((u_) == 0 && ((t0 + CT_PF) & 63) == 0) {                                   \
                    step(integral_constant<int, u_>(), integral_constant<bool, false>(), t0 + (u_)); \
                    FARNN_CT_WINDOW(t0 + CT_PF);                                               \
                    FARNN_CT_ISSUE(u_, (unsigned)__builtin_amdgcn_readlane((int)tkw, 0));      \
                } else                                                                         \
                    step(integral_constant<int, u_>(), integral_constant<bool, true>(), t0 + (u_)); \
            } else                                                                             \
                step(integral_constant<int, u_>(), integral_constant<bool, false>(), t0 + (u_)); \
        }
        FARNN_CT_REST(0) FARNN_CT_REST(1) FARNN_CT_REST(2) FARNN_CT_REST(3)
#undef FARNN_CT_REST
    }
#if defined(FARNN_PROBES)
    if ((p.dbg & 4096) && nsteps == p.L && lane == 0)
        printf("compact tag seq %d dir %d: %d steps, %lld cycles per step, %d.%02d passes per step\n", b, DIR, nsteps,
               (long long)(__builtin_amdgcn_s_memtime() - tb0) / nsteps, npass / nsteps, (100 * npass / nsteps) % 100);
    (void)tk0;
#endif
#undef FARNN_CT_WAIT
#undef FARNN_CT_WAIT_R
#undef FARNN_CT_ISSUE
#undef FARNN_CT_ISSUE_R
#undef FARNN_CT_WINDOW
}

constexpr int CT_RING0 = 128;       // the chains' ring of bitmap rows lives in v[CT_RING0 .. CT_RING0 + 8 CT_PF - 1]: out of the compiler's reach

template <int NW, int NLK>
__global__ void __launch_bounds__(CT_WAVES * 64, 1) __attribute__((amdgpu_num_vgpr(CT_RING0)))
compact_tag_kernel(const CompactParams p, const ScoreParams sp) {
    static_assert(NW >= 1 && NW <= 4, "a bitmap row of up to 128 states");
    constexpr int NS = (NW + 1) / 2;
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
    const int S = p.S, L = p.L;
    const CompactTagLds lds = compact_tag_lds(L, NS);
    constexpr int HS = 64 * NS;
    unsigned *tok = reinterpret_cast<unsigned *>(ct_smem) + lds.tok;     // [nsteps] byte offset of token q's bitmap block | second-plane flag
    float *hA = ct_smem + lds.hA, *hB = ct_smem + lds.hB;
    int *misc = reinterpret_cast<int *>(ct_smem) + lds.misc;          // [0] flat offset, [2] / [3] rows complete (forward / backward)

    // ---- set-up: the tokens (as block offsets), row 0 of both histories -------------------------------------------------
    for (int q = tid; q < nsteps; q += CT_WAVES * 64) tok[q] = p.tokoff[clamp_tok(p.x[(long long)b * L + q], p.V)];
    for (int j = tid; j < HS; j += CT_WAVES * 64) {
        hA[j] = j < S ? p.h0[j] : 0.0f;
        hB[j] = j < S ? p.hT[j] : 0.0f;
    }
    if (tid < 8) misc[tid] = 0;
    __syncthreads();
#if defined(FARNN_PROBES)
    const long long tk0 = (long long)__builtin_amdgcn_s_memtime(), tr0 = (long long)__builtin_amdgcn_s_memrealtime();
#define FARNN_CT_TK0 , tk0
#else
#define FARNN_CT_TK0
#endif

    if (w < 2) {
        if (nsteps > 0) {
            __builtin_amdgcn_s_setprio(2);
            if (w == 0) ct_chain<NW, NLK, 0>(p, ct_smem, lds, lane, b, len, nsteps FARNN_CT_TK0);
            else        ct_chain<NW, NLK, 1>(p, ct_smem, lds, lane, b, len, nsteps FARNN_CT_TK0);
            __builtin_amdgcn_s_setprio(0);
        }
#undef FARNN_CT_TK0
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
            const float *fa = hA + (ia + 1) * HS, *fb = hA + (ib + 1) * HS;
            const float *ba = hB + rowB(ia) * HS, *bb = hB + rowB(ib) * HS;
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
#if defined(FARNN_PROBES)
    // per-workgroup stamps (scripts/debug/ct_stamps.py): where it ran, when it started, when each of its wavefronts was done
    if ((p.dbg & 2048) && lane == 0 && slot < CT_STAMP_MAX) {
        long long *o = g_ct_stamps + 16 * (long long)slot;
        o[8 + w] = (long long)__builtin_amdgcn_s_memrealtime();
        if (w == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[0] = b; o[1] = len; o[2] = tr_start; o[3] = tr0; o[4] = xcc & 15u; o[5] = (hwid >> 13) & 7u; o[6] = (hwid >> 8) & 15u;
        }
    }
#endif
}

// the geometries the one-launch form covers
inline bool compact_tag_fits(int V, int S, int L) {
    return S <= 128 && (size_t)compact_tag_lds(L, S <= 64 ? 1 : 2).total * 4 <= (size_t)150 * 1024 &&
           (unsigned long long)V * S * (S <= 64 ? 1 : 2) * 8ull < (1ull << 32) - 4096;    // (the kernel adds lane * NS * 8 and an offset:1024 to the
                                                                                       //  last block's 32-bit offset: 4 KiB short of 4 GiB, it cannot wrap)
}

}  // namespace farnn
