// K1r -- the dense-block recurrence with the blocks loaded straight into REGISTERS, and the label scores + threshold/argmax
// decode (K2) running BESIDE it: one launch per tagging step, nothing but one small tile left to do when the chains end.
//
// Reference: FARNN_S_O_I_S.forward_score (model_onehot.py:372-403 the two time loops, :411-426 the scores),
// local_decode (:162-180).  Per sequence b and direction d (chain.hip.h has the same equations):
//     fwd:  a[k+1] = nl( (a[k] . M[x_k]) * o )          bwd:  bt-chain over the tokens right to left
//     score[i] = O . (a[i+1] * bt[i+1])                  tag[i] = first-argmax(clamped score[i])
//
// One workgroup of eight wavefronts per (sequence, direction); two workgroups share a compute unit.
//   * SIX COMPUTE wavefronts.  The rows of a block are split into 6*G row groups of RPG <= 4 rows; lane (g, c) owns the
//     16-byte column chunk c of group g's rows.  It loads ITS pieces of the next RG_D steps' blocks with plain
//     global_load_dwordx4 (the token ids are known up front), so a step reads nothing but registers and its four state
//     entries: 16 FMAs, one 16-byte partial-sum store to LDS.  The six wavefronts meet on per-wavefront step flags in LDS (a
//     store and a polled load: no s_barrier, so the other two wavefronts are not tied to the step), then each reduces the
//     rows it consumes next (four lanes per row, the quad joined on the DPP network).  No LDS ring, no loader wavefronts:
//     r02's chain_kernel moved every block through LDS twice (LDS-DMA write, ds_read) and kept two steps in flight;
//     here RG_D = 4 steps are in flight per lane.
//   * ONE WRITER wavefront follows the finished states (kept in LDS for the whole sequence: `hist`) and copies them to the
//     HBM stash with write-through (sc1) stores; after each batch of rows it drains its stores (s_waitcnt vmcnt(0)) and
//     publishes the sequence's progress word {launch epoch, rows stored} with one agent-scope store.
//   * ONE SCORER wavefront scores 16-token tiles WHILE the chains run.  score[i] needs a[i+1] and bt[i+1]; with both chains
//     in step, token i of the upper half of the sequence has its bt row long stored when the forward chain reaches it (and
//     the mirror image for the backward chain), so the forward workgroup scores the upper tiles, the backward workgroup the
//     lower ones: own rows from `hist` (LDS), the other direction's rows from the stash.  A tile is [16 x S].[S x K] on the
//     f32 matrix cores (v_mfma_f32_16x16x4_f32: the ascending-s fmaf chain of K2), then priority / clamp / first-index argmax.
// When the chain ends, whatever is left (the tile that holds the last tokens; for short sequences everything) is scored by
// all eight wavefronts together.
//
// Hand-off between the two workgroups of a sequence (MI355X_MICROARCH.md, "inter-workgroup visibility"; this is
// cdna_hip_programming.md Guideline 16's recipe R1, acquire included -- the form that needs no assumption about how many
// workgroups share a compute unit):
//   producer: every stash row is stored write-through (sc1) by ONE wavefront (the writer), which drains them
//             (s_waitcnt vmcnt(0)) and then stores the progress word (relaxed agent-scope atomic = sc1);
//   consumer: ONE lane polls that word (relaxed, sc1), then ONE agent-scope acquire fence (buffer_inv sc1), then the loads
//             (the acquiring wavefront's own; other wavefronts behind s_waitcnt vmcnt(0) + a workgroup barrier); the loads of
//             handed-off rows are sc1 loads on top of that.
// Nobody waits for a workgroup that may not be resident: every wait on the other direction is bounded.  Until its arrival
// a workgroup scores tiles of its OWN half only (the halves are disjoint), then it exchanges the sequence's arrival word
// for {epoch, the mask of its tiles}: the workgroup that finds the other's word there scores every tile in neither mask --
// by then both stashes are complete and published -- so every tile is scored exactly once.  All words carry the launch's
// epoch: nothing is reset between launches and an aborted launch leaves nothing behind that a later one could mistake
// for its own.
#pragma once
#include "common.hip.h"
#include "score_params.hip.h"
#include "launch_order.hip.h"
#include "chain_regs_params.hip.h"
#include "beside.hip.h"

#ifndef FARNN_ABLATE
#define FARNN_ABLATE 0       /* timing-only ablation builds set bits; the shipped library is built with 0 */
#endif

namespace farnn {

// workgroup-visible words in LDS: relaxed atomics (never cached in a register, never reordered by the hardware: the LDS
// executes a wavefront's operations in order); the compiler barriers keep the plain LDS accesses on their side
__device__ __forceinline__ void lds_flag_set(int *f, int v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int lds_flag_get(const int *f) {
    const int v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    return v;
}
// A compute wavefront's step flag = the number of steps whose partial sums it has written.  It lives in the partial-sum
// buffer the step wrote (wavefront i: float 4 i of the buffer's flag area, `fbase` in buffer 0), stored by the same
// ds_write_b128 as the partial sums.  Writer / scorer count state rows by them: row t + 1 of `hist` is complete once every
// wavefront's newer flag reads t + 2, or nsteps + 1 after the last step.
// (PSTR: floats between the two partial-sum buffers -- RG_PART_STRIDE, or rgw_part_stride(RQ) in the wide form)
template <int PSTR = RG_PART_STRIDE>
__device__ __forceinline__ int regs_flag_newest(const float *fbase, int lane) {
    const int *f = reinterpret_cast<const int *>(fbase) + 4 * (lane < RG_NWC ? lane : 0);
    const int v0 = lds_flag_get(f), v1 = lds_flag_get(f + PSTR);
    return max(v0, v1);
}
template <int PSTR = RG_PART_STRIDE>
__device__ __forceinline__ bool regs_rows_reached(const float *fbase, int lane, int rows) {
    return __ballot(regs_flag_newest<PSTR>(fbase, lane) < rows + 1) == 0ull;
}
template <int PSTR = RG_PART_STRIDE>
__device__ __forceinline__ int regs_rows_done(const float *fbase, int lane) {
    int v = regs_flag_newest<PSTR>(fbase, lane);
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) v = min(v, __shfl_xor(v, off, WAVE));
    return __builtin_amdgcn_readfirstlane(v) - 1;
}

__device__ __forceinline__ float quad_sum(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    return x;
}
__device__ __forceinline__ float quad_max(float x) {
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true)));
    return x;
}

}  // namespace farnn
#include "chain_wide.hip.h"      // the compute wavefronts of the wide form (72 < S <= 128)
#include "chain_dest.hip.h"      // the compute wavefronts of the destination-split form (S <= 72, sum semiring)
namespace farnn {

// FARNN_PROBES (profiling build only): s_memtime stamps of the workgroups of full-length sequences, printed at their end
#if defined(FARNN_PROBES)
constexpr int WG_STAMP_MAX = 4096;
__device__ long long g_wg_stamps[16 * WG_STAMP_MAX];    // FARNN_DBG & 2048 (chain_regs_kernel): {seq, len, start, end, cycles, xcc, se, cu,
                                                        //  setup, chain, scorer done - chain end, meeting - chain end, tiles + arrival} per workgroup
#define FARNN_RG_STAMP(i) do { if (probe && lane == 0) stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define FARNN_RG_STAMP(i) do { } while (0)
#endif

// The kernel's body as a function of (LDS base, thread index 0..511, item = 2 * launch slot + direction): chain_regs_kernel runs
// it on one workgroup per item; chain_viterbi_kernel (chain_viterbi.hip) runs the two directions of a sequence as the two
// halves of ONE sixteen-wavefront workgroup and hangs the CRF decode behind them.  Workgroup barriers in here are reached by
// both halves alike (the same sequence, the same length).  *b_out: the sequence the slot maps to.
// RQ / D: rows of 16 bytes per lane and step, steps in flight.  (RG_RQ, RG_D) is the form for S <= 72 (two workgroups per compute
// unit); RQ > RG_RQ the wide form (chain_wide.hip.h: 72 < S <= 128, one workgroup per compute unit, launch order longest first).
// LMO: label-map path only (the paired wide form): the matrix-core tile code is compiled out.
// DEST: the compute wavefronts split the block by destination (chain_dest.hip.h; narrow form, sum semiring).
template <bool MAXSR, bool SCORE, bool NLX, int RQ = RG_RQ, int D = RG_D, bool LMO = false, bool DEST = false>
__device__ __forceinline__ void chain_regs_body(const RegsParams &p, float *smem, const int tid, const int item, int *b_out) {
    constexpr bool WIDE = RQ != RG_RQ;
    static_assert(!DEST || (!WIDE && !MAXSR), "the destination-split form exists for the narrow kernel and the sum semiring");
    constexpr int PSTR = WIDE ? rgw_part_stride(RQ) : RG_PART_STRIDE;     // floats between the two partial-sum buffers
    constexpr int NG = WIDE ? RGW_NG : RG_NG;                         // state groups of 16 the scoring stage reaches
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nthreads = RG_WAVES * 64;
    // ids 2s / 2s + 1: forward chains on the even XCDs, backward chains on the odd ones -- an L2 caches one direction's blocks
    const int dir = item & 1, slot = item >> 1;
    const int S = p.S, SP = p.SP, G = p.G, RPG = p.RPG, NP = RG_NWC * G;
    const int PS = WIDE ? p.PS : SP;                                  // floats between two partial-sum vectors
    const bool lm_path = SCORE && (LMO || bs_label_map_path(p.sp));     // the tiles are scored through the label map: no tile areas in LDS
    const RegsLds lds = regs_lds(p.L, SP, NP, p.sp.c16, p.sp.Kc, SCORE, RQ, lm_path, DEST);
    long long *tokoff = reinterpret_cast<long long *>(smem + lds.tok);     // [nsteps] byte offset of step k's block
    float *part = smem + lds.part, *ol = smem + lds.ol, *hist = smem + lds.hist;
    float *ab = smem + lds.ab, *scl = smem + lds.scl, *obuf = smem + lds.obuf;
    int *misc = reinterpret_cast<int *>(smem + lds.misc);

    int b = p.order ? p.order[slot] : slot;
    // (a compute unit holds two workgroups of the narrow form: slots i and i + B/2 pair a long with a short sequence; it holds ONE
    //  of the wide form, and workgroups start in slot order as compute units fall free: longest first)
    int len_sel = -1;
    if (p.sort) b = select_by_length_rank(p.len, p.B, p.L, p.pair ? folded_rank(slot, p.B) : slot, reinterpret_cast<int *>(hist), tid, nthreads, &len_sel);
    if (b_out) *b_out = b;
    const int len = len_sel >= 0 ? len_sel : clamp_len(p.len[b], p.L);      // (the selection hands the length back: no len[b] load behind it)
    const int nsteps = p.full ? p.L : len;
    const float *hinit = dir == 0 ? p.h0 : p.hT;
#if defined(FARNN_PROBES)
    __shared__ long long stamps[8];
    const bool quiet = (p.dbg & 2048) != 0;                               // 2048: every workgroup stamps, nobody prints (g_wg_stamps)
    const bool probe = ((p.dbg & 32768) && nsteps == p.L && p.L >= 32) || quiet;   // 32768: the full-length workgroups print their timeline
                                                                                   // (off by default: the A/B build is also what the A/B forms are timed with)
    if (w == 0) FARNN_RG_STAMP(0);
#endif

    // (destination split: the compute wavefronts request their first blocks BEFORE the set-up -- chain_dest.hip.h, regs_dest_prime)
    // (not in the instantiation that carries the matrix-core tile code: with the ring live across the set-up it spilled 20 VGPRs)
    constexpr bool PRIME_EARLY = DEST && (!SCORE || LMO);
    DestRing<RD_D, RD_LPR> dring;
    if constexpr (PRIME_EARLY)
        if (w < RG_NWC && nsteps > 0) regs_dest_prime<RD_D, RD_LPR>(p, dir, w, lane, nsteps, len, b, dring);
    // ---- set-up ----------------------------------------------------------------------------------------------------------
    for (int k = tid; k < nsteps; k += nthreads) {
        const int idx = (dir == 0) ? k : (k < len ? len - 1 - k : k);
        tokoff[k] = (long long)clamp_tok(p.x[(long long)b * p.L + idx], p.V) * p.blk * 4;
    }
    for (int j = tid; j < SP; j += nthreads) ol[j] = (p.o && j < S) ? p.o[j] : 1.0f;
    for (int j = tid; j < (nsteps + 1) * SP; j += nthreads) hist[j] = (j < S) ? hinit[j] : 0.0f;     // row 0; pad columns zero
    if (tid < 32) misc[tid] = tid == RGM_ACQ ? -1 : 0;
    if (tid < 2) part[tid * PSTR + PSTR - 1] = MAXSR ? -INFINITY : 0.0f;     // the reduction's identity (masked reads)
    if (tid < 2 * 4 * RG_NWC) {                     // the step flags (DEST, chain_dest.hip.h: a flag of 1 = "row 0 is there", buffer 0 only)
        const int fb = tid / (4 * RG_NWC), fk = tid % (4 * RG_NWC);
        part[fb * PSTR + NP * PS + fk] = (DEST && fb == 0 && (fk & 3) == 0) ? __int_as_float(1) : 0.0f;
    }
    if constexpr (DEST) {
        // exchange row 0 = the initial state (scaled by o for the backward chain, model_onehot.py:393), zeros behind S and in row 1
        for (int j = tid; j < 2 * RD_XS; j += nthreads)
            smem[lds.xd + j] = j < S ? hinit[j] * ((dir == 1 && p.o) ? p.o[j] : 1.0f) : 0.0f;
    }
    if (WIDE && tid < RG_NWC * RGW_XCH) smem[lds.xch + tid] = 0.0f;          // exchange slots without a row stay exact zeros
    __syncthreads();
    if (w == 0) FARNN_RG_STAMP(1);
#if FARNN_ABLATE & 256                               /* 256 = set-up only */
    return;
#endif

    const float *sflag = part + NP * PS;            // the step flags (buffer 0; buffer 1 is PSTR floats on)
    float *stash = (dir == 0 ? p.A : p.Bk) + (long long)b * (p.L + 1) * SP;
    const int ntl = (nsteps + RG_TT - 1) / RG_TT;
    int kmid = 0, pubmax = 0;                        // this workgroup's half of the tiles; the last row the other half needs of it
    if (SCORE) bs_halves(dir, len, nsteps, kmid, pubmax);
    BesideParams bs;                                 // the scoring stage's view of the parameters (beside.hip.h)
    bs.A = p.A; bs.Bk = p.Bk; bs.B = p.B; bs.L = p.L; bs.SP = p.SP; bs.CPR = p.CPR; bs.prog = p.prog; bs.arr = p.arr;
    bs.done = p.done; bs.spin = p.spin; bs.dbg = p.dbg; bs.sp = p.sp;
    bs.epoch = (SCORE && w >= RG_NWC) ? (p.done ? bs_launch_epoch(p.done, lane) : p.epoch_host) : 0u;   // the writer's and the scorer's wavefront use it
    int wr_next = 0;                                 // (writer wavefront) the next state row to copy to the stash
    // the output matrix is a label map and only tags are asked for (label_map.hip.h): the tiles left when the chain ends are scored
    // token by token, a few per wavefront, instead of on the matrix cores.  Every wavefront fetches its two packed words of the map
    // now (two VGPRs for the length of the chain; fetched at the chain's end they would cost an exposed L2 round trip)
    unsigned lm_pk0 = 0u, lm_pk1 = 0u;
    if (lm_path) lm_load_packed(bs.sp.lm, lane, lm_pk0, lm_pk1);
    if (w < RG_NWC) {
        // =================================================================================================================
        // compute wavefronts
        // =================================================================================================================
        if constexpr (WIDE) {
            if (nsteps > 0) {
                __builtin_amdgcn_s_setprio(2);
                regs_compute_wide<MAXSR, NLX, RQ, D>(p, dir, w, lane, nsteps, tokoff, part, ol, hist, smem + lds.xch);
                __builtin_amdgcn_s_setprio(0);
                if (w == 0) FARNN_RG_STAMP(2);
#if defined(FARNN_PROBES)
                if (!SCORE && probe && !quiet && w == 0 && lane == 0)
                    printf("seq %d dir %d (wide): setup %lld, chain %lld (%lld per step)\n", b, dir, stamps[1] - stamps[0], stamps[2] - stamps[1],
                           (stamps[2] - stamps[1]) / nsteps);
#endif
            }
        } else if constexpr (DEST) {
            if (nsteps > 0) {
                __builtin_amdgcn_s_setprio(2);
#if defined(FARNN_PROBES)
                const bool probe_ = probe;
#else
                const bool probe_ = false;
#endif
                if constexpr (!PRIME_EARLY) regs_dest_prime<RD_D, RD_LPR>(p, dir, w, lane, nsteps, len, b, dring);
                regs_compute_dest<NLX, RD_D, RD_LPR>(p, dir, w, lane, nsteps, tokoff, part + NP * PS, ol, hist, smem + lds.xd, probe_, b, dring);
                __builtin_amdgcn_s_setprio(0);
                if (w == 0) FARNN_RG_STAMP(2);
#if defined(FARNN_PROBES)
                if (!SCORE && probe && !quiet && w == 0 && lane == 0)
                    printf("seq %d dir %d (destination split): setup %lld, chain %lld (%lld per step)\n", b, dir, stamps[1] - stamps[0], stamps[2] - stamps[1],
                           (stamps[2] - stamps[1]) / nsteps);
#endif
            }
        } else if (nsteps > 0) {
            __builtin_amdgcn_s_setprio(2);
            int g = lane / p.CPR;
            const int c = lane - g * p.CPR;
            const bool active = g < G;
            if (!active) g = 0;                                   // idle lanes shadow group 0 (same lines, results unused)
            const int gid = w * G + g, row0 = gid * RPG;
            unsigned voff[RG_RQ];
#pragma unroll
            for (int u = 0; u < RG_RQ; u++) voff[u] = ((unsigned)(row0 + u) * (unsigned)SP + (unsigned)c * 4u) * 4u;
            const char *Mbase = reinterpret_cast<const char *>(dir == 0 ? p.Mf : p.Mb);
            const int rows_w = G * RPG;
            const int rj = lane >> 2, rs = lane & 3;              // reduce: four lanes per row
            const int my_row = w * rows_w + rj;
            const bool my_valid = rj < rows_w && my_row < S;
            const bool my_writer = my_valid && rs == 0;
            const float my_o = my_valid ? ol[my_row] : 1.0f;
            const float c_pre = dir == 0 ? my_o : 1.0f, c_post = dir == 0 ? 1.0f : my_o;
            const float ninf = -INFINITY;
            bool okrow[RG_RQ];
#pragma unroll
            for (int u = 0; u < RG_RQ; u++) okrow[u] = u < RPG && row0 + u < S;
            // Everything a step addresses in LDS is a per-lane pointer computed ONCE, and every lane always has somewhere to
            // read or write: the two partial-sum buffers sit RG_PART_STRIDE floats apart (a compile-time immediate on the
            // ds_ instructions: the unrolled step knows its buffer), idle lanes store to a dump slot, masked reads point at a
            // word that holds the sum's identity.  No EXEC juggling, no address arithmetic, no select in the step.
            constexpr int NQI = (RG_NWC * RG_MAXG + 3) / 4;      // partial vectors per reducing lane, at most
            const float *ident = part + RG_PART_STRIDE - 1;      // the last word of a buffer (never a partial sum: NP * SP < RG_PART_STRIDE)
                                                                 // holds 0.0f (sum) or -inf (max), in BOTH buffers: the reads add the buffer's offset
            float *dump = part + 2 * RG_PART_STRIDE;             // [64][4]
            const float *qptr[NQI];                              // this lane's partial-vector entries in buffer 0
#pragma unroll
            for (int i = 0; i < NQI; i++)
                qptr[i] = (my_valid && rs + 4 * i < NP) ? part + (rs + 4 * i) * SP + my_row : ident;
            // (an idle lane's slot lies in the buffer's unused tail, behind the NP partial vectors and the six flag slots: the step
            //  adds the buffer's offset to every partial-sum pointer, so the slot must exist in both buffers.)  The FIRST idle
            //  lane stores the wavefront's step flag with the same instruction: its slot is the flag slot, its "partial sum" the
            //  step count.  (No idle lane -- 64 = G * CPR: one more ds_write_b32.)
            const int nact = G * p.CPR;                           // lanes that own a piece of the block
            const bool has_flane = nact < 64;
            const bool is_flane = lane == nact;
            float *fslot = part + NP * SP + 4 * w;                // this wavefront's flag (buffer 0)
            float *wptr = active ? part + gid * SP + c * 4 : (is_flane ? fslot : part + NP * SP + 4 * RG_NWC + (lane - nact) * 4);
            float *hptr = my_writer ? hist + SP + my_row : dump + lane;     // where this lane's new state goes (dump: consecutive words --
                                                                            // at a stride of four they met four to a bank: 25 % of the LDS cycles were conflicts)
            const int hstep = my_writer ? SP : 0;
            // The state entries a lane multiplies with -- rows g * RPG + u of this wavefront's share -- come straight from the
            // lanes that finish them (row j sits in lanes 4j .. 4j + 3 after the quad reduce): one ds_bpermute each, no
            // LDS store + load round trip.  hs[u] holds them from step to step (pre-scaled for the backward chain, :393).
            int bsrc[RG_RQ];
            float hs[RG_RQ];
#pragma unroll
            for (int u = 0; u < RG_RQ; u++) {
                // byte address of lane 4 * (g * RPG + u); a slot without a row (u >= RPG, i.e. RPG < 4) reads lane 63 instead: with
                // fewer than four rows per group the wavefront's last row slot (lanes 60-63) holds no row, and a lane without a row
                // exchanges an exact 0 -- so the result needs no select on u < RPG behind the exchange
                bsrc[u] = u < RPG ? 16 * (g * RPG + u) : 4 * 63;
                // (from LDS -- row 0 of `hist`, `ol` -- not from global memory: a compiler-counted global load feeding this
                //  loop-carried value makes hipcc put s_waitcnt vmcnt(0) into the loop, which drains the ring every step)
                hs[u] = okrow[u] ? hist[row0 + u] * (dir == 1 ? ol[row0 + u] : 1.0f) : 0.0f;
            }
            const int nl_mode = p.nl;
            const bool nl_relu = nl_mode == FARNN_NL_RELU;
            const int *pflag = reinterpret_cast<const int *>(part + NP * SP) + 4 * (lane < RG_NWC ? lane : 0);   // the partners' flags (buffer 0)

            // The ring: RG_D steps x RG_RQ rows of 16 bytes per lane, loaded by inline asm so that NO compiler wait ever
            // drains it (hipcc's own bookkeeping merges the loop's back edge into vmcnt(0): measured, the ring then has no depth).
            // An asm load's destination counts as written at the statement: every consumer sits behind a wait statement
            // that names the four registers "+v" (cdna_hip_programming.md 5.7, form ii).  Loads retire in issue order, so
            // step t's four pieces have landed once at most the pieces of the steps issued after it are outstanding.
            v4f r[RG_D][RG_RQ];
            // Block addresses: the byte offsets of 64 steps' blocks sit in a register pair (lane l: step window + l), a step's
            // offset is two v_readlane -- no LDS read in the step; the window is reloaded from LDS once per 64 steps.
#define FARNN_RG_WINDOW(t_)                                                                    \
            do {                                                                               \
                const int ti_ = (t_) + lane;                                                   \
                const long long o_ = tokoff[ti_ < nsteps ? ti_ : nsteps - 1];                  \
                tkw_lo = (int)(unsigned)o_; tkw_hi = (int)(unsigned)(o_ >> 32);                \
            } while (0)
#define FARNN_RG_BASE(t_, lo_, hi_)                                                            \
            do {                                                                               \
                const int li_ = (t_) & 63;                                                     \
                lo_ = (unsigned)__builtin_amdgcn_readlane(tkw_lo, li_);                        \
                hi_ = (unsigned)__builtin_amdgcn_readlane(tkw_hi, li_);                        \
            } while (0)
#define FARNN_RG_ISSUE(d, lo_, hi_)                                                            \
            do {                                                                               \
                const char *bp_ = Mbase + (((long long)(hi_) << 32) | (lo_));                  \
                asm volatile("s_nop 4\n\t"                                                     \
                             "global_load_dwordx4 %0, %4, %8\n\t"                              \
                             "global_load_dwordx4 %1, %5, %8\n\t"                              \
                             "global_load_dwordx4 %2, %6, %8\n\t"                              \
                             "global_load_dwordx4 %3, %7, %8"                                  \
                             : "=&v"(r[d][0]), "=&v"(r[d][1]), "=&v"(r[d][2]), "=&v"(r[d][3]) \
                             : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(bp_)); \
            } while (0)
            // ONE wait statement per step (several, one per count, would meet in a phi: the compiler then copies the ring's
            // registers in front of the waits -- seen in the ISA -- i.e. before the data has landed).  Steady state: the three
            // younger steps' 12 pieces may stay outstanding; the last three steps of a sequence drain.
#define FARNN_RG_WAIT(d, rem_)                                                                 \
            asm volatile("s_cmp_ge_i32 %4, 3\n\t"                                              \
                         "s_cbranch_scc1 1f\n\t"                                               \
                         "s_waitcnt vmcnt(0)\n\t"                                              \
                         "s_branch 2f\n"                                                       \
                         "1:\n\t"                                                              \
                         "s_waitcnt vmcnt(12)\n"                                               \
                         "2:"                                                                  \
                         : "+v"(r[d][0]), "+v"(r[d][1]), "+v"(r[d][2]), "+v"(r[d][3]) : "s"(rem_) : "scc")
            unsigned nlo = 0, nhi = 0;
            int tkw_lo, tkw_hi;
            FARNN_RG_WINDOW(0);
#pragma unroll
            for (int d = 0; d < RG_D; d++) {
#pragma unroll
                for (int u = 0; u < RG_RQ; u++) r[d][u] = v4f{0.f, 0.f, 0.f, 0.f};
                if (d < nsteps) {
                    FARNN_RG_BASE(d, nlo, nhi);
                    FARNN_RG_ISSUE(d, nlo, nhi);
                }
            }
#if defined(FARNN_PROBES)
            long long ph[5] = {0, 0, 0, 0, 0}, pt = 0;
#define FARNN_RG_PHASE(i) do { if (probe && w == 0 && (p.dbg & 256)) { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); ph[i] += n_ - pt; pt = n_; } } while (0)
            if (probe && w == 0) pt = (long long)__builtin_amdgcn_s_memtime();
#else
#define FARNN_RG_PHASE(i) do { } while (0)
#endif
            if (RG_D < nsteps) FARNN_RG_BASE(RG_D, nlo, nhi);          // step 0's look-ahead (the window of steps 0 .. 63 is loaded)
#if FARNN_ABLATE & 512                               /* 512 = no steps at all */
            for (int t0 = 0; t0 < 0; t0 += RG_D) {
#else
            for (int t0 = 0; t0 < nsteps; t0 += RG_D) {
#endif
#pragma unroll
                for (int d = 0; d < RG_D; d++) {
                    const int t = t0 + d;
                    if (t >= nsteps) break;
                    static_assert(RG_D == 4 && RG_RQ == 4, "FARNN_RG_WAIT is written out for a 4 x 4 ring");
                    // (nlo / nhi: where the block of step t + RG_D is -- fixed at the END of the step before, between the issue of
                    //  its state exchange and the wait for it)
                    FARNN_RG_WAIT(d, nsteps - 1 - t);            // steps issued after this one: min(RG_D - 1, nsteps - 1 - t)
                    FARNN_RG_PHASE(0);                           // wait for this step's block pieces
                    v4f acc = MAXSR ? v4f{ninf, ninf, ninf, ninf} : v4f{0.f, 0.f, 0.f, 0.f};
#if FARNN_ABLATE & 8                                 /* 8 = no FMAs */
                    acc = r[d][0] + r[d][1] * hs[0] + r[d][2] + r[d][3];
#else
#pragma unroll
                    for (int u = 0; u < RG_RQ; u++) {
                        if (MAXSR) {
                            acc.x = fmaxf(acc.x, okrow[u] ? hs[u] * r[d][u].x : ninf);
                            acc.y = fmaxf(acc.y, okrow[u] ? hs[u] * r[d][u].y : ninf);
                            acc.z = fmaxf(acc.z, okrow[u] ? hs[u] * r[d][u].z : ninf);
                            acc.w = fmaxf(acc.w, okrow[u] ? hs[u] * r[d][u].w : ninf);
                        } else {
                            acc.x = fmaf(hs[u], r[d][u].x, acc.x);
                            acc.y = fmaf(hs[u], r[d][u].y, acc.y);
                            acc.z = fmaf(hs[u], r[d][u].z, acc.z);
                            acc.w = fmaf(hs[u], r[d][u].w, acc.w);
                        }
                    }
#endif
                    const int boff = (d & 1) * RG_PART_STRIDE;           // this step's partial-sum buffer (t and d have the same parity)
#if !(FARNN_ABLATE & 32)                             /* 32 = no partial-sum / flag stores */
                    if (has_flane) acc.x = is_flane ? __int_as_float(t + 1) : acc.x;      // the flag rides in the idle lane's slot
                    asm volatile("" ::: "memory");
                    *reinterpret_cast<v4f *>(wptr + boff) = acc;
                    if (!has_flane && lane == 0) lds_flag_set(reinterpret_cast<int *>(fslot + boff), t + 1);   // behind the partial sums in LDS order
#endif
                    asm volatile("" : "+v"(acc));                // the slot's registers are dead from here: reload them
#if !(FARNN_ABLATE & 1)                              /* ablation builds (scripts/build_ablate.sh): 1 = no block loads */
                    if (t + RG_D < nsteps) FARNN_RG_ISSUE(d, nlo, nhi);
#endif
                    FARNN_RG_PHASE(1);                           // FMAs, partial store, flag, next loads issued
                    // the other wavefronts' flags FIRST, then this lane's share of the partial sums and the next block address in the
                    // same batch: the LDS serves a wavefront in order, so partial sums read behind flags that say "written" are the
                    // written ones -- one round trip when the partners are on time, the whole batch again when they are not
                    float pv[NQI];
#if FARNN_ABLATE & 16                                /* 16 = no LDS reads at all in the step */
#pragma unroll
                    for (int i = 0; i < NQI; i++) pv[i] = acc.x;
#else
                    for (;;) {
                        const int fl = lds_flag_get(pflag + boff);
#pragma unroll
                        for (int i = 0; i < NQI - 1; i++) pv[i] = qptr[i][boff];
                        pv[NQI - 1] = MAXSR ? ninf : 0.0f;
                        if (NP > 4 * (NQI - 1)) pv[NQI - 1] = qptr[NQI - 1][boff];     // (G = 4 only: more than 20 partial vectors)
                        asm volatile("" ::: "memory");
                        if (__ballot(fl < t + 1) == 0ull) break;
#if FARNN_ABLATE & 2                                 /* 2 = nobody waits for the partners (wrong results) */
                        break;
#endif
                    }
#endif
                    FARNN_RG_PHASE(2);                           // the partners' partial sums
                    static_assert(NQI == 6, "the reduction tree below is written out for six partial sums per lane");
                    float s;
                    if (MAXSR) s = fmaxf(fmaxf(fmaxf(pv[0], pv[1]), fmaxf(pv[2], pv[3])), fmaxf(pv[4], pv[5]));
                    else       s = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + (pv[4] + pv[5]);
                    s = MAXSR ? quad_max(s) : quad_sum(s);
                    // (:377-386) / (:393-402): the forward chain scales by o before the non-linearity, the backward chain after it.
                    // One multiply each by a per-lane constant (o or exactly 1.0) instead of a select on the direction: three
                    // dependent instructions fewer on the step's critical path, the same bits
                    const float pre = s * c_pre;
                    float hn;
                    if (NLX) hn = apply_nl(pre, nl_mode);                                  // tanh, relu-tanh, sigmoid
                    else     hn = nl_relu ? fmaxf(pre, 0.0f) : pre;                        // none / relu: no branch in the step
                    const float hx = my_valid ? hn * c_post : 0.0f;                        // what the next step multiplies with
#if !(FARNN_ABLATE & 64)                             /* 64 = no state store */
                    *hptr = hn;                                                            // row t + 1 of `hist` (or the dump slot)
                    hptr += hstep;
#endif
#if !(FARNN_ABLATE & 4)                              /* 4 = no state exchange (wrong results) */
                    int bp[RG_RQ];
#pragma unroll
                    for (int u = 0; u < RG_RQ; u++) bp[u] = __builtin_amdgcn_ds_bpermute(bsrc[u], __float_as_int(hx));
                    // Where the block of step (t + 1) + RG_D is: two v_readlane and their scalar bookkeeping (a dozen instructions a lone
                    // wavefront issues in ~60 cycles), done HERE, while the exchange is in flight -- at the next step's start they sat
                    // behind the wait for it, on the step's critical path.  (The once-per-64-steps window reload is an LDS read: it
                    // queues behind the exchange, and its wait must not sit between the partial-sum store and the loads' issue.)
                    // No branch around it in three steps of four (t0 is a multiple of RG_D = 4, so only d = 3 can reach a multiple of
                    // 64): a branch is a basic-block boundary, and behind one the scheduler cannot pull the next step's scalar
                    // bookkeeping up under the exchange.  Past the last step the v_readlane read some lane of the window: unused.
                    if (d == RG_D - 1 && ((t + 1 + RG_D) & 63) == 0 && t + 1 + RG_D < nsteps) FARNN_RG_WINDOW(t + 1 + RG_D);
                    FARNN_RG_BASE(t + 1 + RG_D, nlo, nhi);
                    asm volatile("" : "+s"(nlo), "+s"(nhi));   // (pinned here)
#pragma unroll
                    for (int u = 0; u < RG_RQ; u++) hs[u] = __int_as_float(bp[u]);
#else
                    if (t + 1 + RG_D < nsteps) {
                        if (((t + 1 + RG_D) & 63) == 0) FARNN_RG_WINDOW(t + 1 + RG_D);
                        FARNN_RG_BASE(t + 1 + RG_D, nlo, nhi);
                    }
#endif
                    FARNN_RG_PHASE(3);                           // row reduce, nonlinearity, state exchange
                }
            }
            if (lane == 0) lds_flag_set(reinterpret_cast<int *>(fslot), nsteps + 1);   // (the last state row is in `hist`: writer / scorer count rows by these flags)
#if defined(FARNN_PROBES)
            if (probe && w == 0 && lane == 0 && (p.dbg & 256))
                printf("seq %d dir %d chain phases, cycles per step: block wait %lld, fma + store + issue %lld, partner wait + partial reads %lld, reduce + exchange %lld\n",
                       b, dir, ph[0] / nsteps, ph[1] / nsteps, ph[2] / nsteps, ph[3] / nsteps);
#endif
#undef FARNN_RG_PHASE
#undef FARNN_RG_ISSUE
#undef FARNN_RG_BASE
#undef FARNN_RG_WINDOW
#undef FARNN_RG_WAIT
            __builtin_amdgcn_s_setprio(0);
            if (w == 0) FARNN_RG_STAMP(2);
#if defined(FARNN_PROBES)
            if (!SCORE && probe && !quiet && w == 0 && lane == 0)
                printf("seq %d dir %d: setup %lld, chain %lld (%lld per step)\n", b, dir, stamps[1] - stamps[0], stamps[2] - stamps[1],
                       (stamps[2] - stamps[1]) / nsteps);
#endif
        }
    } else if (w == RG_NWC) {
        // =================================================================================================================
        // writer wavefront: hist -> stash, progress word
        // =================================================================================================================
        if (SCORE && dir == 0) {                                  // pad positions of LOCAL mode: tag -1, zero score rows
            for (int i = nsteps + lane; i < p.L; i += WAVE)
                if (p.sp.tags) p.sp.tags[(long long)b * p.L + i] = -1;
            if (p.sp.scores)
                for (long long e = (long long)nsteps * p.sp.K + lane; e < (long long)p.L * p.sp.K; e += WAVE)
                    p.sp.scores[(long long)b * p.L * p.sp.K + e] = 0.0f;
        }
        // rows 0 .. wr_next - 1 are stored.  A pass copies EVERY row the chain has finished since the last one (`hist` and the stash
        // are both [row][SP]: rows a .. b are one contiguous run of floats) -- round 4 copied one row per pass, which the
        // destination-split chain outruns: at 780 cycles per step the writer (priority 0 beside six compute wavefronts at 2) fell
        // behind, the progress word reached `pubmax` late, the other workgroup's scorer found nothing to park before its own chain
        // ended and the tiles behind the chains took 20 k cycles instead of 9 k (profiles/r05_wg_lifetimes_*.txt).  When the chain
        // is done the wavefront goes to the workgroup's meeting point at once: what it has not copied yet it copies after the tiles,
        // before the final publish.
        int published = -1;
        auto copy_rows = [&](int r0, int r1) {                    // rows r0 .. r1
            const float *src = hist + r0 * SP;
            float *dst = stash + (long long)r0 * SP;
            const int nfl = (r1 - r0 + 1) * SP;
            if (SCORE) {
                for (int j = 2 * lane; j < nfl; j += 2 * WAVE) st2_agent(dst + j, src[j], src[j + 1]);
            } else {
                for (int j = lane; j < nfl; j += WAVE) dst[j] = src[j];
            }
        };
        if (!SCORE && !p.A) wr_next = nsteps + 1;                // (chain_viterbi_kernel: the rows are consumed where they lie, in LDS)
        while (wr_next <= nsteps) {
            const int done = nsteps > 0 ? regs_rows_done<PSTR>(sflag, lane) : 0;      // rows 0 .. done are complete
            if (done < wr_next) { __builtin_amdgcn_s_sleep(1); continue; }
            if (SCORE && nsteps > 0 && wr_next > pubmax && done >= nsteps) break;     // the chain is done
            // (a run stops at pubmax: the row the other workgroup's tiles wait for is published as soon as it is there)
            const int hi = min(done, (SCORE && wr_next <= pubmax) ? pubmax : nsteps);
            copy_rows(wr_next, hi);
            wr_next = hi + 1;
            // The progress word feeds the other workgroup's tiles: once it covers the last row they need (pubmax), the
            // rest is drained and published once, at the end -- no write-through round trip per pass after that.
            if (SCORE && hi < nsteps && published < pubmax) {
                published = hi;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store of this (the only storing) wavefront has left
                if (lane == 0)
                    __hip_atomic_store(p.prog + (long long)dir * p.B + b, ((unsigned long long)bs.epoch << 32) | (unsigned)published,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if (SCORE) {
        // =================================================================================================================
        // scorer wavefront: tiles of this workgroup's half, while the chain runs
        // =================================================================================================================
        // where sequence b starts in the flat output (utils.py:153-164): every lane has the sum (bs_flat_offset's butterfly); the LDS
        // word is for the other wavefronts, behind the meeting barrier.  (Not read back here: the compiler may serve a lane that did
        // not store from a load hoisted ABOVE lane 0's store -- seen in round 5, when a second lane of this wavefront stored tags.)
        const int fo = bs_flat_offset(bs, b, lane);
        if (lane == 0) misc[RGM_FOFF] = fo;
        const long long foff = fo;
        unsigned mine = 0u;
        int acq = -1;                 // the other direction's progress as polled before this workgroup's latest acquire
        const unsigned long long *oprog = p.prog + (long long)(dir ^ 1) * p.B + b;
        const int kfirst = dir == 0 ? kmid : kmid - 1, kstep = dir == 0 ? 1 : -1, klast = dir == 0 ? ntl : -1;
        // (label-map path: no tile is scored by this wavefront alone -- the whole workgroup scores a pair of tiles in a fraction of a
        //  chain step once the chain is done, and the matrix-core tile areas do not exist in LDS; it parks rows, below)
        if constexpr (!LMO)
        for (int k = kfirst; k != klast && !lm_path; k += kstep) {
            int na, nb;
            bs_tile_need(k, len, nsteps, na, nb);
            const int need_own = dir == 0 ? na : nb, need_oth = dir == 0 ? nb : na;
            if (nsteps - need_own < p.solo_margin) break;             // the chain ends soon: all eight wavefronts will do it
            while (!regs_rows_reached<PSTR>(sflag, lane, need_own)) __builtin_amdgcn_s_sleep(4);
            if (need_oth > acq) {
                int pr = -1;
                for (;;) {
                    pr = lane == 0 ? bs_read_prog(oprog, bs.epoch) : 0;
                    pr = __builtin_amdgcn_readfirstlane(pr);
                    if (pr >= need_oth) break;
                    if (regs_rows_reached<PSTR>(sflag, lane, nsteps)) break;    // our chain is done: no open-ended wait beyond it
                    __builtin_amdgcn_s_sleep(8);
                }
                if (pr < need_oth) break;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // after the poll matched, before the loads
                acq = pr;
            }
            {
                const rg_f32x4 none[NG] = {};
                bs_score_tiles<false, RG_WAVES, NG, RG_NWC>(bs, b, dir, len, nsteps, k, -1, hist, nullptr, nullptr, ab, scl, foff, 0, lane, none);
            }
            mine |= 1u << k;
        }
        // What is left of this half goes to all eight wavefronts when the chain ends.  Until then this wavefront parks the
        // other direction's rows of (up to RG_NOB of) those tiles in LDS, so that the tiles start from LDS and registers.
        {
            int slot = 0;
            for (int k = kfirst; k != klast && slot < RG_NOB; k += kstep) {
                if ((mine >> k) & 1u) continue;
                int na, nb;
                bs_tile_need(k, len, nsteps, na, nb);
                const int need_oth = dir == 0 ? nb : na;
                if (need_oth > acq) {
                    int pr = -1;
                    for (;;) {
                        pr = lane == 0 ? bs_read_prog(oprog, bs.epoch) : 0;
                        pr = __builtin_amdgcn_readfirstlane(pr);
                        if (pr >= need_oth) break;
                        if (regs_rows_reached<PSTR>(sflag, lane, nsteps)) break;
                        __builtin_amdgcn_s_sleep(16);
                    }
                    if (pr < need_oth) break;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    acq = pr;
                }
                bs_park_rows<NG, (LMO && WIDE) ? NG / 2 : NG>(bs, b, dir, len, nsteps, k, obuf + slot * RG_TT * SP, lane);
                if (lane == 0) misc[RGM_PARK + slot] = k + 1;
                slot++;
            }
        }
        if (lane == 0) { misc[RGM_MINE] = (int)mine; misc[RGM_ACQ] = acq; }
        FARNN_RG_STAMP(3);
    }
    if constexpr (!SCORE) {
        // chain_viterbi_kernel (b_out set): the forward half's scorer wavefront has nothing to do while the chains run -- it leaves
        // the decode {length, flat-output offset} in LDS (misc[RGM_FOFF + 1], misc[RGM_FOFF]): two global round trips the decode
        // would otherwise open with (viterbi_hist_body, `pre`)
        if (b_out && dir == 0 && w == RG_WAVES - 1) {
            int partsum = 0;
            if (p.sp.flat && !p.sp.offs)
                for (int j = lane; j < b; j += WAVE) partsum += clamp_len(p.len[j], p.L);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) partsum += __shfl_xor(partsum, off, WAVE);
            if (lane == 0) { misc[RGM_FOFF] = partsum; misc[RGM_FOFF + 1] = len; }
        }
        return;
    }

    // =====================================================================================================================
    // the chain is done.  Before the arrival a workgroup touches only the tiles of its own half: those the other direction's
    // progress (as covered by an acquire) allows are scored now by all eight wavefronts; the arrival word carries the
    // mask of this workgroup's tiles, and the workgroup that finds the other's word there scores whatever neither has.
    // =====================================================================================================================
    if (w == 0) FARNN_RG_STAMP(4);
    bs_finish<RG_WAVES, NG, RG_WAVES - 1, RG_NWC, true, LMO>(bs, b, dir, len, nsteps, kmid, hist, ab, scl, obuf, misc, w, lane, lm_pk0, lm_pk1, bs.epoch,
        [&]() {                                                      // the state rows the writer had not copied yet: issued now, landed by
            if (w == RG_NWC)                                         // the time the tiles are done (it forms no products meanwhile)
                for (; wr_next <= nsteps; wr_next++)
                    for (int j = 2 * lane; j < SP; j += 2 * WAVE)
                        st2_agent(stash + (long long)wr_next * SP + j, hist[wr_next * SP + j], hist[wr_next * SP + j + 1]);
        },
        [&]() {                                                      // every stash row has been stored by the writer: drained (they had the
            if (w == RG_NWC) {                                       // tiles' time to land), then the full count in the progress word
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0)
                    __hip_atomic_store(p.prog + (long long)dir * p.B + b, ((unsigned long long)bs.epoch << 32) | (unsigned)nsteps,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        });
#if defined(FARNN_PROBES)
    if (quiet && tid == 0 && item < WG_STAMP_MAX) {
        long long *o = g_wg_stamps + 16 * (long long)item + 8;
        o[0] = stamps[1] - stamps[0]; o[1] = stamps[2] - stamps[1]; o[2] = stamps[3] - stamps[2]; o[3] = stamps[4] - stamps[2];
        o[4] = (long long)__builtin_amdgcn_s_memtime() - stamps[4];
    }
    if (probe && !quiet && tid == 0) {
        const long long e = (long long)__builtin_amdgcn_s_memtime();
        printf("seq %d dir %d: setup %lld, chain %lld (%lld per step), scorer alone until +%lld, all waves at the meeting point +%lld, "
               "tiles + arrival %lld; tiles alone %d of %d\n", b, dir, stamps[1] - stamps[0], stamps[2] - stamps[1],
               (stamps[2] - stamps[1]) / nsteps, stamps[3] - stamps[2], stamps[4] - stamps[2], e - stamps[4],
               __popc((unsigned)misc[RGM_MINE]), ntl);
    }
#endif
}

// LMO: the instantiation for the label-map path (scores on, the output matrix a label map, tags only): the matrix-core tile
// code is not compiled in -- a quarter of the instructions, 40 instead of 260 spilled SGPRs
template <bool MAXSR, bool SCORE, bool NLX, bool LMO = false, bool DEST = false>
__global__ void __launch_bounds__(RG_WAVES * 64, 4)          // 4 waves per SIMD = 128 VGPRs: two workgroups per compute unit
chain_regs_kernel(const RegsParams p) {
    extern __shared__ __align__(16) float smem[];
#if defined(FARNN_PROBES)
    // FARNN_DBG & 2048: every workgroup's life on the 100 MHz wall clock (s_memrealtime) and in shader cycles (s_memtime), with the
    // compute unit it ran on, into g_wg_stamps (read back through farnn_debug_wg_stamps; no printf: a device printf is a host call
    // that stalls the compute unit's other wavefronts for milliseconds): who finishes last, how far the starts are apart, the clock
    long long w0r = 0, w0c = 0;
    if ((p.dbg & 2048) && threadIdx.x == 0) { w0r = (long long)__builtin_amdgcn_s_memrealtime(); w0c = (long long)__builtin_amdgcn_s_memtime(); }
    int b_probe = -1;
    chain_regs_body<MAXSR, SCORE, NLX, RG_RQ, RG_D, LMO, DEST>(p, smem, (int)threadIdx.x, (int)blockIdx.x, &b_probe);
    if ((p.dbg & 2048) && threadIdx.x == 0 && blockIdx.x < WG_STAMP_MAX) {
        const long long w1r = (long long)__builtin_amdgcn_s_memrealtime(), w1c = (long long)__builtin_amdgcn_s_memtime();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        long long *o = g_wg_stamps + 16 * (long long)blockIdx.x;
        o[0] = b_probe; o[1] = b_probe >= 0 ? (long long)p.len[b_probe] : -1; o[2] = w0r; o[3] = w1r; o[4] = w1c - w0c;
        o[5] = xcc & 15u; o[6] = (hwid >> 13) & 7u; o[7] = (hwid >> 8) & 15u;
    }
#else
    chain_regs_body<MAXSR, SCORE, NLX, RG_RQ, RG_D, LMO, DEST>(p, smem, (int)threadIdx.x, (int)blockIdx.x, nullptr);
#endif
}

template <bool MAXSR, bool SCORE, bool NLX, int RQ, int D>
__global__ void __launch_bounds__(RG_WAVES * 64, 2)          // 2 waves per SIMD = 256 VGPRs: one workgroup per compute unit
chain_wide_kernel(const RegsParams p) {
    extern __shared__ __align__(16) float smem[];
    chain_regs_body<MAXSR, SCORE, NLX, RQ, D>(p, smem, (int)threadIdx.x, (int)blockIdx.x, nullptr);
}

// the wide form PAIRED: a ring of two steps fits 128 VGPRs (RQ <= 9), the label-map path's LDS half a compute unit
template <bool MAXSR, bool NLX, int RQ>
__global__ void __launch_bounds__(RG_WAVES * 64, 4)
chain_wide_paired_kernel(const RegsParams p) {
    extern __shared__ __align__(16) float smem[];
    chain_regs_body<MAXSR, true, NLX, RQ, 2, true>(p, smem, (int)threadIdx.x, (int)blockIdx.x, nullptr);
}

}  // namespace farnn
