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

#ifndef FARNN_ABLATE
#define FARNN_ABLATE 0       /* timing-only ablation builds set bits; the shipped library is built with 0 */
#endif

namespace farnn {

// what tile k needs: forward rows 0..needA and backward rows 0..needB stored (a row = one state, row 0 the initial one)
__device__ __forceinline__ void regs_tile_need(int k, int len, int nsteps, int &needA, int &needB) {
    const int lo = k * RG_TT;
    const int hi = min(lo + RG_TT, nsteps) - 1;
    needA = hi + 1;                                   // alpha of token i is row i + 1
    int nb = 0;                                       // beta of token i is row len - (i + 1); pads of FULL mode: row i + 1
    if (lo < len) nb = len - lo - 1;
    if (hi >= len) nb = max(nb, hi + 1);
    needB = nb;
}

__device__ __forceinline__ int regs_read_prog(const unsigned long long *w, unsigned epoch) {
    const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)(v >> 32) == epoch ? (int)(unsigned)v : -1;
}

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
__device__ __forceinline__ int regs_flag_newest(const float *fbase, int lane) {
    const int *f = reinterpret_cast<const int *>(fbase) + 4 * (lane < RG_NWC ? lane : 0);
    const int v0 = lds_flag_get(f), v1 = lds_flag_get(f + RG_PART_STRIDE);
    return max(v0, v1);
}
__device__ __forceinline__ bool regs_rows_reached(const float *fbase, int lane, int rows) {
    return __ballot(regs_flag_newest(fbase, lane) < rows + 1) == 0ull;
}
__device__ __forceinline__ int regs_rows_done(const float *fbase, int lane) {
    int v = regs_flag_newest(fbase, lane);
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

typedef float rg_f32x4 __attribute__((ext_vector_type(4)));
constexpr int RG_NG = 5;             // state groups of 16 (c16 <= 5: the launcher checks)
constexpr int RG_NOB = 2;            // tiles whose rows of the other direction the scorer parks in LDS ahead of the chain's end

// B fragments (matrix-core image of O^T, score_decode.hip.h) of column block cb, all state groups: one round trip to L2
__device__ __forceinline__ void regs_load_b(const ScoreParams &sp, int cb, int lane, rg_f32x4 (&bf)[RG_NG]) {
    const rg_f32x4 *otm = reinterpret_cast<const rg_f32x4 *>(sp.OTm);
    const int c16 = sp.c16;
#pragma unroll
    for (int g = 0; g < RG_NG; g++) bf[g] = (otm + ((long long)cb * c16 + (g < c16 ? g : c16 - 1)) * 64)[lane];
}

// the other direction's row that token i of sequence b multiplies with (the stash row index and its base)
__device__ __forceinline__ const float *regs_other_row(const RegsParams &p, int b, int dir, int len, int i) {
    const int ai = i + 1, bi = (i + 1 <= len) ? len - (i + 1) : i + 1;
    const long long base = (long long)b * (p.L + 1) * p.SP;
    return dir == 0 ? p.Bk + base + (long long)bi * p.SP : p.A + base + (long long)ai * p.SP;
}

// park the other direction's rows of tile k in LDS (obuf[16][SP]); one wavefront, after the acquire that covers them
__device__ __forceinline__ void regs_park_rows(const RegsParams &p, int b, int dir, int len, int nsteps, int k, float *obuf, int lane_in) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int SP = p.SP, CPR = p.CPR, t0 = k * RG_TT, nt = min(RG_TT, nsteps - t0);
    constexpr int NIT = 5;                               // 16 tokens x CPR <= 18 chunks of 16 bytes over 64 lanes
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int idx = it * 64 + lane;
        const int tok = idx / CPR, c4 = (idx - tok * CPR) * 4;
        v[it] = ld4_agent(regs_other_row(p, b, dir, len, t0 + (tok < nt ? tok : 0)) + c4);
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int idx = it * 64 + lane;
        const int tok = idx / CPR, c4 = (idx - tok * CPR) * 4;
        if (tok < nt) st4(obuf + tok * SP + c4, v[it]);
    }
}

// ---- 16-token tiles: products, matrix-core product, decode ---------------------------------------------------------------
// COOP: all eight wavefronts of the workgroup call it together (workgroup barriers between the phases) and score up to TWO
// tiles (k0, k1; k1 < 0: one) in one pass -- the two tiles share the B fragments and run as independent accumulator chains on
// the matrix cores, and each half of the workgroup decodes one of them.  Else one wavefront alone scores tile k0.
// Same arithmetic either way: per output the k-steps run in ascending state order.
// par0 / par1: the other direction's rows of the tile parked in LDS ([16][SP]) or nullptr (then they come from the stash);
// bpre: COOP only -- the B fragments of column block `wv`, loaded by the caller ahead of time.
// ab: [NTL][16][SPa], scl: [NTL][16][Kc] with NTL = 2 (COOP) / 1.
template <bool COOP>
__device__ __forceinline__ void regs_score_tiles(const RegsParams &p, const int b, const int dir, const int len, const int nsteps,
                                                 const int k0, const int k1, const float *hist, const float *par0, const float *par1,
                                                 float *ab, float *scl, const long long foff, const int wv, const int lane_in,
                                                 const rg_f32x4 (&bpre)[RG_NG]) {
    // Everything per-lane below is derived from this opaque copy: left to itself the compiler hoists the tile's index and
    // address arithmetic out of the callers' tile loops and then spills it (56-448 bytes of scratch per lane, measured)
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const ScoreParams &sp = p.sp;
    constexpr int NWV = COOP ? RG_WAVES : 1;
    constexpr int NWV1 = COOP ? RG_WAVES - 1 : 1;        // wavefronts that form the products: the writer wavefront (it copies its last
                                                         // state rows to the stash meanwhile: the caller) takes none
    constexpr int NTL = COOP ? 2 : 1;
    constexpr int NIT = COOP ? 2 : 5;                    // product items per lane: NTL x 16 tokens x 4 c16 float4 columns, c16 <= 5
    const int c16 = sp.c16, SPa = 16 * c16 + 4, SP = p.SP, K = sp.K, Kc = sp.Kc, ncb = Kc / 16;
    const int G4 = 4 * c16, TI = RG_TT * G4;             // items per tile
    const int lr = lane & 15, lk = lane >> 4;
    const bool two = COOP && k1 >= 0;
#if defined(FARNN_PROBES)
    const bool tprobe = COOP && nsteps == p.L && p.L >= 32 && wv == 0 && lane_in == 0 && (p.dbg & 512);
    long long tq0 = tprobe ? (long long)__builtin_amdgcn_s_memtime() : 0, tq1 = 0, tq2 = 0;
#endif
    // ---- phase 1: ab[tok][s] = a[i+1][s] * bt[i+1][s]; the own direction's rows from LDS (`hist`), the other's from LDS
    // (parked) or the stash.  Stored in the order the matrix cores' A fragments are read: a lane's four k-steps of a state
    // group -- states 16g + 4e + lk, e = 0..3 -- are four consecutive floats (one ds_read_b128 per group instead of four reads)
    {
        float4 oth[NIT];
        int dstv[NIT], ownoff[NIT];
        bool livev[NIT], wrv[NIT];
        const int pw = COOP ? (wv < RG_NWC ? wv : wv - 1) : 0;       // this wavefront among the NWV1
        const bool p1 = !COOP || wv != RG_NWC;
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int idx = p1 ? it * NWV1 * 64 + pw * 64 + lane : NTL * TI;
            const int ti = (NTL == 2 && idx >= TI) ? 1 : 0;
            const int rem = idx - ti * TI;
            const int tok = rem / G4, s4 = (rem - tok * G4) * 4;
            const int kk = ti ? k1 : k0;
            const int t0 = kk * RG_TT, nt = min(RG_TT, nsteps - t0);
            wrv[it] = tok < RG_TT && kk >= 0;
            livev[it] = wrv[it] && tok < nt && s4 < SP;
            const int tokc = livev[it] ? tok : 0, i = (kk >= 0 ? t0 : 0) + tokc;
            const int ai = i + 1, bi = (i + 1 <= len) ? len - (i + 1) : i + 1;
            ownoff[it] = (dir == 0 ? ai : bi) * SP + (s4 < SP ? s4 : 0);
            // states s4 + lk' of group g = s4 / 16, k-step e = (s4 % 16) / 4: position 16 g + 4 lk' + e
            dstv[it] = (ti * RG_TT + (tok < RG_TT ? tok : 0)) * SPa + (s4 & ~15) + ((s4 >> 2) & 3);
            const float *par = ti ? par1 : par0;
            if (par) oth[it] = ld4(par + tokc * SP + (s4 < SP ? s4 : 0));
            else     oth[it] = ld4_agent(regs_other_row(p, b, dir, len, i) + (s4 < SP ? s4 : 0));
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            if (wrv[it]) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (livev[it]) {
                    const float4 own = ld4(hist + ownoff[it]);
                    v = make_float4(own.x * oth[it].x, own.y * oth[it].y, own.z * oth[it].z, own.w * oth[it].w);
                }
                float *dst = ab + dstv[it];
                dst[0] = v.x; dst[4] = v.y; dst[8] = v.z; dst[12] = v.w;
            }
        }
    }
    if (COOP) __syncthreads(); else asm volatile("" ::: "memory");
#if defined(FARNN_PROBES)
    if (tprobe) tq1 = (long long)__builtin_amdgcn_s_memtime();
#endif
    // ---- phase 2: scl[tile][16][Kc] = ab[tile] . O^T on the f32 matrix cores
    {
        const float *arow = ab + lr * SPa + 4 * lk;
        rg_f32x4 bn[RG_NG];                                  // the next column block's fragments, in flight behind the MFMAs
        if (!COOP) regs_load_b(sp, 0, lane, bn);
        for (int cb = wv; cb < ncb; cb += NWV) {
            rg_f32x4 bf[RG_NG];
            if (COOP) {
                if (cb == wv) {
#pragma unroll
                    for (int g = 0; g < RG_NG; g++) bf[g] = bpre[g];
                } else regs_load_b(sp, cb, lane, bf);
            } else {
#pragma unroll
                for (int g = 0; g < RG_NG; g++) bf[g] = bn[g];
                regs_load_b(sp, cb + 1 < ncb ? cb + 1 : cb, lane, bn);
            }
            rg_f32x4 acc0 = rg_f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int g = 0; g < RG_NG; g++) {
                if (g < c16) {
                    const float4 a4 = ld4(arow + 16 * g);
                    float4 c4 = a4;
                    if (two) c4 = ld4(arow + RG_TT * SPa + 16 * g);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, bf[g].x, acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.x, bf[g].x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, bf[g].y, acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.y, bf[g].y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, bf[g].z, acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.z, bf[g].z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, bf[g].w, acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.w, bf[g].w, acc1, 0, 0, 0);
                }
            }
            float *dst = scl + (lk * 4) * Kc + cb * 16 + lr;                 // rows lk*4 + r, column lr of the block
            dst[0] = acc0.x; dst[Kc] = acc0.y; dst[2 * Kc] = acc0.z; dst[3 * Kc] = acc0.w;
            if (two) {
                dst += RG_TT * Kc;
                dst[0] = acc1.x; dst[Kc] = acc1.y; dst[2 * Kc] = acc1.z; dst[3 * Kc] = acc1.w;
            }
        }
    }
    if (COOP) __syncthreads(); else asm volatile("" ::: "memory");
#if defined(FARNN_PROBES)
    if (tprobe) tq2 = (long long)__builtin_amdgcn_s_memtime();
#endif
    // ---- phase 3: four tokens per pass, 16 lanes per token (score_decode.hip.h's decode): priority matrix, `scores`
    // output, threshold clamp, first index of the row maximum, oo -> o_idx.  COOP: wavefronts 0-3 decode tile k0, 4-7 tile k1.
    const int kch = Kc / 64;
    const int clamp_col = K - 1;                         // model_decompose.py:365 / model_onehot.py:166-167
    const int ti3 = COOP ? (wv >> 2) : 0;
    const int kk3 = ti3 ? k1 : k0;
    const int t0 = kk3 * RG_TT, nt = kk3 >= 0 ? min(RG_TT, nsteps - t0) : 0;
    float *sclt = scl + ti3 * RG_TT * Kc;
    for (int tg = COOP ? 4 * (wv & 3) : 0; tg < RG_TT; tg += COOP ? RG_TT : 4) {
        if (tg >= nt) break;
        if (sp.P) {                                      // PriorityLayer: scores @ P (priority.py:20-30), row by row
#pragma unroll 1
            for (int j = 0; j < 4; j++) {
                if (tg + j >= nt) break;
                float *sr = sclt + (tg + j) * Kc;
                float sc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int cc = 0; cc < K; cc++) {
                    const float sv = sr[cc];
                    const float *prow = sp.P + (long long)cc * Kc + lane;
#pragma unroll
                    for (int m = 0; m < 4; m++)
                        if (m < kch) sc[m] = fmaf(sv, prow[64 * m], sc[m]);
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int m = 0; m < 4; m++)
                    if (m < kch) sr[lane + 64 * m] = sc[m];
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        }
        const int j = lane >> 4, c = lane & 15;
        const int tokl = tg + j, i = t0 + tokl;
        const bool live = tokl < nt;
        float v[4][4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < kch) x4 = ld4(sclt + (live ? tokl : 0) * Kc + 64 * m + 4 * c);
            v[m][0] = x4.x; v[m][1] = x4.y; v[m][2] = x4.z; v[m][3] = x4.w;
        }
        if (sp.scores && live) {
            float *so = sp.scores + ((long long)b * p.L + i) * K;
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int col = 64 * m + 4 * c + e;
                    if (m < kch && col < K) so[col] = v[m][e];
                }
        }
        float best = -INFINITY;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int col = 64 * m + 4 * c + e;
                float x = v[m][e] + 0.0f;                // -0.0 -> +0.0 (torch: -0 == +0)
                if (col == clamp_col) x = fminf(x, sp.threshold);
                x = (m < kch && col < K) ? x : -INFINITY;
                v[m][e] = x;
                best = fmaxf(best, x);
            }
        asm volatile("s_nop 1\n\t"
                     "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1"
                     : "+v"(best));
        unsigned first = 0x7fffffffu;                    // this lane's first column that holds the row maximum
#pragma unroll
        for (int m = 3; m >= 0; m--)
#pragma unroll
            for (int e = 3; e >= 0; e--) first = v[m][e] == best ? (unsigned)(64 * m + 4 * c + e) : first;
        asm volatile("s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1"
                     : "+v"(first));
        if (c == 0 && live) {
            const int bi = first >= (unsigned)K ? 0 : (int)first;             // an all-NaN row gives 0 like torch
            const int tag = (bi == K - 1) ? sp.o_idx : bi;
            if (sp.tags) sp.tags[(long long)b * p.L + i] = tag;
            if (sp.flat && i < len) sp.flat[foff + i] = tag;
        }
    }
#if defined(FARNN_PROBES)
    if (tprobe)
        printf("seq %d dir %d tiles %d %d (all wavefronts): products + barrier %lld, matrix cores + barrier %lld, decode %lld\n", b, dir, k0, k1,
               tq1 - tq0, tq2 - tq1, (long long)__builtin_amdgcn_s_memtime() - tq2);
#endif
}

// misc words in LDS
enum { RGM_FOFF = 16,        // where the sequence starts in the flat output
       RGM_MINE = 17,        // tiles this workgroup's scorer did while the chain ran (bit k = tile k)
       RGM_ACQ = 18,         // the other direction's progress covered by this workgroup's latest acquire
       RGM_TODO = 19,
       RGM_PARK = 21 };      // [RG_NOB] tile + 1 whose rows of the other direction are parked in obuf[slot]     // the partial-sum reduction's identity (0.0f / -inf): what a masked read returns      // tiles the eight wavefronts score together next

// FARNN_PROBES (profiling build only): s_memtime stamps of the workgroups of full-length sequences, printed at their end
#if defined(FARNN_PROBES)
#define FARNN_RG_STAMP(i) do { if (probe && lane == 0) stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define FARNN_RG_STAMP(i) do { } while (0)
#endif

template <bool MAXSR, bool SCORE, bool NLX>
__global__ void __launch_bounds__(RG_WAVES * 64, 4)          // 4 waves per SIMD = 128 VGPRs: two workgroups per compute unit
chain_regs_kernel(const RegsParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nthreads = RG_WAVES * 64;
    // ids 2s / 2s + 1: forward chains on the even XCDs, backward chains on the odd ones -- an L2 caches one direction's blocks
    const int item = blockIdx.x;
    const int dir = item & 1, slot = item >> 1;
    const int S = p.S, SP = p.SP, G = p.G, RPG = p.RPG, NP = RG_NWC * G;
    const RegsLds lds = regs_lds(p.L, SP, NP, p.sp.c16, p.sp.Kc, SCORE);
    long long *tokoff = reinterpret_cast<long long *>(smem + lds.tok);     // [nsteps] byte offset of step k's block
    float *part = smem + lds.part, *ol = smem + lds.ol, *hist = smem + lds.hist;
    float *ab = smem + lds.ab, *scl = smem + lds.scl, *obuf = smem + lds.obuf;
    int *misc = reinterpret_cast<int *>(smem + lds.misc);

    int b = p.order ? p.order[slot] : slot;
    if (p.sort) b = select_by_length_rank(p.len, p.B, p.L, folded_rank(slot, p.B), reinterpret_cast<int *>(hist), tid, nthreads);
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;
    const float *hinit = dir == 0 ? p.h0 : p.hT;
#if defined(FARNN_PROBES)
    __shared__ long long stamps[8];
    const bool probe = nsteps == p.L && p.L >= 32;
    if (w == 0) FARNN_RG_STAMP(0);
#endif

    // ---- set-up ----------------------------------------------------------------------------------------------------------
    for (int k = tid; k < nsteps; k += nthreads) {
        const int idx = (dir == 0) ? k : (k < len ? len - 1 - k : k);
        tokoff[k] = (long long)clamp_tok(p.x[(long long)b * p.L + idx], p.V) * p.blk * 4;
    }
    for (int j = tid; j < SP; j += nthreads) ol[j] = (p.o && j < S) ? p.o[j] : 1.0f;
    for (int j = tid; j < (nsteps + 1) * SP; j += nthreads) hist[j] = (j < S) ? hinit[j] : 0.0f;     // row 0; pad columns zero
    if (tid < 32) misc[tid] = tid == RGM_ACQ ? -1 : 0;
    if (tid < 2) part[tid * RG_PART_STRIDE + RG_PART_STRIDE - 1] = MAXSR ? -INFINITY : 0.0f;     // the reduction's identity (masked reads)
    if (tid < 2 * 4 * RG_NWC) part[(tid / (4 * RG_NWC)) * RG_PART_STRIDE + NP * SP + tid % (4 * RG_NWC)] = 0.0f;   // the step flags
    __syncthreads();
    if (w == 0) FARNN_RG_STAMP(1);
#if FARNN_ABLATE & 256                               /* 256 = set-up only */
    return;
#endif

    const float *sflag = part + NP * SP;            // the step flags (buffer 0; buffer 1 is RG_PART_STRIDE floats on)
    float *stash = (dir == 0 ? p.A : p.Bk) + (long long)b * (p.L + 1) * SP;
    const int ntl = (nsteps + RG_TT - 1) / RG_TT;
    int kmid = 0;                                    // tiles kmid.. belong to the forward workgroup's half, the rest to the backward one's
    int pubmax = 0;                                  // the last row of THIS direction that a tile of the OTHER workgroup's half needs
    if (SCORE) {
        for (; kmid < ntl; kmid++) {
            int na, nb;
            regs_tile_need(kmid, len, nsteps, na, nb);
            if (na >= nb) break;
        }
        for (int k = 0; k < ntl; k++) {
            int na, nb;
            regs_tile_need(k, len, nsteps, na, nb);
            const bool others = dir == 0 ? k < kmid : k >= kmid;
            if (others) pubmax = max(pubmax, dir == 0 ? na : nb);
        }
    }

    int wr_next = 0;                                 // (writer wavefront) the next state row to copy to the stash
    if (w < RG_NWC) {
        // =================================================================================================================
        // compute wavefronts
        // =================================================================================================================
        if (nsteps > 0) {
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
                bsrc[u] = 16 * (g * RPG + (u < RPG ? u : 0));     // byte address of lane 4 * (g * RPG + u)
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
                    // where the block of step t + RG_D is: fixed at the step's start, where the wavefront waits for its state exchange
                    // anyway (the once-per-64-steps window reload is an LDS read: its wait must not sit between the partial-sum
                    // store and the loads' issue)
                    if (t + RG_D < nsteps) {
                        if (((t + RG_D) & 63) == 0) FARNN_RG_WINDOW(t + RG_D);
                        FARNN_RG_BASE(t + RG_D, nlo, nhi);
                    }
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
                    const float pre = dir == 0 ? s * my_o : s;                             // (:377-386) / (:393-402)
                    float hn;
                    if (NLX) hn = apply_nl(pre, nl_mode);                                  // tanh, relu-tanh, sigmoid
                    else     hn = nl_relu ? fmaxf(pre, 0.0f) : pre;                        // none / relu: no branch in the step
                    const float hx = my_valid ? (dir == 0 ? hn : hn * my_o) : 0.0f;        // what the next step multiplies with
#if !(FARNN_ABLATE & 64)                             /* 64 = no state store */
                    *hptr = hn;                                                            // row t + 1 of `hist` (or the dump slot)
                    hptr += hstep;
#endif
#if !(FARNN_ABLATE & 4)                              /* 4 = no state exchange (wrong results) */
#pragma unroll
                    for (int u = 0; u < RG_RQ; u++) {
                        const float v = __int_as_float(__builtin_amdgcn_ds_bpermute(bsrc[u], __float_as_int(hx)));
                        hs[u] = u < RPG ? v : 0.0f;
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
            if (!SCORE && probe && w == 0 && lane == 0)
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
        // rows 0 .. wr_next - 1 are stored.  One row per pass; when the chain is done the wavefront goes to the workgroup's
        // meeting point at once: what it has not copied yet it copies after the tiles, before the final publish.
        int published = -1;
        auto copy_row = [&](int rr) {
            const float *src = hist + rr * SP;
            float *dst = stash + (long long)rr * SP;
            if (SCORE) {
                for (int j = 2 * lane; j < SP; j += 2 * WAVE) st2_agent(dst + j, src[j], src[j + 1]);
            } else {
                for (int j = lane; j < SP; j += WAVE) dst[j] = src[j];
            }
        };
        while (wr_next <= nsteps) {
            if (nsteps > 0 && !regs_rows_reached(sflag, lane, wr_next)) { __builtin_amdgcn_s_sleep(1); continue; }
            if (SCORE && nsteps > 0 && wr_next > pubmax && regs_rows_reached(sflag, lane, nsteps)) break;   // the chain is done
            copy_row(wr_next);
            wr_next++;
            // The progress word feeds the other workgroup's tiles: once it covers the last row they need (pubmax), the
            // rest is drained and published once, at the end -- no write-through round trip per row after that.
            if (SCORE && wr_next - 1 < nsteps && published < pubmax &&
                (wr_next - 1 >= pubmax || !regs_rows_reached(sflag, lane, wr_next))) {   // caught up with the chain, or pubmax reached
                published = wr_next - 1;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store of this (the only storing) wavefront has left
                if (lane == 0)
                    __hip_atomic_store(p.prog + (long long)dir * p.B + b, ((unsigned long long)p.epoch << 32) | (unsigned)published,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if (SCORE) {
        // =================================================================================================================
        // scorer wavefront: tiles of this workgroup's half, while the chain runs
        // =================================================================================================================
        {                                                             // where sequence b starts in the flat output (utils.py:153-164)
            int partsum = 0;
            if (p.sp.flat) {
                if (p.sp.offs) partsum = lane == 0 ? (int)p.sp.offs[b] : 0;
                else for (int j = lane; j < b; j += WAVE) partsum += clamp_len(p.len[j], p.L);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) partsum += __shfl_xor(partsum, off, WAVE);
            if (lane == 0) misc[RGM_FOFF] = partsum;
        }
        const long long foff = misc[RGM_FOFF];
        unsigned mine = 0u;
        int acq = -1;                 // the other direction's progress as polled before this workgroup's latest acquire
        const unsigned long long *oprog = p.prog + (long long)(dir ^ 1) * p.B + b;
        const int kfirst = dir == 0 ? kmid : kmid - 1, kstep = dir == 0 ? 1 : -1, klast = dir == 0 ? ntl : -1;
        for (int k = kfirst; k != klast; k += kstep) {
            int na, nb;
            regs_tile_need(k, len, nsteps, na, nb);
            const int need_own = dir == 0 ? na : nb, need_oth = dir == 0 ? nb : na;
            if (nsteps - need_own < p.solo_margin) break;             // the chain ends soon: all eight wavefronts will do it
            while (!regs_rows_reached(sflag, lane, need_own)) __builtin_amdgcn_s_sleep(4);
            if (need_oth > acq) {
                int pr = -1;
                for (;;) {
                    pr = lane == 0 ? regs_read_prog(oprog, p.epoch) : 0;
                    pr = __builtin_amdgcn_readfirstlane(pr);
                    if (pr >= need_oth) break;
                    if (regs_rows_reached(sflag, lane, nsteps)) break;    // our chain is done: no open-ended wait beyond it
                    __builtin_amdgcn_s_sleep(8);
                }
                if (pr < need_oth) break;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // after the poll matched, before the loads
                acq = pr;
            }
            {
                const rg_f32x4 none[RG_NG] = {};
                regs_score_tiles<false>(p, b, dir, len, nsteps, k, -1, hist, nullptr, nullptr, ab, scl, foff, 0, lane, none);
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
                regs_tile_need(k, len, nsteps, na, nb);
                const int need_oth = dir == 0 ? nb : na;
                if (need_oth > acq) {
                    int pr = -1;
                    for (;;) {
                        pr = lane == 0 ? regs_read_prog(oprog, p.epoch) : 0;
                        pr = __builtin_amdgcn_readfirstlane(pr);
                        if (pr >= need_oth) break;
                        if (regs_rows_reached(sflag, lane, nsteps)) break;
                        __builtin_amdgcn_s_sleep(16);
                    }
                    if (pr < need_oth) break;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    acq = pr;
                }
                regs_park_rows(p, b, dir, len, nsteps, k, obuf + slot * RG_TT * SP, lane);
                if (lane == 0) misc[RGM_PARK + slot] = k + 1;
                slot++;
            }
        }
        if (lane == 0) { misc[RGM_MINE] = (int)mine; misc[RGM_ACQ] = acq; }
        FARNN_RG_STAMP(3);
    }
    if constexpr (!SCORE) return;

    // =====================================================================================================================
    // the chain is done.  Before the arrival a workgroup touches only the tiles of its own half: those the other direction's
    // progress (as covered by an acquire) allows are scored now by all eight wavefronts; the arrival word carries the
    // mask of this workgroup's tiles, and the workgroup that finds the other's word there scores whatever neither has.
    // =====================================================================================================================
    rg_f32x4 bpre[RG_NG];                                            // this wavefront's column block of O^T: in flight across the barrier
    regs_load_b(p.sp, w < p.sp.Kc / 16 ? w : 0, lane, bpre);
    const unsigned all_tiles = ntl >= 32 ? ~0u : ((1u << ntl) - 1u);
    unsigned promised = 0u;
    if (w == RG_WAVES - 1) {
        // which tiles of this half can be scored now: those the other direction's progress, as covered by an acquire, allows
        const unsigned mine = (unsigned)misc[RGM_MINE];
        int acq = misc[RGM_ACQ];
        const unsigned own_half = dir == 0 ? (all_tiles & ~((1u << kmid) - 1u)) : (all_tiles & ((1u << kmid) - 1u));
        const unsigned long long *oprog = p.prog + (long long)(dir ^ 1) * p.B + b;
        unsigned todo = 0u;
        for (int it = 0;; it++) {
            int miss = 0;
            todo = 0u;
            for (int k = 0; k < ntl; k++) {
                if (!((own_half & ~mine) >> k & 1u)) continue;
                int na, nb;
                regs_tile_need(k, len, nsteps, na, nb);
                if ((dir == 0 ? nb : na) <= acq) todo |= 1u << k; else miss++;
            }
            if (!miss || it >= p.spin) break;                         // bounded: what stays open goes to the second arrival
            int pr = lane == 0 ? regs_read_prog(oprog, p.epoch) : 0;
            pr = __builtin_amdgcn_readfirstlane(pr);
            if (pr > acq) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the invalidate has completed before the barrier below
                acq = pr;
            } else __builtin_amdgcn_s_sleep(8);
        }
        promised = mine | todo;
        if (lane == 0) misc[RGM_TODO] = (int)todo;
    }
    wg_barrier_lds();                                                // the chain is done, the mask is there (LDS only: this wavefront's
                                                                     // O^T loads and the writer's last stores stay in flight)
    if (w == 0) FARNN_RG_STAMP(4);
    if (w == RG_NWC)                                                 // the state rows the writer had not copied yet: issued now, landed by the
        for (; wr_next <= nsteps; wr_next++)                         // time the tiles are done (this wavefront forms no products meanwhile)
            for (int j = 2 * lane; j < SP; j += 2 * WAVE)
                st2_agent(stash + (long long)wr_next * SP + j, hist[wr_next * SP + j], hist[wr_next * SP + j + 1]);
    // ---- arrival: ONE lane exchanges the sequence's arrival word for {epoch, the tiles this workgroup scores}; the
    // exchange is in flight while those tiles are scored.  (The word says nothing about this workgroup's stash rows: they
    // are published through the progress word, below; the workgroup that has to read them waits for that.)
    unsigned long long arrived = 0ull;
    if (w == RG_WAVES - 1 && lane == 0)
        arrived = __hip_atomic_exchange(p.arr + b, ((unsigned long long)p.epoch << 32) | 0x80000000ull | promised,
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long foff = misc[RGM_FOFF];
    for (int pass = 0; pass < 2; pass++) {
        unsigned todo = (unsigned)misc[RGM_TODO];
        while (todo) {                                               // two tiles per pass
            const int k0 = __builtin_ctz(todo);
            todo &= todo - 1u;
            const int k1 = todo ? __builtin_ctz(todo) : -1;
            if (todo) todo &= todo - 1u;
            const float *par0 = nullptr, *par1 = nullptr;
#pragma unroll
            for (int sl = 0; sl < RG_NOB; sl++) {
                if (misc[RGM_PARK + sl] == k0 + 1) par0 = obuf + sl * RG_TT * SP;
                if (misc[RGM_PARK + sl] == k1 + 1) par1 = obuf + sl * RG_TT * SP;
            }
            regs_score_tiles<true>(p, b, dir, len, nsteps, k0, k1, hist, par0, par1, ab, scl, foff, w, lane, bpre);
            __syncthreads();                                         // the tiles' LDS is free again
        }
        if (pass == 1) break;
        if (w == 0) FARNN_RG_STAMP(5);
        wg_barrier_lds();                                            // every wavefront has read this pass's mask (it is rewritten below)
        // every stash row of this direction has been stored by the writer wavefront: it drains them (they had the tiles'
        // time to land) and publishes the full count
        if (w == RG_NWC) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(p.prog + (long long)dir * p.B + b, ((unsigned long long)p.epoch << 32) | (unsigned)nsteps,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (w == RG_WAVES - 1) {
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(arrived >> 32));
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)arrived);
            unsigned rest = 0u;
            if (hi == p.epoch && (lo & 0x80000000u)) {               // second of the two
                rest = all_tiles & ~(promised | (lo & 0x7fffffffu));
                if (rest) {
                    // The other workgroup has arrived: it is resident, past its chain, and publishes its full row count
                    // after a bounded amount of work of its own (it waits for nobody) -- so this wait ends.
                    const unsigned long long *oprog = p.prog + (long long)(dir ^ 1) * p.B + b;
                    for (;;) {
                        int pr = lane == 0 ? regs_read_prog(oprog, p.epoch) : 0;
                        pr = __builtin_amdgcn_readfirstlane(pr);
                        if (pr >= nsteps) break;
                        __builtin_amdgcn_s_sleep(8);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            if (lane == 0) misc[RGM_TODO] = (int)rest;
        }
        __syncthreads();
        if (misc[RGM_TODO] == 0) break;
    }
#if defined(FARNN_PROBES)
    if (probe && tid == 0) {
        const long long e = (long long)__builtin_amdgcn_s_memtime();
        printf("seq %d dir %d: setup %lld, chain %lld (%lld per step), scorer alone until +%lld, all waves meet +%lld, tiles together %lld, "
               "arrival + sweep %lld; tiles alone %d of %d\n", b, dir, stamps[1] - stamps[0], stamps[2] - stamps[1],
               (stamps[2] - stamps[1]) / nsteps, stamps[3] - stamps[2], stamps[4] - stamps[2], stamps[5] - stamps[4], e - stamps[5],
               __popc((unsigned)misc[RGM_MINE]), ntl);
    }
#endif
}

}  // namespace farnn
