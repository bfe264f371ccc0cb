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
// Nobody waits for a workgroup that may not be resident: every wait on the other direction is bounded, tiles are claimed
// by an atomic exchange of the launch's epoch (exactly once), and each workgroup ends with an exchange on the sequence's
// arrival word: the one that finds the other's entry there sweeps up every tile nobody claimed -- by then both stashes are
// complete and published.  All words carry the launch's epoch: nothing is reset between launches and an aborted launch
// leaves nothing behind that a later one could mistake for its own.
#pragma once
#include "common.hip.h"
#include "score_params.hip.h"
#include "launch_order.hip.h"
#include "chain_regs_params.hip.h"

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
// steps every compute wavefront has finished, as the polling wavefront sees them (lane i reads wavefront i's flag)
__device__ __forceinline__ bool regs_flags_reached(const int *flags, int lane, int target) {
    const int v = lds_flag_get(flags + (lane < RG_NWC ? lane : 0));
    return __ballot(v < target) == 0ull;
}
__device__ __forceinline__ int regs_flags_min(const int *flags, int lane) {
    int v = lds_flag_get(flags + (lane < RG_NWC ? lane : 0));
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) v = min(v, __shfl_xor(v, off, WAVE));
    return __builtin_amdgcn_readfirstlane(v);
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

// ---- one 16-token tile: products, matrix-core product, decode ------------------------------------------------------------
// COOP: all eight wavefronts of the workgroup call it together (workgroup barriers between the phases); else one wavefront
// alone.  Same arithmetic either way: per output the k-steps run in ascending state order.
template <bool COOP>
__device__ __forceinline__ void regs_score_tile(const RegsParams &p, const int b, const int dir, const int len, const int nsteps,
                                                const int k, const float *hist, float *ab, float *scl, const long long foff,
                                                const int wv, const int lane) {
    const ScoreParams &sp = p.sp;
    constexpr int NWV = COOP ? RG_WAVES : 1;
    constexpr int CH = COOP ? 1 : 4;                     // column blocks a wavefront runs side by side (shared A fragments)
    constexpr int NIT = COOP ? 1 : 5;                    // product items per lane: 16 tokens x 4 c16 float4 columns, c16 <= 5
    const int c16 = sp.c16, SPa = 16 * c16 + 4, SP = p.SP, K = sp.K, Kc = sp.Kc, ncb = Kc / 16;
    const int t0 = k * RG_TT;
    const int nt = min(RG_TT, nsteps - t0);
    const float *Ab = p.A + (long long)b * (p.L + 1) * SP;
    const float *Bb = p.Bk + (long long)b * (p.L + 1) * SP;
    const int G4 = 4 * c16;
    // ---- phase 1: ab[tok][s] = a[i+1][s] * bt[i+1][s]; the own direction's rows from LDS, the other's from the stash
    {
        float4 oth[NIT];
        int tokv[NIT], s4v[NIT], ownrow[NIT];
        bool livev[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int idx = it * NWV * 64 + wv * 64 + lane;
            const int tok = idx / G4, s4 = (idx - tok * G4) * 4;
            tokv[it] = tok; s4v[it] = s4;
            livev[it] = tok < nt && s4 < SP;
            const int i = t0 + (tok < nt ? tok : 0);
            const int ai = i + 1, bi = (i + 1 <= len) ? len - (i + 1) : i + 1;
            ownrow[it] = dir == 0 ? ai : bi;
            const float *src = (dir == 0 ? Bb + (long long)bi * SP : Ab + (long long)ai * SP) + (s4 < SP ? s4 : 0);
            oth[it] = ld4_agent(src);
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            if (tokv[it] < RG_TT) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (livev[it]) {
                    const float4 own = ld4(hist + ownrow[it] * SP + s4v[it]);
                    v = make_float4(own.x * oth[it].x, own.y * oth[it].y, own.z * oth[it].z, own.w * oth[it].w);
                }
                st4(ab + tokv[it] * SPa + s4v[it], v);
            }
        }
    }
    if (COOP) __syncthreads(); else asm volatile("" ::: "memory");
    // ---- phase 2: scl[16][Kc] = ab . O^T on the f32 matrix cores; B fragments from the matrix-core image of O^T in L2
    {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int lr = lane & 15, lk = lane >> 4;
        const float *arow = ab + lr * SPa + lk;
        const f32x4 *otm = reinterpret_cast<const f32x4 *>(sp.OTm) + lane;
        for (int cb0 = wv * CH; cb0 < ncb; cb0 += NWV * CH) {
            f32x4 acc[CH], bc[CH], bn[CH];
#pragma unroll
            for (int q = 0; q < CH; q++) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto load_b = [&](int g, f32x4 (&dst)[CH]) {
                const int gc = g < c16 ? g : c16 - 1;
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const int cb = cb0 + q < ncb ? cb0 + q : ncb - 1;
                    dst[q] = otm[((long long)cb * c16 + gc) * 64];
                }
            };
            load_b(0, bc);
#pragma unroll 1
            for (int g = 0; g < c16; g++) {
                load_b(g + 1, bn);
                const float *ap = arow + 16 * g;
                const float a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
#pragma unroll
                for (int q = 0; q < CH; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bc[q].x, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < CH; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bc[q].y, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < CH; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bc[q].z, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < CH; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, bc[q].w, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < CH; q++) bc[q] = bn[q];
            }
#pragma unroll
            for (int q = 0; q < CH; q++) {
                if (cb0 + q < ncb) {
                    float *dst = scl + (lk * 4) * Kc + (cb0 + q) * 16 + lr;       // rows lk*4 + r, column lr of the block
                    dst[0] = acc[q].x; dst[Kc] = acc[q].y; dst[2 * Kc] = acc[q].z; dst[3 * Kc] = acc[q].w;
                }
            }
        }
    }
    if (COOP) __syncthreads(); else asm volatile("" ::: "memory");
    // ---- phase 3: four tokens per pass, 16 lanes per token (score_decode.hip.h's decode): priority matrix, `scores`
    // output, threshold clamp, first index of the row maximum, oo -> o_idx
    const int kch = Kc / 64;
    const int clamp_col = K - 1;                         // model_decompose.py:365 / model_onehot.py:166-167
    for (int tg = COOP ? 4 * wv : 0; tg < RG_TT; tg += COOP ? RG_TT : 4) {
        if (tg >= nt) break;
        if (sp.P) {                                      // PriorityLayer: scores @ P (priority.py:20-30), row by row
#pragma unroll 1
            for (int j = 0; j < 4; j++) {
                if (tg + j >= nt) break;
                float *sr = scl + (tg + j) * Kc;
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
            if (m < kch) x4 = ld4(scl + (live ? tokl : 0) * Kc + 64 * m + 4 * c);
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
}

// misc words in LDS
enum { RGM_SFLAG = 0,        // [6] partial sums of step t written: t + 1
       RGM_DFLAG = 8,        // [6] state row t + 1 written:        t + 1
       RGM_FOFF = 16,        // where the sequence starts in the flat output
       RGM_MINE_LO = 17, RGM_MINE_HI = 18,     // tiles this workgroup's scorer took while the chain ran
       RGM_COOP_K = 19 };    // the tile the workgroup scores next, or -1

// FARNN_PROBES (profiling build only): s_memtime stamps of the workgroups of full-length sequences, printed at their end
#if defined(FARNN_PROBES)
#define FARNN_RG_STAMP(i) do { if (probe && lane == 0) stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define FARNN_RG_STAMP(i) do { } while (0)
#endif

template <bool MAXSR, bool SCORE>
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
    float *hp = smem + lds.hp, *part = smem + lds.part, *ol = smem + lds.ol, *hist = smem + lds.hist;
    float *ab = smem + lds.ab, *scl = smem + lds.scl;
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
    for (int idx = tid; idx < NP * 4; idx += nthreads) {
        const int gi = idx >> 2, ii = idx & 3, row = gi * RPG + ii;
        float v = 0.0f;
        if (ii < RPG && row < S) {
            v = hinit[row];
            if (dir == 1 && p.o) v *= p.o[row];                       // the backward chain's input is pre-scaled (:393)
        }
        hp[idx] = v;
    }
    for (int j = tid; j < SP; j += nthreads) ol[j] = (p.o && j < S) ? p.o[j] : 1.0f;
    for (int j = tid; j < (nsteps + 1) * SP; j += nthreads) hist[j] = (j < S) ? hinit[j] : 0.0f;     // row 0; pad columns zero
    if (tid < 32) misc[tid] = tid == RGM_COOP_K ? -1 : 0;
    __syncthreads();
    if (w == 0) FARNN_RG_STAMP(1);

    int *sflag = misc + RGM_SFLAG, *dflag = misc + RGM_DFLAG;
    float *stash = (dir == 0 ? p.A : p.Bk) + (long long)b * (p.L + 1) * SP;
    const int ntl = (nsteps + RG_TT - 1) / RG_TT;
    int kmid = 0;                                    // tiles kmid.. belong to the forward workgroup's half, the rest to the backward one's
    if (SCORE) {
        for (; kmid < ntl; kmid++) {
            int na, nb;
            regs_tile_need(kmid, len, nsteps, na, nb);
            if (na >= nb) break;
        }
    }

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
            const float *myhp = hp + gid * 4;
            const int rows_w = G * RPG;
            const int rj = lane >> 2, rs = lane & 3;              // reduce: four lanes per row
            const int my_row = w * rows_w + rj;
            const bool my_valid = rj < rows_w && my_row < S;
            const bool my_writer = my_valid && rs == 0;
            const int my_hp = my_valid ? (w * G + rj / RPG) * 4 + rj % RPG : 0;
            const int rowc = my_valid ? my_row : 0;
            const float my_o = my_valid ? ol[my_row] : 1.0f;
            const int nl_mode = p.nl;
            const float ninf = -INFINITY;
            bool okrow[RG_RQ];
#pragma unroll
            for (int u = 0; u < RG_RQ; u++) okrow[u] = u < RPG && row0 + u < S;
            constexpr int NQI = (RG_NWC * RG_MAXG + 3) / 4;      // partial vectors per reducing lane, at most
            int qoff[NQI];                                       // this lane's partial vectors (clamped: the mask below drops the repeats)
#pragma unroll
            for (int i = 0; i < NQI; i++) qoff[i] = ((rs + 4 * i < NP) ? rs + 4 * i : rs) * SP + rowc;

            // The ring: RG_D steps x RG_RQ rows of 16 bytes per lane, loaded by inline asm so that NO compiler wait ever
            // drains it (hipcc's own bookkeeping merges the loop's back edge into vmcnt(0): measured, the ring then has no depth).
            // An asm load's destination counts as written at the statement: every consumer sits behind a wait statement
            // that names the four registers "+v" (cdna_hip_programming.md 5.7, form ii).  Loads retire in issue order, so
            // step t's four pieces have landed once at most the pieces of the steps issued after it are outstanding.
            v4f r[RG_D][RG_RQ];
#define FARNN_RG_ISSUE(d, t_)                                                                  \
            do {                                                                               \
                const long long off_ = tokoff[t_];                                             \
                const unsigned lo_ = __builtin_amdgcn_readfirstlane((unsigned)off_);           \
                const unsigned hi_ = __builtin_amdgcn_readfirstlane((unsigned)(off_ >> 32));  \
                const char *bp_ = Mbase + (((long long)hi_ << 32) | lo_);                      \
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
#pragma unroll
            for (int d = 0; d < RG_D; d++) {
#pragma unroll
                for (int u = 0; u < RG_RQ; u++) r[d][u] = v4f{0.f, 0.f, 0.f, 0.f};
                if (d < nsteps) FARNN_RG_ISSUE(d, d);
            }
            int pb = 0;
            for (int t0 = 0; t0 < nsteps; t0 += RG_D) {
#pragma unroll
                for (int d = 0; d < RG_D; d++) {
                    const int t = t0 + d;
                    if (t >= nsteps) break;
                    const float4 h4 = ld4(myhp);
                    const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
                    static_assert(RG_D == 4 && RG_RQ == 4, "FARNN_RG_WAIT is written out for a 4 x 4 ring");
                    FARNN_RG_WAIT(d, nsteps - 1 - t);            // steps issued after this one: min(RG_D - 1, nsteps - 1 - t)
                    v4f acc = MAXSR ? v4f{ninf, ninf, ninf, ninf} : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < RG_RQ; u++) {
                        if (MAXSR) {
                            acc.x = fmaxf(acc.x, okrow[u] ? hv[u] * r[d][u].x : ninf);
                            acc.y = fmaxf(acc.y, okrow[u] ? hv[u] * r[d][u].y : ninf);
                            acc.z = fmaxf(acc.z, okrow[u] ? hv[u] * r[d][u].z : ninf);
                            acc.w = fmaxf(acc.w, okrow[u] ? hv[u] * r[d][u].w : ninf);
                        } else {
                            acc.x = fmaf(hv[u], r[d][u].x, acc.x);
                            acc.y = fmaf(hv[u], r[d][u].y, acc.y);
                            acc.z = fmaf(hv[u], r[d][u].z, acc.z);
                            acc.w = fmaf(hv[u], r[d][u].w, acc.w);
                        }
                    }
                    float *pp = part + pb * NP * SP;
                    if (active) *reinterpret_cast<v4f *>(pp + gid * SP + c * 4) = acc;
                    if (lane == 0) lds_flag_set(sflag + w, t + 1);
                    asm volatile("" : "+v"(acc));                // the slot's registers are dead from here: reload them
                    if (t + RG_D < nsteps) FARNN_RG_ISSUE(d, t + RG_D);
                    while (!regs_flags_reached(sflag, lane, t + 1)) {}
                    // reduce the rows this wavefront consumes next: four lanes per row, each a quarter of the partial vectors
                    float pv[NQI];
#pragma unroll
                    for (int i = 0; i < NQI; i++) pv[i] = pp[qoff[i]];
                    float s = MAXSR ? ninf : 0.0f;
#pragma unroll
                    for (int i = 0; i < NQI; i++) {
                        const float v = (i == 0 || rs + 4 * i < NP) ? pv[i] : (MAXSR ? ninf : 0.0f);
                        s = MAXSR ? fmaxf(s, v) : s + v;
                    }
                    s = MAXSR ? quad_max(s) : quad_sum(s);
                    if (my_writer) {
                        const float pre = dir == 0 ? s * my_o : s;                         // (:377-386) / (:393-402)
                        const float hn = nl_mode == FARNN_NL_NONE ? pre : (nl_mode == FARNN_NL_RELU ? fmaxf(pre, 0.0f) : apply_nl(pre, nl_mode));
                        hist[(t + 1) * SP + my_row] = hn;
                        hp[my_hp] = dir == 0 ? hn : hn * my_o;
                    }
                    if (lane == 0) lds_flag_set(dflag + w, t + 1);
                    pb ^= 1;
                }
            }
#undef FARNN_RG_ISSUE
#undef FARNN_RG_WAIT
            __builtin_amdgcn_s_setprio(0);
            if (w == 0) FARNN_RG_STAMP(2);
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
        int next = 0;                                             // rows 0 .. next - 1 are stored
        while (next <= nsteps) {
            const int avail = nsteps > 0 ? regs_flags_min(dflag, lane) : 0;      // rows 0 .. avail are complete in hist
            if (avail < next) { __builtin_amdgcn_s_sleep(2); continue; }
            for (int rr = next; rr <= avail; rr++) {
                const float *src = hist + rr * SP;
                float *dst = stash + (long long)rr * SP;
                if (SCORE) {
                    for (int j = 2 * lane; j < SP; j += 2 * WAVE) st2_agent(dst + j, src[j], src[j + 1]);
                } else {
                    for (int j = lane; j < SP; j += WAVE) dst[j] = src[j];
                }
            }
            next = avail + 1;
            if (SCORE) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store of this (the only storing) wavefront has left
                if (lane == 0)
                    __hip_atomic_store(p.prog + (long long)dir * p.B + b, ((unsigned long long)p.epoch << 32) | (unsigned)avail,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (SCORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
        unsigned long long mine = 0ull;
        const unsigned long long *oprog = p.prog + (long long)(dir ^ 1) * p.B + b;
        const int kfirst = dir == 0 ? kmid : kmid - 1, kstep = dir == 0 ? 1 : -1, klast = dir == 0 ? ntl : -1;
        for (int k = kfirst; k != klast; k += kstep) {
            int na, nb;
            regs_tile_need(k, len, nsteps, na, nb);
            const int need_own = dir == 0 ? na : nb, need_oth = dir == 0 ? nb : na;
            if (nsteps - need_own < p.solo_margin) break;             // the chain ends soon: all eight wavefronts will do it
            while (!regs_flags_reached(dflag, lane, need_own)) __builtin_amdgcn_s_sleep(4);
            bool ready = false;
            for (;;) {
                int pr = lane == 0 ? regs_read_prog(oprog, p.epoch) : 0;
                pr = __builtin_amdgcn_readfirstlane(pr);
                if (pr >= need_oth) { ready = true; break; }
                if (regs_flags_reached(dflag, lane, nsteps)) break;   // our chain is done: no open-ended wait beyond it
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ready) break;
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_exchange(p.claim + (long long)b * p.NT + k, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == p.epoch) continue;                             // the other workgroup took it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // after the poll matched, before the loads
            regs_score_tile<false>(p, b, dir, len, nsteps, k, hist, ab, scl, foff, 0, lane);
            mine |= 1ull << k;
        }
        if (lane == 0) { misc[RGM_MINE_LO] = (int)(unsigned)mine; misc[RGM_MINE_HI] = (int)(unsigned)(mine >> 32); }
        FARNN_RG_STAMP(3);
    }
    if constexpr (!SCORE) return;

    // =====================================================================================================================
    // the chain is done: all eight wavefronts score what is left, then the arrival
    // =====================================================================================================================
    __syncthreads();
    if (w == 0) FARNN_RG_STAMP(4);
    const long long foff = misc[RGM_FOFF];
    unsigned long long mine = ((unsigned long long)(unsigned)misc[RGM_MINE_HI] << 32) | (unsigned)misc[RGM_MINE_LO];
    const unsigned long long all_tiles = ntl >= 64 ? ~0ull : ((1ull << ntl) - 1ull);
    const unsigned long long *oprog = p.prog + (long long)(dir ^ 1) * p.B + b;
    unsigned *claims = p.claim + (long long)b * p.NT;
    int other_tiles = -1;                                            // >= 0 once this workgroup is the second arrival
    // the scorer wavefront decides (polls, claims, acquires); the decision reaches the others through LDS + barrier
    for (int pass = 0; pass < 2; pass++) {                           // 0: before the arrival; 1: the second arrival's sweep
        unsigned long long taken = 0ull;                             // tiles somebody has claimed, as of the snapshot
        if (w == RG_WAVES - 1) {
            unsigned cv = 0;
            if (lane < ntl) cv = __hip_atomic_load(claims + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            taken = __ballot(lane < ntl && cv == p.epoch);
        }
        for (int ord = 0; ord < ntl; ord++) {
            if (w == RG_WAVES - 1) {
                // this workgroup's half first (forward: kmid upwards, backward: kmid - 1 downwards), then the other half
                int k;
                if (dir == 0) k = ord < ntl - kmid ? kmid + ord : ntl - 1 - ord;
                else          k = ord < kmid ? kmid - 1 - ord : ord;
                int go = -1;
                if (!((taken >> k) & 1ull)) {
                    bool ready = pass == 1;                          // the sweep: both stashes are complete and published
                    if (!ready) {
                        int na, nb;
                        regs_tile_need(k, len, nsteps, na, nb);
                        const int need_oth = dir == 0 ? nb : na;
                        const bool own_half = dir == 0 ? k >= kmid : k < kmid;
                        const int polls = own_half ? p.spin : 1;
                        for (int it = 0; it < polls && !ready; it++) {
                            int pr = lane == 0 ? regs_read_prog(oprog, p.epoch) : 0;
                            pr = __builtin_amdgcn_readfirstlane(pr);
                            ready = pr >= need_oth;
                            if (!ready && it + 1 < polls) __builtin_amdgcn_s_sleep(8);
                        }
                    }
                    if (ready) {
                        unsigned old = 0;
                        if (lane == 0) old = __hip_atomic_exchange(claims + k, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        old = __builtin_amdgcn_readfirstlane(old);
                        if (old != p.epoch) {
                            go = k;
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the invalidate has completed before the barrier
                        }
                    }
                }
                if (lane == 0) misc[RGM_COOP_K] = go;
            }
            __syncthreads();
            const int k = misc[RGM_COOP_K];
            if (k >= 0) {
                regs_score_tile<true>(p, b, dir, len, nsteps, k, hist, ab, scl, foff, w, lane);
                mine |= 1ull << k;
            }
            __syncthreads();                                         // the tile's LDS and the decision word are free again
        }
        if (pass == 1) break;
        if (w == 0) FARNN_RG_STAMP(5);
        // ---- arrival: this workgroup's stash rows are stored and drained (the writer), its tiles are done
        if (w == RG_WAVES - 1) {
            unsigned long long old = 0ull;
            const unsigned long long me = ((unsigned long long)p.epoch << 32) | 0x80000000ull | (unsigned)__popcll(mine);
            if (lane == 0) old = __hip_atomic_exchange(p.arr + b, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(old >> 32));
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)old);
            int ot = -1;
            if (hi == p.epoch && (lo & 0x80000000u)) ot = (int)(lo & 0x7fffffffu);
            if (lane == 0) misc[RGM_COOP_K] = ot;
        }
        __syncthreads();
        other_tiles = misc[RGM_COOP_K];
        __syncthreads();
        if (other_tiles < 0) break;                                  // first of the two: the other one sweeps
        if (other_tiles + __popcll(mine) >= ntl) break;              // everything was scored
        (void)all_tiles;
    }
#if defined(FARNN_PROBES)
    if (probe && tid == 0) {
        const long long e = (long long)__builtin_amdgcn_s_memtime();
        printf("seq %d dir %d: setup %lld, chain %lld (%lld per step), scorer alone until +%lld, all waves meet +%lld, tiles together %lld, "
               "arrival + sweep %lld; tiles alone %d of %d, second arrival %d\n", b, dir, stamps[1] - stamps[0], stamps[2] - stamps[1],
               (stamps[2] - stamps[1]) / nsteps, stamps[3] - stamps[2], stamps[4] - stamps[2], stamps[5] - stamps[4], e - stamps[5],
               __popcll(((unsigned long long)(unsigned)misc[RGM_MINE_HI] << 32) | (unsigned)misc[RGM_MINE_LO]), ntl, other_tiles >= 0);
    }
#endif
}

}  // namespace farnn
