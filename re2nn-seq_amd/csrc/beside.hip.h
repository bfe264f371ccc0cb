// The scores + threshold/argmax decode (K2) run BESIDE a recurrence kernel: shared by chain_regs.hip.h (dense blocks from registers)
// and decomp_regs.hip.h (decomposed model, packed rows in registers).  What a host kernel provides: its own direction's states in
// LDS (`hist`, [nsteps + 1][SP]), its stash rows published through the sequence's progress word (write-through stores, drained,
// then the word: the producer half of cdna_hip_programming.md Guideline 16's recipe R1), and the LDS areas below.  What this
// header does: tiles of 16 tokens -- products of the two directions' states, [16 x S].[S x K] on the f32 matrix cores, decode --
// by one wavefront alone or by the whole workgroup (two tiles per pass), and the end-of-chain protocol between the two
// workgroups of a sequence (own half, arrival word with the mask of scored tiles, the second arrival sweeps the rest).
// chain_regs.hip.h's header has the full account of the hand-off.
#pragma once
#include "common.hip.h"
#include "score_params.hip.h"

namespace farnn {

// the scorer wavefront takes the label-map path (label_map.hip.h: token by token behind the chain) when the output matrix is a
// label map and the call wants tags only; the matrix form (tiles on the f32 matrix cores) otherwise
__host__ __device__ __forceinline__ bool bs_label_map_path(const ScoreParams &sp) { return sp.lm.on && !sp.P && !sp.scores; }

// what tile k needs: forward rows 0..needA and backward rows 0..needB stored (a row = one state, row 0 the initial one)
__device__ __forceinline__ void bs_tile_need(int k, int len, int nsteps, int &needA, int &needB) {
    const int lo = k * RG_TT;
    const int hi = min(lo + RG_TT, nsteps) - 1;
    needA = hi + 1;                                   // alpha of token i is row i + 1
    int nb = 0;                                       // beta of token i is row len - (i + 1); pads of FULL mode: row i + 1
    if (lo < len) nb = len - lo - 1;
    if (hi >= len) nb = max(nb, hi + 1);
    needB = nb;
}

// The launch's epoch, from device memory, for ANY sequence of batch sizes.  `done` is a 64-bit counter to which the second workgroup
// to arrive of every sequence adds once (bs_finish): 1, except for sequence 0, which adds BS_EPOCH_SPAN - (B - 1) -- a launch adds
// exactly BS_EPOCH_SPAN in all, whatever its B, and any proper subset of its adds less than that.  Between launches the counter is a
// multiple of the span; during launch e it runs from e SPAN towards (e + 1) SPAN and reaches that only with the launch's LAST second
// arrival -- i.e. after every workgroup of the launch has started and read it.  So done >> BS_EPOCH_SHIFT is the same number for all
// workgroups of a launch however late they start, and one more for the next launch: no kernel argument, no host-side counter, no
// reset when the batch size changes, and launches captured into HIP graphs at different B replay and interleave with eager calls
// (stream-ordered) with the right epoch.  (Round 4 divided a count of sequences by B: valid for one batch size only -- a replayed
// graph behind an eager call of another B read an epoch that changed mid-launch.)  Never 0: zeroed words carry 0.
// ONE lane of the wavefronts that use the epoch reads the counter (all of a launch's workgroups start together: thousands of
// L1-bypassing loads of one line at once cost the launch microseconds -- measured, +4 us on the 35 us headline step).
constexpr int BS_EPOCH_SHIFT = 24;                              // B <= 2^24 sequences per launch; 2^40 launches per handle
constexpr unsigned long long BS_EPOCH_SPAN = 1ull << BS_EPOCH_SHIFT;
__device__ __forceinline__ unsigned bs_launch_epoch(const unsigned long long *done, int lane) {
    unsigned long long d = 0ull;
    if (lane == 0) d = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)d), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(d >> 32));
    const unsigned long long q = (((unsigned long long)hi << 32) | lo) >> BS_EPOCH_SHIFT;
    return (unsigned)(q % 0xffffffffull) + 1u;
}
// what the second arrival of sequence b adds to the counter
__device__ __forceinline__ unsigned long long bs_epoch_add(int b, int B) { return b == 0 ? BS_EPOCH_SPAN - (unsigned long long)(B - 1) : 1ull; }

__device__ __forceinline__ int bs_read_prog(const unsigned long long *w, unsigned epoch) {
    const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)(v >> 32) == epoch ? (int)(unsigned)v : -1;
}

typedef float rg_f32x4 __attribute__((ext_vector_type(4)));

// B fragments (matrix-core image of O^T, score_decode.hip.h) of column block cb, all state groups: one round trip to L2
template <int NG>
__device__ __forceinline__ void bs_load_b(const ScoreParams &sp, int cb, int lane, rg_f32x4 (&bf)[NG]) {
    const rg_f32x4 *otm = reinterpret_cast<const rg_f32x4 *>(sp.OTm);
    const int c16 = sp.c16;
#pragma unroll
    for (int g = 0; g < NG; g++) bf[g] = (otm + ((long long)cb * c16 + (g < c16 ? g : c16 - 1)) * 64)[lane];
}

// the other direction's row that token i of sequence b multiplies with (the stash row index and its base)
__device__ __forceinline__ const float *bs_other_row(const BesideParams &p, int b, int dir, int len, int i) {
    const int ai = i + 1, bi = (i + 1 <= len) ? len - (i + 1) : i + 1;
    const long long base = (long long)b * (p.L + 1) * p.SP;
    return dir == 0 ? p.Bk + base + (long long)bi * p.SP : p.A + base + (long long)ai * p.SP;
}

// park the other direction's rows of tile k in LDS (obuf[16][SP]); one wavefront, after the acquire that covers them
// (NB: loads in flight per batch -- NG of them, or fewer batches' worth where registers are short)
template <int NG, int NB = NG>
__device__ __forceinline__ void bs_park_rows(const BesideParams &p, int b, int dir, int len, int nsteps, int k, float *obuf, int lane_in) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int SP = p.SP, CPR = p.CPR, t0 = k * RG_TT, nt = min(RG_TT, nsteps - t0);
    constexpr int NIT = NG;                              // 16 tokens x CPR <= 4 NG chunks of 16 bytes over 64 lanes
    static_assert(NIT % NB == 0, "whole batches");
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += NB) {
        float4 v[NB];
#pragma unroll
        for (int it = 0; it < NB; it++) {
            const int idx = (it0 + it) * 64 + lane;
            const int tok = idx / CPR, c4 = (idx - tok * CPR) * 4;
            v[it] = ld4_agent(bs_other_row(p, b, dir, len, t0 + (tok < nt ? tok : 0)) + c4);
        }
#pragma unroll
        for (int it = 0; it < NB; it++) {
            const int idx = (it0 + it) * 64 + lane;
            const int tok = idx / CPR, c4 = (idx - tok * CPR) * 4;
            if (tok < nt) st4(obuf + tok * SP + c4, v[it]);
        }
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
// NWV wavefronts cooperate (COOP); wavefront WSKIP of them, if any (>= 0), forms no products (the host kernel's writer: it copies
// its last state rows meanwhile); NG = state groups of 16 the instantiation reaches (S <= 16 NG).
template <bool COOP, int NWV_, int NG, int WSKIP>
__device__ __forceinline__ void bs_score_tiles(const BesideParams &p, const int b, const int dir, const int len, const int nsteps,
                                                 const int k0, const int k1, const float *hist, const float *par0, const float *par1,
                                                 float *ab, float *scl, const long long foff, const int wv, const int lane_in,
                                                 const rg_f32x4 (&bpre)[NG]) {
    // Everything per-lane below is derived from this opaque copy: left to itself the compiler hoists the tile's index and
    // address arithmetic out of the callers' tile loops and then spills it (56-448 bytes of scratch per lane, measured)
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const ScoreParams &sp = p.sp;
    constexpr int NWV = COOP ? NWV_ : 1;
    constexpr int NWV1 = COOP ? (WSKIP >= 0 ? NWV_ - 1 : NWV_) : 1;      // wavefronts that form the products
    constexpr int NTL = COOP ? 2 : 1;
    constexpr int NIT = (NTL * RG_TT * 4 * NG + NWV1 * 64 - 1) / (NWV1 * 64);   // product items per lane: NTL x 16 tokens x 4 c16 float4 columns
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
        const int pw = (COOP && WSKIP >= 0 && wv > WSKIP) ? wv - 1 : (COOP ? wv : 0);   // this wavefront among the NWV1
        const bool p1 = !COOP || WSKIP < 0 || wv != WSKIP;
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
            else     oth[it] = ld4_agent(bs_other_row(p, b, dir, len, i) + (s4 < SP ? s4 : 0));
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
        rg_f32x4 bn[NG];                                  // the next column block's fragments, in flight behind the MFMAs
        if (!COOP) bs_load_b(sp, 0, lane, bn);
        for (int cb = wv; cb < ncb; cb += NWV) {
            rg_f32x4 bf[NG];
            if (COOP) {
                if (cb == wv) {
#pragma unroll
                    for (int g = 0; g < NG; g++) bf[g] = bpre[g];
                } else bs_load_b(sp, cb, lane, bf);
            } else {
#pragma unroll
                for (int g = 0; g < NG; g++) bf[g] = bn[g];
                bs_load_b(sp, cb + 1 < ncb ? cb + 1 : cb, lane, bn);
            }
            rg_f32x4 acc0 = rg_f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int g = 0; g < NG; g++) {
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
    constexpr int HW = COOP ? NWV / 2 : 1;               // wavefronts per tile in the decode
    const int ti3 = COOP ? (wv / HW) : 0;
    const int kk3 = ti3 ? k1 : k0;
    const int t0 = kk3 * RG_TT, nt = kk3 >= 0 ? min(RG_TT, nsteps - t0) : 0;
    float *sclt = scl + ti3 * RG_TT * Kc;
    for (int tg = COOP ? 4 * (wv % HW) : 0; tg < RG_TT; tg += 4 * HW) {
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

// ---- the same two tiles when the output matrix is a label map (label_map.hip.h): every wavefront takes tokens w, w + NWV, ... of
// each tile and scores + decodes them by itself -- no staged products, no matrix-core pass, no workgroup barrier.
// Straight-line code: every row entry of the wavefront's tokens is fetched before the first token is scanned (tokens that do not
// exist read token 0's rows and store nothing), two tokens are scanned at a time.  PARKED: the other direction's rows of both
// tiles are in LDS (par0 / par1), else they come from the stash (sc1 loads; the acquire that covers them is the caller's).
template <int NWV, bool PARKED>
__device__ __forceinline__ void bs_label_map_tiles(const BesideParams &p, const int b, const int dir, const int len, const int nsteps,
                                                   const int k0, const int k1, const float *hist, const float *par0, const float *par1,
                                                   const long long foff, const int wv, const int lane, const LabelMapRegs &lr) {
    static_assert(RG_TT % NWV == 0, "whole tokens-per-tile share per wavefront");
    constexpr int TPT = RG_TT / NWV, NTOK = 2 * TPT;     // tokens per tile and wavefront; two tiles
    static_assert(NTOK % 2 == 0, "tokens are scanned in pairs");
    const ScoreParams &sp = p.sp;
    const int SP = p.SP;
    float a0[NTOK], a1[NTOK], o0[NTOK], o1[NTOK];
    int pos[NTOK];
#pragma unroll
    for (int q = 0; q < NTOK; q++) {
        constexpr int dummy = 0; (void)dummy;
        const int ti = q / TPT, tokl = wv + (q % TPT) * NWV;                 // (compile-time tile, wave-uniform token slot)
        const int kk = ti ? k1 : k0;
        const int t0 = (kk >= 0 ? kk : 0) * RG_TT, nt = kk >= 0 ? min(RG_TT, nsteps - t0) : 0;
        const bool live = tokl < nt;
        const int tk = live ? tokl : 0;
        const int i = t0 + tk;
        pos[q] = live ? i : -1;
        const int ai = i + 1, bi = (i + 1 <= len) ? len - (i + 1) : i + 1;
        const float *own = hist + (dir == 0 ? ai : bi) * SP;
        a0[q] = own[lr.st0]; a1[q] = own[lr.st1];
        if (PARKED) {
            const float *par = (ti ? par1 : par0) + tk * SP;
            o0[q] = par[lr.st0]; o1[q] = par[lr.st1];
        } else {
            const float *orow = bs_other_row(p, b, dir, len, i);
            o0[q] = __hip_atomic_load(orow + lr.st0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            o1[q] = __hip_atomic_load(orow + lr.st1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const bool two = sp.lm.nq > 1;                       // one register in use: the second one's products are zero (selected, not
                                                         // multiplied by 0: an infinite state must not turn into a NaN there)
#pragma unroll
    for (int q = 0; q < NTOK; q += 2) {
        float ya0, ya1, yb0, yb1;
        lm_scan_scores2(lr, a0[q] * o0[q], two ? a1[q] * o1[q] : 0.0f, a0[q + 1] * o0[q + 1], two ? a1[q + 1] * o1[q + 1] : 0.0f, ya0, ya1, yb0, yb1);
        float ma = fmaxf(ya0, ya1), mb = fmaxf(yb0, yb1);
        wave_max_dpp2(ma, mb);
        const int taga = lm_tag_from_candidates(sp.lm, lr, ya0, ya1, ma, sp.K, sp.o_idx);
        const int tagb = lm_tag_from_candidates(sp.lm, lr, yb0, yb1, mb, sp.K, sp.o_idx);
        if (lane < 2) {
            const int i = lane ? pos[q + 1] : pos[q];
            const int tag = lane ? tagb : taga;
            if (i >= 0) {
                if (sp.tags) sp.tags[(long long)b * p.L + i] = tag;
                if (sp.flat && i < len) sp.flat[foff + i] = tag;
            }
        }
    }
}

// misc words in LDS
enum { RGM_FOFF = 16,        // where the sequence starts in the flat output
       RGM_MINE = 17,        // tiles this workgroup's scorer did while the chain ran (bit k = tile k)
       RGM_ACQ = 18,         // the other direction's progress covered by this workgroup's latest acquire
       RGM_TODO = 19,
       RGM_PARK = 21 };      // [RG_NOB] tile + 1 whose rows of the other direction are parked in obuf[slot]     // the partial-sum reduction's identity (0.0f / -inf): what a masked read returns      // tiles the eight wavefronts score together next


// where sequence b starts in the flat output (utils.py:153-164): the sum of the lengths in front of it.  One wavefront.
__device__ __forceinline__ int bs_flat_offset(const BesideParams &p, int b, int lane) {
    int partsum = 0;
    if (p.sp.flat) {
        if (p.sp.offs) partsum = lane == 0 ? (int)p.sp.offs[b] : 0;
        else for (int j = lane; j < b; j += WAVE) partsum += clamp_len(p.sp.len[j], p.L);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) partsum += __shfl_xor(partsum, off, WAVE);
    return partsum;
}

// tiles kmid.. belong to the forward workgroup's half, the rest to the backward one's; pubmax = the last row of THIS direction
// that a tile of the OTHER workgroup's half needs (the progress word is published per batch of rows only up to there)
__device__ __forceinline__ void bs_halves(int dir, int len, int nsteps, int &kmid, int &pubmax) {
    const int ntl = (nsteps + RG_TT - 1) / RG_TT;
    kmid = 0; pubmax = 0;
    for (; kmid < ntl; kmid++) {
        int na, nb;
        bs_tile_need(kmid, len, nsteps, na, nb);
        if (na >= nb) break;
    }
    for (int k = 0; k < ntl; k++) {
        int na, nb;
        bs_tile_need(k, len, nsteps, na, nb);
        const bool others = dir == 0 ? k < kmid : k >= kmid;
        if (others) pubmax = max(pubmax, dir == 0 ? na : nb);
    }
}

// ---- the chain is done.  Before the arrival a workgroup touches only the tiles of its own half: those the other direction's
// progress (as covered by an acquire) allows are scored now by all NWV wavefronts; the arrival word carries the mask of this
// workgroup's tiles, and the workgroup that finds the other's word there scores whatever neither has.
// Every wavefront of the workgroup calls it.  WDEC = the wavefront that polls / decides (mine, acq in misc[RGM_MINE / RGM_ACQ]);
// on_meet(): called by every wavefront right behind the meeting barrier (chain_regs: the writer issues the state rows it has not
// copied yet); publish_all(): called before the second arrival may be acted on -- when it returns on the wavefront that stored
// this direction's stash rows, they are drained and the progress word says nsteps.
// WARM (label-map path): the deciding wavefront is idle while the chain runs its last steps -- it executes the tile code ONCE WITH
// NOTHING TO STORE before the meeting barrier (the same call site, no tile selected), so that the instructions are in the
// compute unit's instruction cache when all wavefronts run them for real: run cold, a few KB of straight-line code cost the
// workgroup ~8 k cycles of instruction fetch (profiles/r04_probe_finish_phases.txt), more than the arithmetic.
// LMO: the instantiation serves the label-map path only (the caller guarantees bs_label_map_path): the matrix-core tile code
// is not compiled in -- its B fragments and accumulators are what the 128-VGPR wide form has no room for.
template <int NWV, int NG, int WDEC, int WSKIP, bool WARM = false, bool LMO = false, class OnMeet, class PublishAll>
__device__ __forceinline__ void bs_finish(const BesideParams &p, const int b, const int dir, const int len, const int nsteps, const int kmid,
                                          const float *hist, float *ab, float *scl, const float *obuf, int *misc,
                                          const int w, const int lane, const unsigned lm_pk0, const unsigned lm_pk1,
                                          const unsigned epoch,      // the launch's epoch (bs_launch_epoch); read on wavefront WDEC only
                                          OnMeet on_meet, PublishAll publish_all) {
    const int SP = p.SP;
    const int ntl = (nsteps + RG_TT - 1) / RG_TT;
    rg_f32x4 bpre[NG];                                            // this wavefront's column block of O^T: in flight across the barrier
    const bool label_map = LMO || bs_label_map_path(p.sp);        // the tiles go through bs_label_map_tiles, not the matrix cores
    LabelMapRegs lr;
    if (label_map) {
        lm_unpack(p.sp.lm, lm_pk0, lm_pk1, lr);
#pragma unroll
        for (int g = 0; g < NG; g++) bpre[g] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
    } else if constexpr (!LMO) bs_load_b<NG>(p.sp, w < p.sp.Kc / 16 ? w : 0, lane, bpre);
    const unsigned all_tiles = ntl >= 32 ? ~0u : ((1u << ntl) - 1u);
    unsigned promised = 0u;
#if defined(FARNN_PROBES)
    const bool fprobe = nsteps == p.L && p.L >= 32 && w == WDEC && (p.dbg & 1024);
    long long fq[6] = {0, 0, 0, 0, 0, 0};
#define FARNN_BS_STAMP(i) do { if (fprobe) fq[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define FARNN_BS_STAMP(i) do { } while (0)
#endif
#if defined(FARNN_PROBES)
    __shared__ long long bs_arrive[16];                              // when each wavefront reached the meeting barrier
#endif
    FARNN_BS_STAMP(0);
    if (w == WDEC) {
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
                bs_tile_need(k, len, nsteps, na, nb);
                if ((dir == 0 ? nb : na) <= acq) todo |= 1u << k; else miss++;
            }
            if (!miss || it >= p.spin) break;                         // bounded: what stays open goes to the second arrival
            int pr = lane == 0 ? bs_read_prog(oprog, epoch) : 0;
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
    unsigned long long arrived = 0ull;
    long long foff = 0;
    int rep = __builtin_amdgcn_readfirstlane((WARM && label_map && w == WDEC) ? 0 : 1);
    // (ONE copy of the loop's body, not a peeled dry run beside the real one: the code-object check in tests/ counts the scans)
#pragma clang loop unroll(disable)
    for (; rep < 2; rep++) {
    const bool dry = rep == 0;                                       // (WARM: see above)
    if (!dry) {
#if defined(FARNN_PROBES)
    if (lane == 0) bs_arrive[w] = (long long)__builtin_amdgcn_s_memtime();
#endif
    wg_barrier_lds();                                                // the chain is done, the mask is there (LDS only: this wavefront's
                                                                     // O^T loads and the writer's last stores stay in flight)
    FARNN_BS_STAMP(1);
    on_meet();
    // ---- arrival: ONE lane exchanges the sequence's arrival word for {epoch, the tiles this workgroup scores}; the
    // exchange is in flight while those tiles are scored.  (The word says nothing about this workgroup's stash rows: they
    // are published through the progress word, below; the workgroup that has to read them waits for that.)
    if (w == WDEC && lane == 0)
        arrived = __hip_atomic_exchange(p.arr + b, ((unsigned long long)epoch << 32) | 0x80000000ull | promised,
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    foff = misc[RGM_FOFF];
    }
    for (int pass = 0; pass < 2; pass++) {
        unsigned todo = dry ? 1u : (unsigned)misc[RGM_TODO];
        while (todo) {                                               // two tiles per pass
            int k0 = __builtin_ctz(todo);
            todo &= todo - 1u;
            int k1 = todo ? __builtin_ctz(todo) : -1;
            if (todo) todo &= todo - 1u;
            const float *par0 = nullptr, *par1 = nullptr;
#pragma unroll
            for (int sl = 0; sl < RG_NOB; sl++) {
                if (misc[RGM_PARK + sl] == k0 + 1) par0 = obuf + sl * RG_TT * SP;
                if (misc[RGM_PARK + sl] == k1 + 1) par1 = obuf + sl * RG_TT * SP;
            }
            if (dry) { k0 = -1; k1 = -1; par0 = obuf; par1 = obuf; }
            if (label_map) {
                if (par0 && (k1 < 0 || par1)) bs_label_map_tiles<NWV, true>(p, b, dir, len, nsteps, k0, k1, hist, par0, par1 ? par1 : par0, foff, w, lane, lr);
                else bs_label_map_tiles<NWV, false>(p, b, dir, len, nsteps, k0, k1, hist, nullptr, nullptr, foff, w, lane, lr);
            } else if constexpr (!LMO) {
                bs_score_tiles<true, NWV, NG, WSKIP>(p, b, dir, len, nsteps, k0, k1, hist, par0, par1, ab, scl, foff, w, lane, bpre);
                __syncthreads();                                     // the tiles' LDS is free again
            }
        }
        if (dry || pass == 1) break;
        FARNN_BS_STAMP(2);
        wg_barrier_lds();                                            // every wavefront has read this pass's mask (it is rewritten below)
        // every stash row of this direction has been stored by the writer wavefront: it drains them (they had the tiles'
        // time to land) and publishes the full count
        publish_all();
        FARNN_BS_STAMP(3);
        if (w == WDEC) {
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(arrived >> 32));
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)arrived);
            unsigned rest = 0u;
            if (hi == epoch && (lo & 0x80000000u)) {               // second of the two
                if (lane == 0 && p.done) __hip_atomic_fetch_add(p.done, bs_epoch_add(b, p.B), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (bs_launch_epoch)
                rest = all_tiles & ~(promised | (lo & 0x7fffffffu));
                if (rest) {
                    // The other workgroup has arrived: it is resident, past its chain, and publishes its full row count
                    // after a bounded amount of work of its own (it waits for nobody) -- so this wait ends.
                    const unsigned long long *oprog = p.prog + (long long)(dir ^ 1) * p.B + b;
                    for (;;) {
                        int pr = lane == 0 ? bs_read_prog(oprog, epoch) : 0;
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
        FARNN_BS_STAMP(4);
        __syncthreads();
        if (misc[RGM_TODO] == 0) break;
    }
    }
#if defined(FARNN_PROBES)
    if (fprobe && lane == 0) {
        printf("finish seq %d dir %d: decide + meet %lld, own tiles %lld, barrier + publish %lld, arrival known + sweep decision %lld, to the end %lld\n", b, dir,
               fq[1] - fq[0], fq[2] - fq[1], fq[3] - fq[2], fq[4] - fq[3], (long long)__builtin_amdgcn_s_memtime() - fq[4]);
        long long lo = bs_arrive[0];
        for (int i = 1; i < NWV; i++) lo = bs_arrive[i] < lo ? bs_arrive[i] : lo;
        printf("meet seq %d dir %d: wavefronts reach the barrier at +%lld %lld %lld %lld %lld %lld %lld %lld (first = 0), released at +%lld\n", b, dir,
               bs_arrive[0] - lo, bs_arrive[1 % NWV] - lo, bs_arrive[2 % NWV] - lo, bs_arrive[3 % NWV] - lo, bs_arrive[4 % NWV] - lo,
               bs_arrive[5 % NWV] - lo, bs_arrive[6 % NWV] - lo, bs_arrive[7 % NWV] - lo, fq[1] - lo);
    }
#endif
#undef FARNN_BS_STAMP
}

}  // namespace farnn
