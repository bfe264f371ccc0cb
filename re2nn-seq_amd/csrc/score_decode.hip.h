// K2 -- per-token label scores from the stashed forward/backward states, fused with the decode.
//
// Reference: the second loop of forward_score + get_final_score + local_decode
//   i-FST     model_onehot.py:346-349, :417-426, :162-180     score[c] = sum_s O[c,s] a[i+1][s] b~[i+1][s]
//   decomposed model_decompose_single.py:202-205, :263-269    (same form with C_output_mat)
//   decode    model_decompose.py:339-371 (argmax or CRF) ; crf.py:102-195 (Viterbi)
//
// score_tile_kernel: one workgroup (8 wavefronts) per (sequence, 32-token tile).  Per tile this is
// a small GEMM [32 x S] . [S x K] whose right operand (the transposed output matrix) is shared by
// every token of every sequence: it is pulled into LDS by LDS-DMA while the a*b products of the
// tile are formed, then the GEMM is register-blocked (a wavefront scores 4 tokens at a time, a lane
// owns label columns {lane, lane+64, ...}).  Scores never go to HBM unless the caller asks for
// them: threshold clamp, first-index argmax (on the DPP network, no LDS round trips) and the
// `oo -> o_idx` mapping run in the same kernel.
//
// viterbi_kernel (use_crf=1): one workgroup per sequence runs the max-plus DP with the transition
// table, the partitions and the back-pointers in LDS; it reads the clamped scores the tile kernel
// left in the workspace (B*L*K floats, ~1% of the chain kernel's traffic).
#pragma once
#include <type_traits>
#include "common.hip.h"
#include "score_params.hip.h"

namespace farnn {

constexpr int SCORE_KCH = 4;       // label columns per lane: K <= 256
constexpr int SCORE_WAVES = 8;
constexpr int SCORE_TT = 32;       // tokens per tile (4 per wavefront)

// KCH = label columns per lane in the decode (K <= 64*KCH), compile-time.
//
// score_tiles: the tiles tile_first, tile_first + tile_step, ... (32 tokens each) of sequence b, by the 8 wavefronts
// of the calling workgroup (every thread calls it; `smem` = 16-byte aligned LDS of score_lds_bytes()).  SC1: the stash
// is read with agent-scope (sc1) loads -- the form a fused epilogue needs, where another workgroup of the same launch
// wrote (part of) it.
//
// Per tile: (1) ab[tok][s] = alpha * beta products from the two stashes into LDS; (2) scores[32][Kc] = ab . O^T on the
// f32 matrix cores: v_mfma_f32_16x16x4_f32 accumulates its four products in k order, i.e. the same ascending-s fmaf
// chain the r01 VALU loop ran (MI355X_MICROARCH.md: bitwise the fmaf chain) -- a wavefront owns one token half and KCH
// of the Kc/16 column blocks, its A fragments come from the LDS tile (shared by its blocks), its B fragments straight
// from the matrix-core image of the output matrix in L2 (OTm: one 16-byte load per lane covers four k-steps; nothing is
// staged -- r01/r02a moved the whole 54 KB matrix into every workgroup's LDS), the accumulators go to an LDS score tile;
// (3) 16 lanes per token: priority matrix / threshold clamp / first index of the row maximum.
// The f32 matrix cores run at the f32 VALU rate (256 flop/clk/CU): the gain over the VALU loop is the LDS operand
// traffic and the issue slots, not the arithmetic.  Measured (FARNN_DBG=16384, one tile of the config-1 batch, two
// workgroups per CU, cycles): r02a products 1.8 k, DMA wait 0.7 k, VALU GEMM 5.8 k, keyed-DPP decode 5.4 k; now products
// 2.3 k, barrier 1.0 k, GEMM 4.3 k, decode 1.7 k.  score_tile_kernel for the config-1 batch: 14.4 -> 10.6 us.
// foff_pre >= 0 / len_pre >= 0: the sequence's offset in the flat output / its clamped length are already known (the fused
// epilogue of chain_kernel has both)
template <int KCH, bool SC1>
__device__ __forceinline__ void score_tiles(const ScoreParams &p, const int b, const int tile_first, const int tile_step,
                                            float *smem, const int tid, const long long foff_pre = -1, const int len_pre = -1) {
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nthreads = SCORE_WAVES * 64;
    const int len = len_pre >= 0 ? len_pre : clamp_len(p.len[b], p.L);   // (the fused epilogue knows it: no L2 round trip)
    const int nsteps = p.full ? p.L : len;
    const int SP = p.SP, K = p.K, Kc = p.Kc;
    const int ntiles = (p.L + SCORE_TT - 1) / SCORE_TT;

    // ---- LDS carve ---------------------------------------------------------------------------
    const int c16 = p.c16;                               // 16-state groups of the output matrix image
    const int SPa = 16 * c16 + 4;                        // row stride of the products tile: whole groups (zero beyond SP), and
                                                         // SPa/4 odd -- the 64 lanes of an A-fragment read hit 64 banks
    float *ab = smem;                                    // [TT][SPa] alpha*beta of the tile
    float *scl = ab + SCORE_TT * SPa;                    // [TT][Kc]  scores of the tile
    // no prepared offsets: where this sequence starts in the flat output = sum of the lengths before it
    // (utils.py:153-164); B <= 1024 here, two loads per thread
    __shared__ int foff_w[SCORE_WAVES];
    long long foff = foff_pre >= 0 ? foff_pre : (p.offs ? p.offs[b] : 0);
    if (foff_pre < 0 && !p.offs && p.flat) {
        int part = 0;
        for (int j = tid; j < b; j += nthreads) {
            const int v = (int)p.len[j];
            part += v < 0 ? 0 : (v > p.L ? p.L : v);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, WAVE);
        if (lane == 0) foff_w[w] = part;
        __syncthreads();
        int foff_s = 0;
#pragma unroll
        for (int ww = 0; ww < SCORE_WAVES; ww++) foff_s += foff_w[ww];
        foff = (long long)foff_s;
    }

    const float *Ab = p.A + (long long)b * (p.L + 1) * SP;
    const float *Bb = p.Bk + (long long)b * (p.L + 1) * SP;
    const int SP4 = SP >> 2;
    const int clamp_col = p.use_crf ? K - 3 : K - 1;      // model_decompose.py:353 / :365
    const int lr = lane & 15, lk = lane >> 4;            // MFMA fragment coordinates
    const bool probe = FARNN_PROBE_ON(p.dbg & 16384) && nsteps == p.L && tid == 0;     // diagnostic: cycles of the phases
    long long q0 = probe ? (long long)__builtin_amdgcn_s_memtime() : 0;
    for (int tile = tile_first; tile < ntiles; tile += tile_step) {
        const int t0 = tile * SCORE_TT;
        const int nt = min(SCORE_TT, nsteps - t0);           // tokens of this tile that were computed
        const int ntL = min(SCORE_TT, p.L - t0);             // tokens of this tile that exist
        if (nt <= 0) {      // a tile of pads only (LOCAL mode); uniform over the workgroup
            for (int i = t0 + w; i < t0 + ntL; i += SCORE_WAVES) {
                if (p.tags && lane == 0) p.tags[(long long)b * p.L + i] = -1;
                if (p.scores)
                    for (int col = lane; col < K; col += WAVE) p.scores[((long long)b * p.L + i) * K + col] = 0.0f;
            }
            continue;
        }
        // This wavefront's output blocks: token half w & 1, column blocks (w >> 1) * KCH + q -- KCH blocks that share their A
        // fragments (one LDS read serves KCH matrix-core instructions) and accumulate independently (consecutive MFMAs never
        // wait for each other).  B fragments: KCH 16-byte loads per state group, kept two groups ahead of the matrix cores;
        // the first two groups fly while the products are formed.  All loops here are ROLLED: a tile runs this code once,
        // and with the loops unrolled (r02a: 27-32 KB of straight-line code per tile) instruction fetch was a visible cost.
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 *otm = reinterpret_cast<const f32x4 *>(p.OTm) + ((long long)(w >> 1) * KCH * c16) * 64 + lane;
        auto load_b = [&](int g, f32x4 (&dst)[KCH]) {            // group g (clamped) of the KCH column blocks
            const int gc = g < c16 ? g : c16 - 1;
#pragma unroll
            for (int q = 0; q < KCH; q++) dst[q] = otm[((long long)q * c16 + gc) * 64];
        };
        f32x4 be[KCH], bo[KCH];                                  // even / odd groups
        load_b(0, be); load_b(1, bo);
        // ---- phase 1: ab[tok][s] = a[i+1][s] * b~[i+1][s]; alpha = state after i+1 tokens, beta =
        // backward state before token i+1 is consumed (reversed_backward_score_x[:, i+1], :415-420)
        const int G4 = 4 * c16;                               // float4 columns of a tile row (those past SP stay zero)
        for (int idx0 = 0; idx0 < SCORE_TT * G4; idx0 += 2 * nthreads) {
            v4f a4[2], b4[2];
            int tokv[2], s4v[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int idx = idx0 + r * nthreads + tid;
                const int tok = idx / G4;
                tokv[r] = tok; s4v[r] = (idx - tok * G4) * 4;
                a4[r] = v4f{0.f, 0.f, 0.f, 0.f}; b4[r] = a4[r];
                if (tok < nt && s4v[r] < SP) {
                    const int i = t0 + tok;
                    const int bidx = (i + 1 <= len) ? len - (i + 1) : i + 1;
                    if (SC1) {
                        ld4_agent_issue(a4[r], Ab + (long long)(i + 1) * SP + s4v[r]);
                        ld4_agent_issue(b4[r], Bb + (long long)bidx * SP + s4v[r]);
                    } else {
                        a4[r] = *reinterpret_cast<const v4f *>(Ab + (long long)(i + 1) * SP + s4v[r]);
                        b4[r] = *reinterpret_cast<const v4f *>(Bb + (long long)bidx * SP + s4v[r]);
                    }
                }
            }
            if (SC1) wait_sc1_loads(a4[0], b4[0], a4[1], b4[1]);
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (tokv[r] < SCORE_TT)
                    st4(ab + tokv[r] * SPa + s4v[r], make_float4(a4[r].x * b4[r].x, a4[r].y * b4[r].y,
                                                                  a4[r].z * b4[r].z, a4[r].w * b4[r].w));
        }
        long long q1 = probe ? (long long)__builtin_amdgcn_s_memtime() : 0;
        __syncthreads();
        long long q2 = probe ? (long long)__builtin_amdgcn_s_memtime() : 0;

        // ---- phase 2: the wavefront's KCH output blocks on the matrix cores -----------------------------------------
        if (!(p.dbg & 32)) {
            const float *arow = ab + ((w & 1) * 16 + lr) * SPa + lk;           // this lane's k of k-step 4g + e: 16g + 4e + lk
            f32x4 acc[KCH];
#pragma unroll
            for (int q = 0; q < KCH; q++) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto a_group = [&](int g, float (&a)[4]) {                         // A fragments of group g (clamped: unused past c16)
                const float *ap = arow + 16 * (g < c16 ? g : c16 - 1);
                a[0] = ap[0]; a[1] = ap[4]; a[2] = ap[8]; a[3] = ap[12];
            };
            auto mfma_group = [&](const float (&a)[4], const f32x4 (&bb)[KCH]) {
#pragma unroll
                for (int q = 0; q < KCH; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], bb[q].x, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < KCH; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], bb[q].y, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < KCH; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], bb[q].z, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < KCH; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], bb[q].w, acc[q], 0, 0, 0);
            };
            float ae[4], ao[4];
            a_group(0, ae);
#pragma unroll 1
            for (int g = 0; g < c16; g += 2) {                // a register set is reloaded right after its use
                a_group(g + 1, ao);
                mfma_group(ae, be);
                load_b(g + 2, be);
                a_group(g + 2, ae);
                if (g + 1 < c16) mfma_group(ao, bo);
                load_b(g + 3, bo);
            }
            // rows lk*4 + r of the token half, column lr of each block
#pragma unroll
            for (int q = 0; q < KCH; q++) {
                float *dst = scl + ((w & 1) * 16 + lk * 4) * Kc + ((w >> 1) * KCH + q) * 16 + lr;
                dst[0] = acc[q].x; dst[Kc] = acc[q].y; dst[2 * Kc] = acc[q].z; dst[3 * Kc] = acc[q].w;
            }
        }
        __syncthreads();
        long long q3 = probe ? (long long)__builtin_amdgcn_s_memtime() : 0;
        // ---- phase 3: the wavefront's 4 tokens AT ONCE: 16 lanes per token, lane c of a token's row holds the columns
        // 64m + 4c + e (one 16-byte LDS read per m).  Priority, outputs, threshold clamp, then the first index of the row
        // maximum (torch.max's rule): lane maximum, 4 DPP steps over the 16 lanes, the lane's first column that equals the
        // row maximum, 4 DPP steps of an unsigned minimum.  (r02a decoded token by token on all 64 lanes: ~100 issued
        // instructions per token, 3.4 k cycles per tile with the CU's sixteen wavefronts all in this phase.)
        const int tg = w * 4;
        if (p.P) {              // PriorityLayer: scores @ P (priority.py:20-30), row by row, back into the score tile
#pragma unroll 1
            for (int j = 0; j < 4; j++) {
                if (tg + j >= nt) break;
                float *sr = scl + (tg + j) * Kc;
                float sc[KCH];
#pragma unroll
                for (int k = 0; k < KCH; k++) sc[k] = 0.0f;
                for (int cc = 0; cc < K; cc++) {
                    const float sv = sr[cc];
                    const float *prow = p.P + (long long)cc * Kc + lane;
#pragma unroll
                    for (int k = 0; k < KCH; k++) sc[k] = fmaf(sv, prow[64 * k], sc[k]);
                }
                __builtin_amdgcn_wave_barrier();              // the row is this wavefront's alone: all reads before the writes
#pragma unroll
                for (int k = 0; k < KCH; k++) sr[lane + 64 * k] = sc[k];
            }
            __builtin_amdgcn_wave_barrier();
        }
        {
            const int j = lane >> 4, c = lane & 15;
            const int tokl = tg + j, i = t0 + tokl;
            const bool live = tokl < nt;                          // computed token; else a pad inside the tile (or nothing)
            float v[KCH][4];
#pragma unroll
            for (int m = 0; m < KCH; m++) {
                const float4 x4 = ld4(scl + (tokl < SCORE_TT ? tokl : 0) * Kc + 64 * m + 4 * c);
                v[m][0] = x4.x; v[m][1] = x4.y; v[m][2] = x4.z; v[m][3] = x4.w;
            }
            if (p.scores && tokl < ntL) {                         // unclamped; zero rows at the pads of LOCAL mode
                float *so = p.scores + ((long long)b * p.L + i) * K;
#pragma unroll
                for (int m = 0; m < KCH; m++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int col = 64 * m + 4 * c + e;
                        if (col < K) so[col] = live ? v[m][e] : 0.0f;
                    }
            }
            float best = -INFINITY;
#pragma unroll
            for (int m = 0; m < KCH; m++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int col = 64 * m + 4 * c + e;
                    float x = v[m][e] + 0.0f;                     // -0.0 -> +0.0 (torch: -0 == +0)
                    if (col == clamp_col) x = fminf(x, p.threshold);
                    if (p.use_crf && live && col < K) p.crf_scores[((long long)b * p.L + i) * p.Kp + col] = x;
                    x = col < K ? x : -INFINITY;
                    v[m][e] = x;
                    best = fmaxf(best, x);
                }
            if (!p.use_crf) {
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
                unsigned first = 0x7fffffffu;                     // this lane's first column that holds the row maximum
#pragma unroll
                for (int m = KCH - 1; m >= 0; m--)
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
                if (c == 0 && tokl < ntL) {
                    int bi = (first >= (unsigned)K || (p.dbg & 64)) ? 0 : (int)first;   // an all-NaN row gives 0 like torch
                    const int tag = live ? ((bi == K - 1) ? p.o_idx : bi) : -1;
                    if (p.tags) p.tags[(long long)b * p.L + i] = tag;
                    if (p.flat && live && i < len) p.flat[foff + i] = tag;
                }
            }
        }
        if (probe) {
            const long long q4 = (long long)__builtin_amdgcn_s_memtime();
            printf("score tile %d of sequence %d: offsets+products %lld cycles, barrier %lld, GEMM %lld, decode %lld\n",
                   tile, b, q1 - q0, q2 - q1, q3 - q2, q4 - q3);
        }
        if (tile + tile_step < ntiles) __syncthreads();      // the next tile overwrites ab and the score tile
        if (probe) q0 = (long long)__builtin_amdgcn_s_memtime();
    }
}

template <int KCH>
__global__ void __launch_bounds__(SCORE_WAVES * 64)
score_tile_kernel(const ScoreParams p) {
    extern __shared__ __align__(16) float smem[];
    score_tiles<KCH, false>(p, blockIdx.y, blockIdx.x, gridDim.x, smem, threadIdx.x);
}

inline size_t score_lds_bytes(int S, int Kc) {         // products tile (row stride 16*ceil(S/16) + 4) + score tile
    return (size_t)SCORE_TT * (16 * ((S + 15) / 16) + 4) * 4 + (size_t)SCORE_TT * Kc * 4;
}

// Matrix-core image of the transposed output matrix OT[S][Kc]: OTm[Kc/16][c16][64 lanes][4], c16 = ceil(S/16): lane
// (lr = lane % 16, lk = lane / 16) of column block cb holds, for the four k-steps 4g + e of state group g, the entries
// OT[16g + 4e + lk][16cb + lr] (zero beyond S) -- the B operand of v_mfma_f32_16x16x4_f32, one 16-byte load per group.
__global__ void ot_to_mfma_kernel(const float *OT, float *OTm, int S, int Kc, int c16) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)(Kc / 16) * c16 * 256;
    if (i >= n) return;
    const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const long long gi = i >> 8;
    const int g = (int)(gi % c16), cb = (int)(gi / c16);
    const int s = 16 * g + 4 * e + (lane >> 4);
    OTm[i] = s < S ? OT[(long long)s * Kc + cb * 16 + (lane & 15)] : 0.0f;
}

// ---- Viterbi (crf.py:102-195) over the valid positions, one workgroup per sequence ------------
// A quad of lanes shares one destination tag j; lane q of the quad owns the CONTIGUOUS block of
// source tags i in [q*IB, (q+1)*IB).  The transition column trT[j][block] never changes over time,
// so it lives in registers for the whole sequence; per step a lane reads its block of the previous
// partition with 16-byte LDS reads, and the quad is combined with (value desc, index asc) --
// torch.max's first-index rule (lane-local strict `>` keeps the first index inside a block).
// The sequence's clamped scores and the back-pointers stay in LDS.  IB4 = IB/4 is compile-time.
// flat-output offset of sequence b without a prepared prefix array: the sum of the (clamped) lengths in
// front of it (utils.py:153-164).  Called by every thread of the workgroup; B <= 1024.
__device__ __forceinline__ long long flat_offset_in_kernel(const int64_t *len, int b, int L, int tid, int nthreads) {
    __shared__ int fo_w[16];
    int part = 0;
    for (int j = tid; j < b; j += nthreads) {
        const int v = (int)len[j];
        part += v < 0 ? 0 : (v > L ? L : v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, WAVE);
    if ((tid & 63) == 0) fo_w[tid >> 6] = part;
    __syncthreads();
    int tot = 0;
    for (int ww = 0; ww < (nthreads >> 6); ww++) tot += fo_w[ww];
    __syncthreads();
    return (long long)tot;
}

template <int IB4>
__global__ void __launch_bounds__(1024)
viterbi_kernel(const ScoreParams p) {
    constexpr int IB = IB4 * 4;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nthreads = blockDim.x;
    const int b = blockIdx.x;
    const int n = clamp_len(p.len[b], p.L);
    (void)p.full;
    const int K = p.K, Kp = p.Kp;
    const int PW = 4 * IB;                               // padded partition width (>= K)
    float *part = smem;                                  // [2][PW], pad entries -inf
    float *scl = part + 2 * PW;                          // [L][Kp] clamped scores of this sequence
    unsigned short *bp = reinterpret_cast<unsigned short *>(scl + (size_t)p.L * Kp);   // [L][Kp]
    const float *sc = p.crf_scores + (long long)b * p.L * Kp;
    const long long foff = p.offs ? p.offs[b] : (p.flat ? flat_offset_in_kernel(p.len, b, p.L, tid, nthreads) : 0);
    const int START = K - 2, STOP = K - 1;
    const float ninf = -INFINITY;

    for (int i = tid * 4; i < n * Kp; i += nthreads * 4) st4(scl + i, ld4(sc + i));
    for (int i = tid; i < 2 * PW; i += nthreads) part[i] = ninf;
    const int j = tid >> 2, q = tid & 3;
    const bool owner = j < K;
    float trr[IB];                                       // tr[i][j] for this lane's block of i
#pragma unroll
    for (int k = 0; k < IB; k++) {
        const int i = q * IB + k;
        trr[k] = (owner && i < K) ? p.trT[(long long)j * Kp + i] : ninf;
    }
    __syncthreads();
    for (int jj = tid; jj < K; jj += nthreads)
        part[jj] = scl[jj] + p.trT[(long long)jj * Kp + START];              // crf.py:135
    __syncthreads();
    int pc = 0;
    for (int t = 1; t < n; t++) {
        const float *pin = part + pc * PW + q * IB;
        float *pout = part + (pc ^ 1) * PW;
        const float f = owner ? scl[(long long)t * Kp + j] : 0.0f;
        float best = ninf; int bi = 0x7fffffff;
#pragma unroll
        for (int k4 = 0; k4 < IB4; k4++) {
            const float4 p4 = ld4(pin + k4 * 4);
            const float pv[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float v = (f + trr[k4 * 4 + u]) + pv[u];                // crf.py:123,145
                if (v > best) { best = v; bi = q * IB + k4 * 4 + u; }
            }
        }
#pragma unroll
        for (int off = 1; off <= 2; off <<= 1) {
            const float ov = __shfl_xor(best, off, WAVE);
            const int oi = __shfl_xor(bi, off, WAVE);
            argmax_combine(best, bi, ov, oi);
        }
        if (owner && q == 0) {
            pout[j] = best;
            bp[(long long)t * Kp + j] = (unsigned short)(bi >= K ? 0 : bi);
        }
        __syncthreads();
        pc ^= 1;
    }
    if (w == 0) {
        const float *pin = part + pc * PW;
        float bv = ninf; int bi = 0x7fffffff;
        for (int i = lane; i < K; i += WAVE) {
            const float v = pin[i] + p.trT[(long long)STOP * Kp + i];          // crf.py:168-169
            if (v > bv) { bv = v; bi = i; }
        }
        wave_argmax(bv, bi);
        if (lane == 0) {
            if (bi >= K) bi = 0;
            int ptr = bi;
            for (int t = n - 1; t >= 0; t--) {
                const int tag = (ptr == K - 3) ? p.o_idx : ptr;               // model_decompose.py:356
                if (p.tags) p.tags[(long long)b * p.L + t] = tag;
                if (p.flat) p.flat[foff + t] = tag;
                if (t > 0) ptr = bp[(long long)t * Kp + ptr];
            }
        }
    }
    if (p.tags)
        for (int i = n + tid; i < p.L; i += nthreads) p.tags[(long long)b * p.L + i] = -1;   // pads (LOCAL and FULL)
}

// floats of the history area of viterbi_hist_kernel: [L][Kp] partitions; the fused form first stages the [SP][L]
// alpha*beta products there (row stride L + 16)
__host__ __device__ inline size_t viterbi_hist_floats(int Kp, int SP, int L, bool fused) {
    const size_t a = (size_t)L * Kp, c = fused ? (size_t)SP * (((L + 3) & ~3) + 16) : 0;
    return a > c ? a : c;
}

// LDS reads / counted waits as explicit instructions (the forward step of viterbi_hist_kernel issues ALL of a step's reads up
// front; left to the compiler they were issued piecemeal through a recycled register quad -- three exposed LDS round trips per
// step).  The registers a wait releases are its "+v" operands, so no use of them can move in front of it.
template <int OFF> __device__ __forceinline__ void lds_read16_at(v4f &d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lds_wait_for(v4f &d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }

// History variant of the DP (used when the LDS holds it): the forward pass keeps only the partition
// VALUES of every step; the back-pointers the reference stores (crf.py:147-149) are recomputed lazily along the ONE
// path the backtrace follows: bp_t[j] = first argmax_i ((f_t[j] + tr[i][j]) + part_{t-1}[i]) is the same f32
// expression on the same values, so the path is bit-identical.  The transposed transition table stays in LDS for
// that second pass.
// Forward pass layout (r02; FARNN_DBG=8192 prints the phase cycle counts).  The step is VALU-issue bound on the one CU
// that owns the sequence: two adds (crf.py:123,145) and half a v_max3 per (source, tag), K*K pairs -- 660 cycles per step
// at K = 130 if the four SIMDs were perfectly balanced, plus ~300 of LDS latency and barrier.  EIGHT lanes share a PAIR of
// destination tags (j0, j0+1); lane g of the group owns the sources 32*k + 4*g .. +3 of every 32-source block k < IB4 (a
// 16-byte LDS read per block, the eight lanes cover a contiguous 128 bytes: no bank conflicts) plus up to four leftover
// sources 32*IB4 + g + 8x (K - 32*IB4 < 32): at most one wasted source slot per lane for any K (r01: four lanes per tag,
// block sizes 8/16/36/52/64 -- K = 75 ran 36 slots per lane for 19 useful ones).  The two tags ride in the halves of
// packed-f32 registers (v_pk_add_f32 issues at half rate on gfx950, so this saves registers and LDS reads, not issue
// slots); the group is combined by three v_max with DPP operands per tag.  Tag pairs that do not fill a wavefront (K = 130:
// the START/STOP pair) go to a TAIL wavefront that spreads them over all its lanes and runs at raised priority -- its
// step is a short latency chain that otherwise runs behind the SIMD's older, issue-bound wavefronts.
// Measured on the config-3 batch (K = 130, 64 positions; cycles at ~2.3 GHz): set-up + scores 25.6 k, forward pass
// 1 190 per step (K = 128: 910), backtrace 590 per step (r01: 1 140 -- the keyed 64-bit DPP argmax).  Tried and dropped:
// hoisting (f + tr) of the coming step behind the LDS write (no gain: the idle time there is ~100 cycles), conditional
// leftover reads (waits inside branches: +130 per step).
// FUSED: the workgroup also computes the clamped scores of its sequence (what score_tile_kernel would
// have written to crf_scores) straight into LDS -- one kernel from stash to tags, no score round trip
// through HBM: alpha*beta products staged transposed in the (not yet used) history area, then a
// register-blocked [tokens x S].[S x K] product, 4 tokens x 4 tags per lane, against the L2-resident
// transposed output matrix.  Same fmaf chain in s order as score_tile_kernel: identical bits.
template <int IB4, bool FUSED>
__global__ void __launch_bounds__(IB4 < 7 ? 128 * IB4 + 128 : 1024)        // K < 32*IB4 + 32: 8 lanes per tag pair
viterbi_hist_kernel(const ScoreParams p) {
    constexpr int IB = IB4 * 4;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nthreads = blockDim.x;
    const int b = blockIdx.x;
    const int n = clamp_len(p.len[b], p.L);
    (void)p.full;
    const int K = p.K, Kp = p.Kp;
    const int PW = Kp;                                   // partition row stride (K rounded up to 4), pads -inf
    float *hist = smem;                                  // [L][PW] partitions of every step (FUSED: first the products)
    const int sc_pieces = (p.L * Kp * 4 + 1023) / 1024, tr_pieces = (K * Kp * 4 + 1023) / 1024;
    float *scl = hist + viterbi_hist_floats(Kp, p.SP, p.L, FUSED);   // [L][Kp] clamped scores of this sequence (whole KiB)
    float *trl = scl + sc_pieces * 256;                  // [K][Kp] trT: trl[j][i] = transitions[i][j]
    const float *sc = p.crf_scores + (long long)b * p.L * Kp;
    const long long foff = p.offs ? p.offs[b] : (p.flat ? flat_offset_in_kernel(p.len, b, p.L, tid, nthreads) : 0);
    const int START = K - 2, STOP = K - 1;
    const float ninf = -INFINITY;
    const int wu = __builtin_amdgcn_readfirstlane(w), nwaves = nthreads >> 6;
    const bool probe = FARNN_PROBE_ON(p.dbg & 8192) && n == p.L;       // diagnostic: cycle counts of the phases of a full-length sequence
    long long pc0 = probe ? (long long)__builtin_amdgcn_s_memtime() : 0, pc1 = 0, pc2 = 0, pc3 = 0, pa = 0, pb = 0;

    // set-up without a register round trip: the scores and (behind them) the transition table stream
    // into LDS by LDS-DMA; the table is only needed by the backtrace, so its pieces stay in flight
    // during the forward pass (counted vmcnt: this wavefront's table pieces are its youngest operations)
    if (!FUSED) {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)scl);
        const int need = (n * Kp * 4 + 1023) / 1024;
        for (int k = wu; k < need; k += nwaves)
            lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, reinterpret_cast<const char *>(sc), lds0 + (unsigned)k * 1024u);
    } else {
        const int SP = p.SP, SP4 = SP >> 2, Lq = ((p.L + 3) & ~3) + 16;   // row stride of the products: the four state rows of an
                                                                            // A-fragment read land in different banks (L = 64: 80)
        // the matrix-core image of the output matrix (K2: one 1 KiB piece per (column block, state group)) borrows the
        // transition table's LDS area until the scores are done, when it fits
        const int otm_pieces = ((K + 15) >> 4) * p.c16;
        const bool otm_lds = otm_pieces <= tr_pieces;
        if (otm_lds) {
            const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)trl);
            for (int k = wu; k < otm_pieces; k += nwaves)
                lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, reinterpret_cast<const char *>(p.OTm), lds0 + (unsigned)k * 1024u);
        }
        float *abT = hist;                               // [SP][Lq] alpha*beta, token-contiguous (aliases hist)
        const float *Ab = p.A + (long long)b * (p.L + 1) * SP;
        const float *Bb = p.Bk + (long long)b * (p.L + 1) * SP;
        for (int idx = tid; idx < (Lq - 16) * SP4; idx += nthreads) {
            const int tok = idx / SP4, s4 = (idx - tok * SP4) * 4;
            float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a4;
            if (tok < n) {       // alpha = state after tok+1 tokens; beta = backward state before token tok+1 (:415-420)
                a4 = ld4(Ab + (long long)(tok + 1) * SP + s4);
                b4 = ld4(Bb + (long long)(n - (tok + 1)) * SP + s4);
            }
            abT[(s4 + 0) * Lq + tok] = a4.x * b4.x; abT[(s4 + 1) * Lq + tok] = a4.y * b4.y;
            abT[(s4 + 2) * Lq + tok] = a4.z * b4.z; abT[(s4 + 3) * Lq + tok] = a4.w * b4.w;
        }
        __syncthreads();                                 // (drains vmcnt too: the image has landed)
        if (probe) pa = (long long)__builtin_amdgcn_s_memtime();
        // scores[n][K] = abT^T . O^T on the f32 matrix cores (v_mfma_f32_16x16x4_f32: the ascending-s fmaf chain of K2, same
        // bits): units of one 16-token block x two 16-tag blocks (shared A fragments from the staged products, independent
        // accumulators), B fragments from the matrix-core image of the output matrix (OTm, K2: staged in LDS by LDS-DMA when
        // it fits the transition table's area, else from L2), two state groups ahead.  At most eight wavefronts take units
        // -- two per SIMD (wavefront w sits on SIMD w % 4, HW_ID).  Measured: 12.9 k cycles for the config-3 sequence against
        // 14 k for r02a's 4 x 4 VALU blocking (1 136 FMAs per lane on nine wavefronts, three of them on one SIMD); the
        // matrix cores are not the bound (720 MFMAs = 5.8 k cycles over four SIMDs), the 4-way bank conflicts of the LDS waits are.
        {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const int lr = lane & 15, lk = lane >> 4;
            const int c16 = p.c16, ntb = (n + 15) >> 4, ncb = (K + 15) >> 4, nblk = ntb * ncb;
            const int ngw = nwaves < 8 ? nwaves : 8;
            const int clamp_col = K - 3;                 // model_decompose.py:353
            auto product = [&](auto otm) {               // otm: this lane's entry of the image, typed LDS or global pointer
            const int ncp = (ncb + 1) >> 1;              // a unit = one token block x TWO column blocks: shared A fragments,
#pragma unroll 1                                         // two independent accumulators
            for (int unit = wu; unit < ntb * ncp && wu < ngw; unit += ngw) {
                const int tb = unit / ncp, cb0 = 2 * (unit - tb * ncp), cb1 = cb0 + 1 < ncb ? cb0 + 1 : cb0;
                auto bp0 = otm + cb0 * c16 * 64, bp1 = otm + cb1 * c16 * 64;
                lds_cfloat *ap = (lds_cfloat *)abT + tb * 16 + lr;
                auto a_group = [&](int g, float (&a)[4]) {
                    const int gc = g < c16 ? g : c16 - 1;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int s = 16 * gc + 4 * e + lk;                   // this lane's k of k-step 4g + e
                        a[e] = ap[(s < SP ? s : SP - 1) * Lq];                // (no product row there: B is zero, any finite A will do)
                    }
                };
                auto b_group = [&](decltype(bp0) bp, int g) -> f32x4 { const v4f t = bp[(g < c16 ? g : c16 - 1) * 64]; return f32x4{t.x, t.y, t.z, t.w}; };
                f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                auto mfma8 = [&](const float (&a)[4], const f32x4 &b0, const f32x4 &b1) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b1.w, acc1, 0, 0, 0);
                };
                f32x4 be0 = b_group(bp0, 0), be1 = b_group(bp1, 0), bo0 = b_group(bp0, 1), bo1 = b_group(bp1, 1);
                float ae[4], ao[4];
                a_group(0, ae);
#pragma unroll 1
                for (int g = 0; g < c16; g += 2) {
                    a_group(g + 1, ao);
                    mfma8(ae, be0, be1);
                    be0 = b_group(bp0, g + 2); be1 = b_group(bp1, g + 2);
                    a_group(g + 2, ae);
                    if (g + 1 < c16) mfma8(ao, bo0, bo1);
                    bo0 = b_group(bp0, g + 3); bo1 = b_group(bp1, g + 3);
                }
                auto store = [&](const f32x4 &acc, int cb) {                  // rows lk*4 + r of the token block, column lr
                    const float av[4] = {acc.x, acc.y, acc.z, acc.w};
                    const int col = cb * 16 + lr;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int tok = tb * 16 + lk * 4 + r;
                        float v = av[r] + 0.0f;                               // -0.0 -> +0.0 like score_tile_kernel
                        if (col == clamp_col) v = fminf(v, p.threshold);
                        if (tok < n && col < Kp) scl[(long long)tok * Kp + col] = v;
                    }
                };
                store(acc0, cb0);
                if (cb1 != cb0) store(acc1, cb1);
            }
            };
            if (otm_lds) product((lds_cv4f *)((lds_cfloat *)trl) + lane);
            else product((glb_cv4f *)p.OTm + lane);
        }
        __syncthreads();                                 // abT (aliasing hist) is free again
        if (probe) pb = (long long)__builtin_amdgcn_s_memtime();
    }
    // ---- who does what: full wavefronts own eight tag pairs each (eight lanes per pair); the pairs left over go to one
    // TAIL wavefront that spreads them over all its lanes (GT = 64, 32, 16 or 8 lanes per pair, sources at stride GT), so
    // that e.g. K = 130 (64 pairs + START/STOP) costs the ninth wavefront 3 source slots per lane instead of 17
    const int npairs = (K + 1) >> 1, nfull = npairs >> 3, rem = npairs & 7;
    const bool tail = wu >= nfull;                       // wave-uniform
    const int GT = rem <= 1 ? 64 : rem == 2 ? 32 : rem <= 4 ? 16 : 8;
    const int g = tail ? (lane & (GT - 1)) : (tid & 7);  // lane of its group
    const int grp = tail ? lane / GT : 0;
    const int pair = tail ? nfull * 8 + grp : (tid >> 3);
    const int j0 = 2 * pair;
    const bool own0 = j0 < K && (!tail || grp < rem), own1 = own0 && j0 + 1 < K;
    const bool writer = own0 && (tail ? g == GT - 1 : g == 0);
    const int XS = (K - 8 * IB + 7) >> 3;                // full wavefronts: leftover source slots per lane (0..4)
    const int nst = (K + GT - 1) / GT;                   // tail wavefront: source slots per lane (<= IB + 4)
    constexpr int NSL = IB + 4;
    v2f trs[NSL];                                        // tr[i][j0], tr[i][j0+1] of this lane's sources
#define FARNN_TRS_SET(SL, X, Y) do { trs[SL] = v2f{(X), (Y)}; } while (0)
    int ixs[4];                                          // full: leftover sources (clamped into the row; their tr is -inf)
    {
        const float *row0 = p.trT + (long long)(own0 ? j0 : 0) * Kp, *row1 = p.trT + (long long)(own1 ? j0 + 1 : 0) * Kp;
        if (!tail) {
#pragma unroll
            for (int k4 = 0; k4 < IB4; k4++) {           // sources 32*k4 + 4*g + u: a 128-byte span per read, no bank conflicts
                const float4 a = ld4(row0 + k4 * 32 + g * 4), c = ld4(row1 + k4 * 32 + g * 4);
                trs[k4 * 4 + 0] = v2f{own0 ? a.x : ninf, own1 ? c.x : ninf}; trs[k4 * 4 + 1] = v2f{own0 ? a.y : ninf, own1 ? c.y : ninf};
                trs[k4 * 4 + 2] = v2f{own0 ? a.z : ninf, own1 ? c.z : ninf}; trs[k4 * 4 + 3] = v2f{own0 ? a.w : ninf, own1 ? c.w : ninf};
            }
#pragma unroll
            for (int xk = 0; xk < 4; xk++) {
                const int i = 8 * IB + 8 * xk + g;
                const bool ok = i < K;
                ixs[xk] = ok ? i : K - 1;
                FARNN_TRS_SET(IB + xk, (ok && own0) ? row0[ixs[xk]] : ninf, (ok && own1) ? row1[ixs[xk]] : ninf);
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < NSL; sl++) {
                const int i = g + GT * sl;
                const bool ok = i < K;
                FARNN_TRS_SET(sl, (ok && own0) ? row0[ok ? i : K - 1] : ninf, (ok && own1) ? row1[ok ? i : K - 1] : ninf);
            }
#pragma unroll
            for (int xk = 0; xk < 4; xk++) ixs[xk] = 0;
        }
    }
    const v2f t_start = writer ? v2f{p.trT[(long long)j0 * Kp + START], own1 ? p.trT[(long long)(j0 + 1) * Kp + START] : 0.0f}
                               : v2f{0.f, 0.f};                                                  // before the table DMA
    int my_tr = 0;
    {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)trl);
        for (int k = wu; k < tr_pieces; k += nwaves, my_tr++)
            lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, reinterpret_cast<const char *>(p.trT), lds0 + (unsigned)k * 1024u);
    }
    if (PW > K)
        for (int i = tid; i < n * (PW - K); i += nthreads) hist[(i / (PW - K)) * PW + K + i % (PW - K)] = ninf;
    wait_vmcnt(my_tr);                                   // scores + this lane's transition entries landed
    wg_barrier_lds();                                    // (a __syncthreads would drain the table DMA)
    // scores of this lane's two tags (a tag beyond K reads a finite pad column and its transitions are -inf)
    const float *fcol = scl + (own0 ? j0 : 0);
    auto scores_at = [&](int t) {                        // f_t[j0], f_t[j0+1]; nothing is read into a half without a tag
        v2f f = *reinterpret_cast<const v2f *>(fcol + (size_t)t * Kp);
        f.y = own1 ? f.y : 0.0f;
        return f;
    };
    if (writer) {                                                             // crf.py:135
        const v2f f0 = scores_at(0);
        hist[j0] = f0.x + t_start.x;
        if (own1) hist[j0 + 1] = f0.y + t_start.y;
    }
    wg_barrier_lds();
    if (probe) pc1 = (long long)__builtin_amdgcn_s_memtime();
    // the tail wavefront's step is a short latency chain (LDS read, a few adds, six DPP levels, LDS write): at the default
    // priority the SIMD's older wavefronts starve it until their own issue-bound step is over and the chain then runs
    // behind them (K = 130: +330 cycles per step); first in line, it hides inside their step
    if (tail) __builtin_amdgcn_s_setprio(3);
    // One step: two adds (crf.py:123,145) and half a v_max3 per (source, tag).  The step is VALU-issue bound on the CU:
    // hoisting the partition-independent add (f + tr) behind the step's LDS write was measured and bought nothing (the
    // idle time around the write and the barrier is ~100 cycles, not the ~350 the first probe suggested), and
    // v_pk_add_f32 issues at half rate on gfx950, so the packed form saves registers and LDS reads, not issue slots.
    // Two loops, the full wavefronts' specialised on its number of leftover slots: with the tail / leftover / ablation tests
    // inside one loop a step spent ~250 cycles on a dozen scalar branches (a taken branch refills the instruction buffer).
    const unsigned pin_lane = (unsigned)(size_t)(hist + g * 4);               // this lane's 16 bytes of a 32-source block
    const unsigned f_lane = (unsigned)(size_t)fcol;
    unsigned px_lane[4];
#pragma unroll
    for (int xk = 0; xk < 4; xk++) px_lane[xk] = (unsigned)(size_t)(hist + ixs[xk]);
    auto publish = [&](int t, const v2f &best) {
        if (writer) *reinterpret_cast<v2f *>(hist + (size_t)t * PW + j0) = best;   // (j0 + 1 == K: -inf into the pad)
        wg_barrier_lds();
    };
    auto full_steps = [&](auto xs_c) {
        constexpr int XSC = decltype(xs_c)::value;       // leftover source slots per lane (0..4)
        v2f fnext = n > 1 ? scores_at(1) : v2f{0.f, 0.f};
        for (int t = 1; t < n; t++) {
            const v2f f = fnext;
            auto f_tr = [&](int sl) { return f + trs[sl]; };
            v2f best = v2f{ninf, ninf};
            // every LDS read of the step up front, oldest first: the scores of the NEXT step, the IB4 blocks, the leftovers
            // (left to the compiler they went through one recycled register quad: three exposed LDS round trips per step)
            const unsigned row = (unsigned)((t - 1) * PW) * 4u;
            const unsigned frow = (unsigned)((t + 1 < n ? t + 1 : t) * Kp) * 4u;
            v4f p4[IB4 > 0 ? IB4 : 1];
            float px[XSC > 0 ? XSC : 1];
            asm volatile("ds_read_b64 %0, %1" : "=v"(fnext) : "v"(f_lane + frow));
            if constexpr (IB4 > 0) lds_read16_at<0>(p4[0], pin_lane + row);
            if constexpr (IB4 > 1) lds_read16_at<128>(p4[1], pin_lane + row);
            if constexpr (IB4 > 2) lds_read16_at<256>(p4[2], pin_lane + row);
            if constexpr (IB4 > 3) lds_read16_at<384>(p4[3], pin_lane + row);
            if constexpr (IB4 > 4) lds_read16_at<512>(p4[4], pin_lane + row);
            if constexpr (IB4 > 5) lds_read16_at<640>(p4[5], pin_lane + row);
            if constexpr (IB4 > 6) lds_read16_at<768>(p4[6], pin_lane + row);
#pragma unroll
            for (int xk = 0; xk < XSC; xk++) asm volatile("ds_read_b32 %0, %1" : "=v"(px[xk]) : "v"(px_lane[xk] + row));
            auto block = [&](int k4) {
                const v2f v0 = f_tr(k4 * 4 + 0) + v2f{p4[k4].x, p4[k4].x}, v1 = f_tr(k4 * 4 + 1) + v2f{p4[k4].y, p4[k4].y};
                const v2f v2 = f_tr(k4 * 4 + 2) + v2f{p4[k4].z, p4[k4].z}, v3 = f_tr(k4 * 4 + 3) + v2f{p4[k4].w, p4[k4].w};
                best.x = fmaxf(fmaxf(best.x, v0.x), v1.x); best.y = fmaxf(fmaxf(best.y, v0.y), v1.y);
                best.x = fmaxf(fmaxf(best.x, v2.x), v3.x); best.y = fmaxf(fmaxf(best.y, v2.y), v3.y);
            };
            if constexpr (IB4 > 0) { lds_wait_for<IB4 + XSC - 1>(p4[0]); block(0); }
            if constexpr (IB4 > 1) { lds_wait_for<IB4 + XSC - 2>(p4[1]); block(1); }
            if constexpr (IB4 > 2) { lds_wait_for<IB4 + XSC - 3>(p4[2]); block(2); }
            if constexpr (IB4 > 3) { lds_wait_for<IB4 + XSC - 4>(p4[3]); block(3); }
            if constexpr (IB4 > 4) { lds_wait_for<IB4 + XSC - 5>(p4[4]); block(4); }
            if constexpr (IB4 > 5) { lds_wait_for<IB4 + XSC - 6>(p4[5]); block(5); }
            if constexpr (IB4 > 6) { lds_wait_for<IB4 + XSC - 7>(p4[6]); block(6); }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fnext));
#pragma unroll
            for (int xk = 0; xk < XSC; xk++) {
                asm volatile("" : "+v"(px[xk]));         // (behind the wait)
                const v2f v = f_tr(IB + xk) + v2f{px[xk], px[xk]};
                best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y);
            }
            fnext.y = own1 ? fnext.y : 0.0f;             // (scores_at's rule: nothing is read into a half without a tag)
            // the eight lanes of the group: xor 1, xor 2 inside the quad, then the mirrored quad of the half row
            asm volatile("s_nop 1\n\t"
                         "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                         "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1"
                         : "+v"(best.x), "+v"(best.y));
            publish(t, best);
        }
    };
    auto tail_steps = [&]() {
        for (int t = 1; t < n; t++) {
            const float *pin = hist + (size_t)(t - 1) * PW;
            const v2f f = scores_at(t);
            auto f_tr = [&](int sl) { return f + trs[sl]; };
            v2f best = v2f{ninf, ninf};
#pragma unroll
            for (int s4 = 0; s4 < NSL; s4 += 4) {        // four slots at a time (slots beyond nst: -inf transitions)
                if (s4 >= nst) continue;
                float ps[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = g + GT * (s4 + u); ps[u] = pin[i < PW ? i : PW - 1]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const v2f v = f_tr(s4 + u) + v2f{ps[u], ps[u]};
                    best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y);
                }
            }
            // max scan over the GT lanes of the group: its last lane ends up with the group's maximum
#define FARNN_SCAN2(CTRL) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %1, %1, %1 " CTRL "\n\ts_nop 1" : "+v"(best.x), "+v"(best.y))
            FARNN_SCAN2("row_shr:1 row_mask:0xf bank_mask:0xf");
            FARNN_SCAN2("row_shr:2 row_mask:0xf bank_mask:0xf");
            FARNN_SCAN2("row_shr:4 row_mask:0xf bank_mask:0xf");
            if (GT >= 16) FARNN_SCAN2("row_shr:8 row_mask:0xf bank_mask:0xf");
            if (GT >= 32) FARNN_SCAN2("row_bcast:15 row_mask:0xa bank_mask:0xf");
            if (GT >= 64) FARNN_SCAN2("row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef FARNN_SCAN2
            publish(t, best);
        }
    };
    if (tail) tail_steps();
    else switch (XS) {
        case 0: full_steps(std::integral_constant<int, 0>{}); break;
        case 1: full_steps(std::integral_constant<int, 1>{}); break;
        case 2: full_steps(std::integral_constant<int, 2>{}); break;
        case 3: full_steps(std::integral_constant<int, 3>{}); break;
        default: full_steps(std::integral_constant<int, 4>{}); break;
    }
    if (tail) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the transition table is in LDS
    __syncthreads();
    if (probe) pc2 = (long long)__builtin_amdgcn_s_memtime();
    if (w == 0 && n > 0) {
        // One wavefront walks the path, and a lone wavefront issues one instruction every ~4-5 cycles: the step is bound by its
        // instruction COUNT plus one LDS round trip (r02: eight 4-byte reads, a six-level DPP max, three ballots -- 590 cycles).
        // Lane q holds FOUR consecutive candidates (4q .. 4q+3; PW / 4 <= 64 lanes cover a row): the table's row and the
        // partitions' row are one 16-byte read each, the latter fetched a step ahead.  There is NO reduction: the maximum of a
        // step's candidates IS part_t[ptr], which the forward pass took over exactly these values (a max returns one of its
        // operands' bits) -- two broadcast reads fetch it and the score f_t[ptr] beside the table's row.
        const int nq = PW >> 2, q = lane < nq ? lane : nq - 1;
        const unsigned long long valid = nq >= 64 ? ~0ull : ((1ull << nq) - 1ull);
        const float *hq = hist + 4 * q, *tq = trl + 4 * q;
        auto first_equal = [&](const float4 &c, float m) {                    // first index (torch.max's rule); 0 if none (NaN)
            const bool h0 = c.x == m, h1 = c.y == m, h2 = c.z == m, h3 = c.w == m;
            const int e = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3;                       // this lane's first hit, if it has one
            const unsigned long long any = (__ballot(h0) | __ballot(h1) | __ballot(h2) | __ballot(h3)) & valid;
            const int l = __builtin_ctzll(any | (1ull << 63));
            return any ? 4 * l + __builtin_amdgcn_readlane(e, l) : 0;
        };
        auto candidates = [&](float f, const float4 &tr, const float4 &pp) {  // (feat + trans) + partition: crf.py:123,145
            return make_float4((f + tr.x) + pp.x, (f + tr.y) + pp.y, (f + tr.z) + pp.z, (f + tr.w) + pp.w);
        };
        float4 prv = *reinterpret_cast<const float4 *>(hq + (n - 1) * PW);
        float4 c = candidates(0.0f, *reinterpret_cast<const float4 *>(tq + STOP * Kp), prv);      // crf.py:168-169 (0 + x = x)
        // (candidates in [K, PW) are -inf through the partitions' pads; lanes beyond the row repeat its last four)
        int ptr = first_equal(c, wave_max_dpp(fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w))));
        prv = *reinterpret_cast<const float4 *>(hq + (n > 1 ? n - 2 : 0) * PW);                    // part_{t-1} of the first step
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const unsigned hq0 = (unsigned)(size_t)hq, tq0 = (unsigned)(size_t)tq, h0 = (unsigned)(size_t)hist;
        const unsigned sc_off = (unsigned)(size_t)scl - h0;
        int mytag = 0;                                   // lane t % 64 keeps the tag of position t until the next flush
        for (int t = n - 1; t >= 0; t--) {
            {
                const int tag = (ptr == K - 3) ? p.o_idx : ptr;               // model_decompose.py:356
                const int tl = __builtin_amdgcn_readfirstlane(t & 63), tg = __builtin_amdgcn_readfirstlane(tag);
                asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(mytag) : "s"(tg), "s"(tl) : "m0");   // (one SGPR per instruction)
            }
            if ((t & 63) == 0) {                         // 64 positions at a time, coalesced
                if (t + lane < n) {
                    if (p.tags) p.tags[(long long)b * p.L + t + lane] = mytag;
                    if (p.flat) p.flat[foff + t + lane] = mytag;
                }
            }
            // the back-pointer of step t at tag ptr (crf.py:147-149), recomputed: same f32 expression, same values
            if (t > 0) {
                const int tp = __builtin_amdgcn_readfirstlane(t > 1 ? t - 2 : 0);
                const unsigned a_pre = hq0 + 4u * (unsigned)(tp * PW), a_tr = tq0 + 4u * (unsigned)(ptr * Kp);
                const unsigned a_m = h0 + 4u * (unsigned)(t * PW + ptr), a_f = a_m + sc_off;     // (PW == Kp)
                f32x4 pre, tr;
                float m, f;
                // one statement, so that the order is this one: the row fetched ahead first (its latency hides behind the others'
                // -- LDS reads return in order), then the three the step waits for
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(pre), "=&v"(tr), "=&v"(m), "=&v"(f) : "v"(a_pre), "v"(a_tr), "v"(a_m), "v"(a_f) : "memory");
                c = candidates(f, make_float4(tr.x, tr.y, tr.z, tr.w), prv);
                ptr = first_equal(c, m);
                prv = make_float4(pre.x, pre.y, pre.z, pre.w);
            }
        }
    }
    if (probe && tid == 0) {
        pc3 = (long long)__builtin_amdgcn_s_memtime();
        printf("viterbi wg %d (%d positions, %d threads): set-up + scores %lld cycles (products staged at %lld, scores done at %lld), forward pass %lld (%lld per step), backtrace %lld (%lld per step)\n",
               b, n, nthreads, pc1 - pc0, pa - pc0, pb - pc0, pc2 - pc1, (pc2 - pc1) / (n > 1 ? n - 1 : 1), pc3 - pc2, (pc3 - pc2) / n);
    }
    if (p.tags)
        for (int i = n + tid; i < p.L; i += nthreads) p.tags[(long long)b * p.L + i] = -1;   // pads (LOCAL and FULL)
}

// block size (in float4s) of viterbi_kernel (the back-pointer variant) for K tags: ceil(ceil(K/4)/4) rounded up to a built one
inline int viterbi_ib4(int K) {
    const int need = ((K + 3) / 4 + 3) / 4;
    return need <= 2 ? 2 : need <= 4 ? 4 : need <= 9 ? 9 : need <= 13 ? 13 : 16;
}
inline size_t viterbi_lds_bytes(int K, int Kp, int L) {
    return (size_t)2 * 16 * viterbi_ib4(K) * 4 + (size_t)L * Kp * 4 + (size_t)L * Kp * 2;
}
// viterbi_hist_kernel: contiguous float4 blocks per lane (K = 32*IB4 + leftovers) and its thread count
inline int viterbi_hist_ib4(int K) { return K / 32; }
inline int viterbi_hist_threads(int K) { return round_up(((K + 1) / 2) * 8, 64); }
inline size_t viterbi_hist_lds_bytes(int K, int Kp, int SP, int L, bool fused) {
    return viterbi_hist_floats(Kp, SP, L, fused) * 4 + ((size_t)L * Kp * 4 + 1023) / 1024 * 1024 +
           ((size_t)K * Kp * 4 + 1023) / 1024 * 1024;
}

// Batch preparation (one workgroup; B is a batch size, not a corpus):
//   offs[B+1]  exclusive prefix sum of lengths -> where each sequence starts in the flat output
//   order[B]   launch order of the chain kernel: sequences sorted by length (descending, counting
//              sort) and folded so that block i and block i + B/2 pair a long with a short one.
//              Workgroups that share a CU share its memory path; on a ragged batch the unsorted
//              order left the longest chains co-located (58 us vs 44 us for the same work).
__global__ void __launch_bounds__(1024)
batch_prep_kernel(const int64_t *len, int64_t *offs, int *order, int B, int L) {
    extern __shared__ __align__(16) int prep_smem[];
    __shared__ long long sums[1024];
    const int tid = threadIdx.x;
    if (offs) {
        const int per = (B + 1023) / 1024;
        const int lo = tid * per, hi = min(lo + per, B);
        long long s = 0;
        for (int i = lo; i < hi; i++) s += len[i];
        sums[tid] = s;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            long long v = (tid >= off) ? sums[tid - off] : 0;
            __syncthreads();
            sums[tid] += v;
            __syncthreads();
        }
        long long run = (tid == 0) ? 0 : sums[tid - 1];
        for (int i = lo; i < hi; i++) { offs[i] = run; run += len[i]; }
        if (tid == 1023) offs[B] = sums[1023];
    }
    if (order) {
        int *cnt = prep_smem;                 // [L+2] histogram, then bucket cursors
        for (int i = tid; i < L + 2; i += 1024) cnt[i] = 0;
        __syncthreads();
        for (int i = tid; i < B; i += 1024) {
            int n = (int)len[i];
            n = n < 0 ? 0 : (n > L ? L : n);
            atomicAdd(&cnt[n], 1);
        }
        __syncthreads();
        // exclusive scan over the buckets in DESCENDING length order (bucket L first); thread `tid`
        // owns `per` consecutive positions of that order
        {
            const int nb = L + 1, per = (nb + 1023) / 1024;
            const int lo = tid * per, hi = min(lo + per, nb);
            long long s2 = 0;
            for (int r = lo; r < hi; r++) s2 += cnt[L - r];
            __syncthreads();
            sums[tid] = s2;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) {
                long long v = (tid >= off) ? sums[tid - off] : 0;
                __syncthreads();
                sums[tid] += v;
                __syncthreads();
            }
            int run = (tid == 0) ? 0 : (int)sums[tid - 1];
            for (int r = lo; r < hi; r++) { const int c2 = cnt[L - r]; cnt[L - r] = run; run += c2; }
        }
        __syncthreads();
        const int half = B / 2;
        for (int i = tid; i < B; i += 1024) {
            int n = (int)len[i];
            n = n < 0 ? 0 : (n > L ? L : n);
            const int pos = atomicAdd(&cnt[n], 1);                    // rank in descending order
            const int slot = pos < half ? pos : half + (B - 1 - pos);  // fold the shorter half
            order[slot] = i;
        }
    }
}

// Small-batch variant (B <= 1024): one pass, one barrier.  Thread i ranks its own sequence against
// all others by (length desc, index asc) and sums the lengths before it, reading the B lengths as
// LDS broadcasts -- O(B) per thread instead of two block scans with ~40 barriers (4.6 us -> ~1 us
// at B = 256, and this kernel sits in front of the chain kernel on every call).
__global__ void __launch_bounds__(1024)
batch_prep_small_kernel(const int64_t *len, int64_t *offs, int *order, int B, int L, int G) {
    // G (power of two, 1..16) adjacent lanes share one sequence and split the j range
    __shared__ __align__(16) int ls[1024];
    const int tid = threadIdx.x;
    for (int k = tid; k < 1024; k += blockDim.x) {
        int v = -1;                                  // padding entries: shorter than anything
        if (k < B) {
            const int n = (int)len[k];
            v = n < 0 ? 0 : (n > L ? L : n);
        }
        ls[k] = v;
    }
    __syncthreads();
    const int i = tid / G, part = tid - i * G;
    const bool live = i < B;
    const int mine = live ? ls[i] : -1;
    int rank = 0;
    unsigned before = 0;                             // B * L < 2^31 for any batch this kernel sees
    const int4 *ls4 = reinterpret_cast<const int4 *>(ls);
    const int n4 = (B + 3) >> 2;
    const int per = (n4 + G - 1) / G, lo = part * per, hi = min(lo + per, n4);
#pragma unroll 4
    for (int j4 = lo; j4 < hi; j4++) {
        const int4 v = ls4[j4];                      // 4 lengths per LDS read
        const int lj[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j4 * 4 + u;
            rank += (lj[u] > mine) || (lj[u] == mine && j < i);
            before += (j < i && lj[u] > 0) ? (unsigned)lj[u] : 0u;
        }
    }
    for (int off = 1; off < G; off <<= 1) {          // combine the G partial results
        rank += __shfl_xor(rank, off, WAVE);
        before += __shfl_xor(before, off, WAVE);
    }
    if (!live || part != 0) return;
    if (offs) {
        offs[i] = (long long)before;
        if (i == B - 1) offs[B] = (long long)before + mine;
    }
    if (order) {
        const int half = B / 2;
        order[rank < half ? rank : half + (B - 1 - rank)] = i;          // fold the shorter half
    }
}

}  // namespace farnn
