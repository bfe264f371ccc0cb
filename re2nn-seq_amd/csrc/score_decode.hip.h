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
#include "viterbi_hist.hip.h"

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
    const long long foff_v = p.offs ? p.offs[b] : (p.flat ? flat_offset_in_kernel(p.len, b, p.L, tid, nthreads) : 0);
    // (the same value in every lane: kept in scalar registers over the forward pass)
    const long long foff = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(foff_v >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)foff_v));
    const int START = K - 2, STOP = K - 1;
    const float ninf = -INFINITY;

    // (the set-up loops stay rolled and index in 32 bits: unrolled by four with 64-bit addresses they took registers the 128 of a
    // sixteen-wavefront workgroup do not have beside the IB transition entries -- r03: 5 / 1 VGPRs in scratch at IB4 = 13 / 16)
#pragma unroll 1
    for (int i = tid * 4; i < n * Kp; i += nthreads * 4) st4(scl + i, ld4(sc + i));
#pragma unroll 1
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
#pragma unroll 1
    for (int jj = tid; jj < K; jj += nthreads)
        part[jj] = scl[jj] + p.trT[jj * Kp + START];                         // crf.py:135 (K * Kp < 2^17)
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
        {   // the quad: lanes xor 1, then xor 2 (DPP quad_perm: no address registers, no LDS crossbar)
            const float ov1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, best), 0xB1, 0xf, 0xf, false));
            const int oi1 = __builtin_amdgcn_update_dpp(0, bi, 0xB1, 0xf, 0xf, false);
            argmax_combine(best, bi, ov1, oi1);
            const float ov2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, best), 0x4E, 0xf, 0xf, false));
            const int oi2 = __builtin_amdgcn_update_dpp(0, bi, 0x4E, 0xf, 0xf, false);
            argmax_combine(best, bi, ov2, oi2);
        }
        if (owner && q == 0) {
            pout[j] = best;
            bp[(long long)t * Kp + j] = (unsigned short)(bi >= K ? 0 : bi);
        }
        __syncthreads();
        pc ^= 1;
    }
    if (w == 0) {
        int lane;                                        // re-derived here: at IB4 = 16 the copy made at the kernel's top was the one
        asm volatile("v_and_b32 %0, 63, %1" : "=v"(lane) : "v"(tid));   // register too many over the forward pass (r03: in scratch)
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

// block size (in float4s) of viterbi_kernel (the back-pointer variant) for K tags: ceil(ceil(K/4)/4) rounded up to a built one
inline int viterbi_ib4(int K) {
    const int need = ((K + 3) / 4 + 3) / 4;
    return need <= 2 ? 2 : need <= 4 ? 4 : need <= 9 ? 9 : need <= 13 ? 13 : 16;
}
inline size_t viterbi_lds_bytes(int K, int Kp, int L) {
    return (size_t)2 * 16 * viterbi_ib4(K) * 4 + (size_t)L * Kp * 4 + (size_t)L * Kp * 2;
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

// ---- K2l: the stand-alone score + decode launch when the output matrix is a LABEL MAP (label_map.hip.h) and only tags are asked
// for (round 5).  score_tile_kernel stages products, runs [32 x S].[S x K] on the matrix cores and decodes K columns per token:
// 10 us for the headline batch.  A label map needs S multiply-adds and one segmented scan per token: one workgroup per sequence,
// wavefront w takes the token pairs w, w + LMS_WAVES, ...; both state rows come straight from the stash (L2-resident: the
// recurrence kernel has just written it), the next pair's entries are in flight while the current one is scanned.  No LDS but the
// flat offset, no barrier but the one behind it.  Reference: model_onehot.py:411-426 (scores), :162-180 (decode).
constexpr int LMS_WAVES = 16;

__global__ void __launch_bounds__(LMS_WAVES * 64)
label_map_score_kernel(const ScoreParams p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int L = p.L, SP = p.SP;
    const int len = clamp_len(p.len[b], L);
    const int nsteps = p.full ? L : len;
    __shared__ int foff_s;
    if (w == 0) {                                          // the sequence's flat offset (utils.py:153-164)
        int partsum = 0;
        if (p.flat) {
            if (p.offs) partsum = lane == 0 ? (int)p.offs[b] : 0;
            else for (int j = lane; j < b; j += WAVE) partsum += clamp_len(p.len[j], L);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) partsum += __shfl_xor(partsum, off, WAVE);
        if (lane == 0) foff_s = partsum;
    } else if (w == 1 && p.tags) {                         // pad positions of LOCAL mode: tag -1
        for (int i = nsteps + lane; i < L; i += WAVE) p.tags[(long long)b * L + i] = -1;
    }
    LabelMapRegs lr;
    lm_load(p.lm, lane, lr);
    const bool two = p.lm.nq > 1;
    const float *A = p.A + (long long)b * (L + 1) * SP, *Bk = p.Bk + (long long)b * (L + 1) * SP;
    auto rowB = [&](int i) { return (i + 1 <= len) ? len - (i + 1) : i + 1; };    // beta of token i (pads of FULL mode: row i + 1)
    // this wavefront's first pair's entries
    float fa0 = 0.f, fa1 = 0.f, ba0 = 0.f, ba1 = 0.f, fb0 = 0.f, fb1 = 0.f, bb0 = 0.f, bb1 = 0.f;
    auto fetch = [&](int q) {
        const int ia = q, ib = q + 1 < nsteps ? q + 1 : q;
        const float *fa = A + (long long)(ia + 1) * SP, *fb = A + (long long)(ib + 1) * SP;
        const float *ba = Bk + (long long)rowB(ia) * SP, *bb = Bk + (long long)rowB(ib) * SP;
        fa0 = fa[lr.st0]; ba0 = ba[lr.st0]; fb0 = fb[lr.st0]; bb0 = bb[lr.st0];
        if (two) { fa1 = fa[lr.st1]; ba1 = ba[lr.st1]; fb1 = fb[lr.st1]; bb1 = bb[lr.st1]; }
    };
    int q = 2 * w;
    if (q < nsteps) fetch(q);
    __syncthreads();
    const long long foff = foff_s;
    for (; q < nsteps; q += 2 * LMS_WAVES) {
        const bool hasb = q + 1 < nsteps;
        const int ia = q, ib = hasb ? q + 1 : q;
        const float xa0 = fa0 * ba0, xa1 = two ? fa1 * ba1 : 0.0f, xb0 = fb0 * bb0, xb1 = two ? fb1 * bb1 : 0.0f;
        if (q + 2 * LMS_WAVES < nsteps) fetch(q + 2 * LMS_WAVES);              // the next pair's entries, under this pair's scan
        float ya0, ya1, yb0, yb1;
        lm_scan_scores2(lr, xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1);
        float ma = fmaxf(ya0, ya1), mb = fmaxf(yb0, yb1);
        wave_max_dpp2(ma, mb);
        const int taga = lm_tag_from_candidates(p.lm, lr, ya0, ya1, ma, p.K, p.o_idx);
        const int tagb = lm_tag_from_candidates(p.lm, lr, yb0, yb1, mb, p.K, p.o_idx);
        if (lane < (hasb ? 2 : 1)) {
            const int i = lane ? ib : ia;
            const int tag = lane ? tagb : taga;
            if (p.tags) p.tags[(long long)b * L + i] = tag;
            if (p.flat && i < len) p.flat[foff + i] = tag;
        }
    }
}

}  // namespace farnn
