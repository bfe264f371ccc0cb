// Scoring of the decomposed independent=1 model, FARNN_S_D_W_I.get_final_score
// (reference model_decompose_independent.py:199-207), one workgroup per token:
//
//     bss[i][j] = sum_r (S1[i,r] S2[j,r]) v_r + W[i][j]
//     abw[i][j] = (a_i b~_j) * bss[i][j]                     a = state BEFORE token i (:262)
//     br[q]     = sum_{i,j} abw[i][j] S1o[i,q] S2o[j,q]
//     score[c]  = sum_q br[q] Cout[c,q]
//
// followed by the shared epilogue (priority, optional score output, clamp, argmax or hand-off to
// the Viterbi kernel).  Secondary path of SURVEY.md 8a (row a15): written for correctness, the
// S*S*(R+RO) flops per token are done with plain loops against L2-resident factors.
#pragma once
#include "common.hip.h"

namespace farnn {

struct Decomp1ScoreParams {
    const float *A, *Bk;            // stash [B][L+1][SP]
    const float *Vgen;              // [V][Rp]
    const float *S1, *S2;           // [S][Rp]
    const float *W;                 // [S][SP]
    const float *S1o, *S2o;         // [S][ROp]
    const float *CoutT;             // [RO][Kc]
    const float *P;                 // [K][Kc] or nullptr
    const int64_t *x, *len, *offs;
    int32_t *tags; int64_t *flat; float *scores; float *crf_scores;
    int B, L, S, SP, R, Rp, RO, ROp, K, Kp, Kc, V;
    int full, use_crf, o_idx;
    float threshold;
};

__global__ void __launch_bounds__(256)
decomp1_score_kernel(const Decomp1ScoreParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, nt = blockDim.x;
    const int i = blockIdx.x, b = blockIdx.y;
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, R = p.R, Rp = p.Rp, RO = p.RO, ROp = p.ROp, K = p.K;
    if (i >= nsteps) {
        if (p.tags && tid == 0) p.tags[(long long)b * p.L + i] = -1;
        if (p.scores) for (int c = tid; c < K; c += nt) p.scores[((long long)b * p.L + i) * K + c] = 0.0f;
        return;
    }
    float *alpha = smem;                 // [SP]
    float *beta = alpha + SP;            // [SP]
    float *v = beta + SP;                // [Rp]
    float *br = v + Rp;                  // [ROp]
    float *sc = br + ROp;                // [Kc]
    float *sc2 = sc + p.Kc;              // [Kc]
    float *abw = sc2 + p.Kc;             // [S][SP]

    const float *ar = p.A + ((long long)b * (p.L + 1) + i) * SP;
    const int bidx = (i + 1 <= len) ? len - (i + 1) : i + 1;
    const float *brow = p.Bk + ((long long)b * (p.L + 1) + bidx) * SP;
    const float *vg = p.Vgen + (long long)clamp_tok(p.x[(long long)b * p.L + i], p.V) * Rp;
    for (int s = tid; s < SP; s += nt) { alpha[s] = ar[s]; beta[s] = brow[s]; }
    for (int r = tid; r < Rp; r += nt) v[r] = r < R ? vg[r] : 0.0f;
    __syncthreads();
    for (int idx = tid; idx < S * S; idx += nt) {
        const int ii = idx / S, jj = idx - ii * S;
        const float *s1 = p.S1 + (long long)ii * Rp, *s2 = p.S2 + (long long)jj * Rp;
        float d = 0.0f;
        for (int r = 0; r < R; r++) d = fmaf(s1[r] * s2[r], v[r], d);                 // :201
        abw[ii * SP + jj] = (alpha[ii] * beta[jj]) * (d + p.W[(long long)ii * SP + jj]);   // :202-203
    }
    __syncthreads();
    for (int q = tid; q < RO; q += nt) {                                              // :204
        float acc = 0.0f;
        for (int ii = 0; ii < S; ii++) {
            float t = 0.0f;
            const float *arow = abw + ii * SP;
            for (int jj = 0; jj < S; jj++) t = fmaf(arow[jj], p.S2o[(long long)jj * ROp + q], t);
            acc = fmaf(p.S1o[(long long)ii * ROp + q], t, acc);
        }
        br[q] = acc;
    }
    __syncthreads();
    for (int c = tid; c < K; c += nt) {                                               // :205
        float s = 0.0f;
        for (int q = 0; q < RO; q++) s = fmaf(br[q], p.CoutT[(long long)q * p.Kc + c], s);
        sc[c] = s;
    }
    __syncthreads();
    const float *fin = sc;
    if (p.P) {
        for (int d = tid; d < K; d += nt) {
            float s = 0.0f;
            for (int c = 0; c < K; c++) s = fmaf(sc[c], p.P[(long long)c * p.Kc + d], s);
            sc2[d] = s;
        }
        __syncthreads();
        fin = sc2;
    }
    if (p.scores)
        for (int c = tid; c < K; c += nt) p.scores[((long long)b * p.L + i) * K + c] = fin[c];
    const int clamp_col = p.use_crf ? K - 3 : K - 1;
    if (p.use_crf) {
        for (int c = tid; c < K; c += nt) {
            float vv = fin[c] + 0.0f;
            if (c == clamp_col) vv = fminf(vv, p.threshold);
            p.crf_scores[((long long)b * p.L + i) * p.Kp + c] = vv;
        }
    } else if (w == 0) {
        float bv = -INFINITY; int bi = 0x7ffffffe;
        for (int c = lane; c < K; c += WAVE) {
            float vv = fin[c] + 0.0f;
            if (c == clamp_col) vv = fminf(vv, p.threshold);
            if (vv > bv) { bv = vv; bi = c; }
        }
        bi = wave_argmax_dpp(bv, bi);
        if (lane == 0) {
            if (bi >= K) bi = 0;
            const int tag = (bi == K - 1) ? p.o_idx : bi;
            if (p.tags) p.tags[(long long)b * p.L + i] = tag;
            if (p.flat && i < len) p.flat[p.offs[b] + i] = tag;
        }
    }
}

// ---- the same scoring on the f32 matrix cores, from a per-word table -----------------------------------
// bss[x] = sum_r S1 S2 v_x + W depends on the word only: it is built once at create time (0.05 MB per word
// at S=100).  Per token the work is then a real GEMM against weights shared by every token,
//     U[S x RO] = abw[S x S] . S2o[S x RO],   abw_ij = (a_i b~_j) bss[x]_ij,      br_q = sum_i S1o_iq U_iq
// 1.5 MFLOP on v_mfma_f32_16x16x4_f32 (exact f32 fma chains).  Two kernels:
//
// decomp1_br_mfma_kernel -- br for every live position.  Persistent wavefronts (2 per SIMD), each owning a
// contiguous slice of the batch's live tokens (flat offsets), so there is no workgroup-level synchronisation
// after start-up.  (Handing tokens out by ticket cost more than the MFMAs: 16 k atomics on one address
// serialise at ~13 ns each.)
//   * B operand: S2o is the same for every token, so the wavefront keeps ALL of it in registers in MFMA
//     operand order (S2oP[k-group][col tile][lane][4], <= 140 registers) for its whole life -- the inner
//     loop has no LDS or memory operand but the A stream.  (Versions with S2o in LDS ran the matrix cores
//     at 40 %: one LDS read per MFMA, each waited for just before its use.)
//   * A operand: the table is stored in operand order too (BSSp[x][row tile][k-group][lane][4]): a lane's
//     four k-steps are ONE coalesced 16-byte load straight from HBM into the registers the MFMA reads,
//     scaled by a_i b~_j on the way; a row tile's registers are refilled for the next tile as soon as their
//     group has been issued.  All these loads are unconditional (clamped index) and pinned in program
//     order: a load under a branch, or a reordered prologue, makes the compiler wait for vmcnt(0|1) at
//     every use instead of vmcnt(6).
//   * br: S1o in accumulator order (S1oP[row tile][col tile][lane][4]) staged once per workgroup in LDS and
//     read through lgkmcnt, so the tile epilogues never drain the A stream's vmcnt.
// decomp1_label_kernel -- score = br . Cout^T (:205), the priority product, CRF emissions or the argmax, and
// the fill values of the dead positions: a few microseconds, weights staged in LDS per workgroup.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int D1M_MAXNT = 5;            // 16-column tiles of U held in registers: RO <= 80
constexpr int D1M_MAXKQ4 = 7;           // groups of 4 k-steps per row tile: S <= 112 (register budget)

struct Decomp1MfmaParams {
    Decomp1ScoreParams base;
    const float *BSSp;                  // [V][MT][KQ4][64][4]
    const float *S1oP;                  // [MT][NT][64][4]
    const float *S2oP;                  // [KQ4][NT][64][4]
    float *br;                          // [B*L][MT][NT*16] out: per-row-tile partial br
    int MT, NT, KQ4;                    // row tiles, column tiles, groups of 4 k-steps
};

__global__ void pack_bss_operand_kernel(const float *__restrict__ BSS, float *__restrict__ BSSp, long long total,
                                        int S, int SP, int MT, int KQ4) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int u = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const long long blk = idx >> 8;
    const int g = (int)(blk % KQ4), mt = (int)((blk / KQ4) % MT);
    const long long x = blk / ((long long)KQ4 * MT);
    const int i = mt * 16 + (lane & 15), j = 4 * (g * 4 + u) + (lane >> 4);
    BSSp[idx] = (i < S && j < S) ? BSS[(x * S + i) * SP + j] : 0.0f;
}

__global__ void pack_s1o_operand_kernel(const float *__restrict__ S1o, float *__restrict__ S1oP, int total,
                                        int S, int RO, int ROp, int NT) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int r = idx & 3, lane = (idx >> 2) & 63, blk = idx >> 8;
    const int n = blk % NT, mt = blk / NT;
    const int row = mt * 16 + (lane >> 4) * 4 + r, col = n * 16 + (lane & 15);
    S1oP[idx] = (row < S && col < RO) ? S1o[(long long)row * ROp + col] : 0.0f;
}

__global__ void pack_s2o_operand_kernel(const float *__restrict__ S2o, float *__restrict__ S2oP, int total,
                                        int S, int RO, int ROp, int NT) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int u = idx & 3, lane = (idx >> 2) & 63, blk = idx >> 8;
    const int n = blk % NT, g = blk / NT;
    const int j = 4 * (g * 4 + u) + (lane >> 4), col = n * 16 + (lane & 15);
    S2oP[idx] = (j < S && col < RO) ? S2o[(long long)j * ROp + col] : 0.0f;
}

template <int NT>
__global__ void __launch_bounds__(256, 2)
decomp1_br_mfma_kernel(const Decomp1MfmaParams q) {
    const Decomp1ScoreParams &p = q.base;
    extern __shared__ __align__(16) float smem[];
    constexpr int NC = NT * 16;
    const int tid = threadIdx.x, lane = tid & 63, nt_ = blockDim.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = p.S, SP = p.SP;
    const int MT = q.MT, KQ4 = q.KQ4, SPa = MT * 16;
    const int klast = (S + 3) / 4 - (KQ4 - 1) * 4;   // k-steps of 4 in the last group that hold data (1..4)
    float *s1l = smem;                               // [MT][NT][64][4] S1o in accumulator order
    float *al = s1l + MT * NT * 256 + w * 2 * SPa;   // this wavefront's [SPa] a, [SPa] b~
    float *be = al + SPa;

    const int lr = lane & 15, lk = lane >> 4;
    f32x4_t Bq[D1M_MAXKQ4][NT];
#pragma unroll
    for (int g = 0; g < D1M_MAXKQ4; g++)
#pragma unroll
        for (int n = 0; n < NT; n++)
            Bq[g][n] = *(const f32x4_t *)(q.S2oP + ((long long)((g < KQ4 ? g : KQ4 - 1) * NT + n) * 64 + lane) * 4);
    for (int base = 0; base < MT * NT * 64; base += 8 * nt_) {
        f32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = base + u * nt_ + tid;
            v[u] = *(const f32x4_t *)(q.S1oP + (long long)(e < MT * NT * 64 ? e : 0) * 4);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = base + u * nt_ + tid;
            if (e < MT * NT * 64) *(f32x4_t *)(s1l + e * 4) = v[u];
        }
    }
    for (int idx = lane; idx < 2 * SPa; idx += WAVE) al[idx] = 0.0f;           // pads stay 0
    __syncthreads();
    // live tokens in flat order: [offs[b], offs[b+1]) belongs to sequence b (every position in full mode)
    const long long total = p.full ? (long long)p.B * p.L : p.offs[p.B];
    const long long nw = (long long)gridDim.x * 4, wid = (long long)blockIdx.x * 4 + w;
    // the unit of work is one row tile of one token (total * MT units, an equal contiguous run per wavefront:
    // whole tokens would quantise 4.3 tokens per wavefront to 5); a token's tiles may end up on two wavefronts,
    // so every tile writes its own partial br row and the label kernel adds them in tile order
    const long long u0 = total * MT * wid / nw, u1 = total * MT * (wid + 1) / nw;
    const long long f0 = u0 / MT;
    int b = 0;
    long long ob = 0, ob1 = 0;
    if (!p.full && u0 < u1) {
        int lo = 0, hi = p.B - 1;                      // largest b with offs[b] <= f0
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (p.offs[mid] <= f0) lo = mid; else hi = mid - 1;
        }
        b = lo; ob = p.offs[b]; ob1 = p.offs[b + 1];
    }
    for (long long u = u0; u < u1;) {
        const long long f = u / MT;
        const int mt0 = (int)(u - f * MT);
        const int mt1 = (int)(u1 - u < MT - mt0 ? mt0 + (u1 - u) : MT);      // this token's tiles [mt0, mt1)
        u += mt1 - mt0;
        int i;
        if (p.full) { b = (int)(f / p.L); i = (int)(f - (long long)b * p.L); }
        else {
            while (ob1 <= f) { b++; ob = ob1; ob1 = p.offs[b + 1]; }
            i = (int)(f - ob);
        }
        const int item = b * p.L + i;
        const int len = clamp_len(p.len[b], p.L);
        const int xw = (int)clamp_tok(p.x[(long long)b * p.L + i], p.V);
        const int bidx = (i + 1 <= len) ? len - (i + 1) : i + 1;
        for (int s = lane; s < S; s += WAVE) {
            al[s] = p.A[((long long)b * (p.L + 1) + i) * SP + s];
            be[s] = p.Bk[((long long)b * (p.L + 1) + bidx) * SP + s];
        }
        // ---- U = abw . S2o on the matrix cores; br straight from the accumulators  (:201-204) ------------
        const float *ap = q.BSSp + (long long)xw * MT * KQ4 * 256 + lane * 4;
        const float *bet = be + lk;
        f32x4_t A[D1M_MAXKQ4];
#pragma unroll
        for (int g = 0; g < D1M_MAXKQ4; g++) A[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // b~ of a k-group is read from LDS one group ahead of its use (the wrap-around read serves the next
        // row tile), so the only wait in front of the MFMAs is for data requested 20 MFMAs earlier
        float bj[4];
#pragma unroll
        for (int u = 0; u < 4; u++) bj[u] = bet[u * 4];
        // pass mt0 - 1 only issues the first row tile's loads, in the same program order as every later refill
        for (int mt = mt0 - 1; mt < mt1; mt++) {
            const bool act = mt >= mt0;
            f32x4_t acc[NT];
#pragma unroll
            for (int n = 0; n < NT; n++) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const float ai = al[(act ? mt : mt0) * 16 + lr];
            const float *an = ap + (long long)(mt + 1 < mt1 ? mt + 1 : mt1 - 1) * KQ4 * 256;
#pragma unroll
            for (int g = 0; g < D1M_MAXKQ4; g++) {
                float bjn[4];
                {
                    const int gn = (g + 1 < D1M_MAXKQ4 && g + 1 < KQ4) ? g + 1 : 0;
#pragma unroll
                    for (int u = 0; u < 4; u++) bjn[u] = bet[gn * 16 + u * 4];
                }
                if (act && g < KQ4) {
                    // all-padding k-steps are skipped in the last register group (compile-time position: guards in
                    // every group cost 40 spilled registers)
                    const int nu = g == D1M_MAXKQ4 - 1 ? klast : 4;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (g < D1M_MAXKQ4 - 1 || u < nu) {
                            const float a = (ai * bj[u]) * A[g][u];                      // :202-203
#pragma unroll
                            for (int n = 0; n < NT; n++)
                                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bq[g][n][u], acc[n], 0, 0, 0);
                        }
                    }
                }
                A[g] = *(const f32x4_t *)(an + (g < KQ4 ? g : KQ4 - 1) * 256);           // next row tile
                __builtin_amdgcn_sched_barrier(0);          // keep the refills in program order (vmcnt is in-order)
#pragma unroll
                for (int u = 0; u < 4; u++) bj[u] = bjn[u];
            }
            // D: column = n*16 + lr, rows = mt*16 + lk*4 + {0..3}; this tile's share of br_q (:204)
            if (act) {
                float *bo = q.br + ((long long)item * MT + mt) * NC;
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    const f32x4_t s1 = *(const f32x4_t *)(s1l + ((mt * NT + n) * 64 + lane) * 4);
                    float v = 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; r++) v = fmaf(s1[r], acc[n][r], v);
                    v += __shfl_xor(v, 16, WAVE);
                    v += __shfl_xor(v, 32, WAVE);
                    if (lk == 0) bo[n * 16 + lr] = v;
                }
            }
        }
    }
}

inline size_t decomp1_mfma_lds_bytes(int MT, int NT) {
    return ((size_t)MT * NT * 256 + 4 * 2 * (size_t)MT * 16) * 4;
}

// score = br . Cout^T (:205), priority product, emissions / argmax, one position per wavefront at a time; the
// weights are staged in LDS once per workgroup (STAGED) or read through the caches when they do not fit.
template <bool STAGED>
__global__ void __launch_bounds__(1024)
decomp1_label_kernel(const Decomp1ScoreParams p, const float *__restrict__ brg, int NC, int MT,
                     const int64_t *__restrict__ offs_all) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, nt_ = blockDim.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RO = p.RO, K = p.K, Kc = p.Kc;
    float *brl = smem + w * (NC + 2 * Kc);        // [NC] this wavefront's br row
    float *sc = brl + NC;                         // [Kc]
    float *sc2 = sc + Kc;                         // [Kc]
    const int nwv = nt_ >> 6;                     // wavefronts per workgroup (they share the staged weights)
    float *Ct = smem + nwv * (NC + 2 * Kc);       // [RO][Kc]
    float *Pl = Ct + RO * Kc;                     // [K][Kc]
    if (STAGED) {
        // 16-byte copies, 8 in flight per lane (Kc is a multiple of 64; Ct and Pl are contiguous in LDS)
        const int n4 = RO * Kc / 4, m4 = p.P ? K * Kc / 4 : 0;
        for (int base = 0; base < n4 + m4; base += 8 * nt_) {
            f32x4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = base + u * nt_ + tid;
                const float *src = e < n4 ? p.CoutT + (long long)e * 4 : p.P + (long long)(e < n4 + m4 ? e - n4 : 0) * 4;
                v[u] = *(const f32x4_t *)(e < n4 + m4 ? src : p.CoutT);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = base + u * nt_ + tid;
                if (e < n4 + m4) *(f32x4_t *)(Ct + e * 4) = v[u];
            }
        }
        __syncthreads();
    }
    const int clamp_col = p.use_crf ? K - 3 : K - 1;
    const long long gw = (long long)gridDim.x * nwv, wid = (long long)blockIdx.x * nwv + w;
    // positions past the end of their row: fill values, dealt round-robin
    if (!p.full && (p.tags || p.scores))
        for (long long pos = wid; pos < (long long)p.B * p.L; pos += gw) {
            const int b = (int)(pos / p.L), i = (int)(pos - (long long)b * p.L);
            if (i >= clamp_len(p.len[b], p.L)) {
                if (p.tags && lane == 0) p.tags[pos] = -1;
                if (p.scores) for (int c = lane; c < K; c += WAVE) p.scores[pos * K + c] = 0.0f;
            }
        }
    // live tokens: an equal contiguous slice of the flat order per wavefront (offs_all), or every position
    const long long total = p.full ? (long long)p.B * p.L : offs_all[p.B];
    const long long f0 = total * wid / gw, f1 = total * (wid + 1) / gw;
    int b = 0;
    long long ob = 0, ob1 = 0;
    if (!p.full && f0 < f1) {
        int lo = 0, hi = p.B - 1;                      // largest b with offs[b] <= f0
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (offs_all[mid] <= f0) lo = mid; else hi = mid - 1;
        }
        b = lo; ob = offs_all[b]; ob1 = offs_all[b + 1];
    }
    for (long long f = f0; f < f1; f++) {
        int i;
        if (p.full) { b = (int)(f / p.L); i = (int)(f - (long long)b * p.L); }
        else {
            while (ob1 <= f) { b++; ob = ob1; ob1 = offs_all[b + 1]; }
            i = (int)(f - ob);
        }
        const long long pos = (long long)b * p.L + i;
        const int len = clamp_len(p.len[b], p.L);
        for (int c = lane; c < NC; c += WAVE) {         // br = sum of the row tiles' partial rows, in tile order
            float v = 0.0f;
            for (int mt = 0; mt < MT; mt++) v += brg[(pos * MT + mt) * NC + c];
            brl[c] = v;
        }
        // two columns per lane (c, c+64; Kc is a multiple of 64 so the second read stays inside the padded
        // row) and the reduction unrolled by 8: the LDS reads of a batch are in flight together
        for (int c = lane; c < K; c += 2 * WAVE) {
            const bool two = c + WAVE < Kc;
            float s0 = 0.0f, s1 = 0.0f;
            int qq = 0;
            for (; qq + 8 <= RO; qq += 8) {
                float bq[8], c0[8], c1[8];
                *(f32x4_t *)&bq[0] = *(const f32x4_t *)(brl + qq);         // broadcast reads, 16 B each
                *(f32x4_t *)&bq[4] = *(const f32x4_t *)(brl + qq + 4);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    c0[u] = STAGED ? Ct[(qq + u) * Kc + c] : p.CoutT[(long long)(qq + u) * Kc + c];
                    c1[u] = !two ? 0.0f : STAGED ? Ct[(qq + u) * Kc + c + WAVE] : p.CoutT[(long long)(qq + u) * Kc + c + WAVE];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) { s0 = fmaf(bq[u], c0[u], s0); s1 = fmaf(bq[u], c1[u], s1); }
            }
            for (; qq < RO; qq++) {
                const float bq = brl[qq];
                s0 = fmaf(bq, STAGED ? Ct[qq * Kc + c] : p.CoutT[(long long)qq * Kc + c], s0);
                if (two) s1 = fmaf(bq, STAGED ? Ct[qq * Kc + c + WAVE] : p.CoutT[(long long)qq * Kc + c + WAVE], s1);
            }
            sc[c] = s0;
            if (two) sc[c + WAVE] = s1;
        }
        const float *fin = sc;
        if (p.P) {
            for (int d = lane; d < K; d += 2 * WAVE) {
                const bool two = d + WAVE < Kc;
                float s0 = 0.0f, s1 = 0.0f;
                int c = 0;
                for (; c + 8 <= K; c += 8) {
                    float sv[8], p0[8], p1[8];
                    *(f32x4_t *)&sv[0] = *(const f32x4_t *)(sc + c);
                    *(f32x4_t *)&sv[4] = *(const f32x4_t *)(sc + c + 4);
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        p0[u] = STAGED ? Pl[(c + u) * Kc + d] : p.P[(long long)(c + u) * Kc + d];
                        p1[u] = !two ? 0.0f : STAGED ? Pl[(c + u) * Kc + d + WAVE] : p.P[(long long)(c + u) * Kc + d + WAVE];
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) { s0 = fmaf(sv[u], p0[u], s0); s1 = fmaf(sv[u], p1[u], s1); }
                }
                for (; c < K; c++) {
                    const float sv = sc[c];
                    s0 = fmaf(sv, STAGED ? Pl[c * Kc + d] : p.P[(long long)c * Kc + d], s0);
                    if (two) s1 = fmaf(sv, STAGED ? Pl[c * Kc + d + WAVE] : p.P[(long long)c * Kc + d + WAVE], s1);
                }
                sc2[d] = s0;
                if (two) sc2[d + WAVE] = s1;
            }
            fin = sc2;
        }
        if (p.scores)
            for (int c = lane; c < K; c += WAVE) p.scores[(long long)pos * K + c] = fin[c];
        if (p.use_crf) {
            for (int c = lane; c < K; c += WAVE) {
                float vv = fin[c] + 0.0f;
                if (c == clamp_col) vv = fminf(vv, p.threshold);
                p.crf_scores[(long long)pos * p.Kp + c] = vv;
            }
        } else {
            float bv = -INFINITY; int bi = 0x7ffffffe;
            for (int c = lane; c < K; c += WAVE) {
                float vv = fin[c] + 0.0f;
                if (c == clamp_col) vv = fminf(vv, p.threshold);
                if (vv > bv) { bv = vv; bi = c; }
            }
            bi = wave_argmax_dpp(bv, bi);
            if (lane == 0) {
                if (bi >= K) bi = 0;
                const int tag = (bi == K - 1) ? p.o_idx : bi;
                if (p.tags) p.tags[pos] = tag;
                if (p.flat && i < len) p.flat[p.offs[b] + i] = tag;
            }
        }
    }
}

inline size_t decomp1_label_lds_bytes(int NC, int RO, int K, int Kc, bool staged, int nwaves) {
    return ((size_t)nwaves * (NC + 2 * (size_t)Kc) + (staged ? ((size_t)RO + K) * Kc : 0)) * 4;
}

inline size_t decomp1_score_lds_bytes(int S, int SP, int Rp, int ROp, int Kc) {
    return ((size_t)2 * SP + Rp + ROp + 2 * (size_t)Kc + (size_t)S * SP) * sizeof(float);
}

// Osum[from][to] = sum_q csum[q] S1o[from][q] S2o[to][q] (+ Wo), csum[q] = sum_k Cout[k][q]
// (get_output_tensor_sum, model_decompose_independent.py:210-217); one thread per (from,to).
__global__ void output_sum_kernel(const float *Cout, const float *S1o, const float *S2o, const float *Wo,
                                  float *Osum, int K, int S, int SP, int RO) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * S) return;
    int fr = idx / S, to = idx - fr * S;
    float acc = 0.0f;
    for (int q = 0; q < RO; q++) {
        float cs = 0.0f;
        for (int k = 0; k < K; k++) cs += Cout[(long long)k * RO + q];
        acc = fmaf(S2o[(long long)to * RO + q], cs * S1o[(long long)fr * RO + q], acc);
    }
    if (Wo) acc += Wo[(long long)fr * S + to];
    Osum[(long long)fr * SP + to] = acc;
}

// ---- decomposed independent=0 (FARNN_S_D_W.get_final_score, model_decompose.py:309-323) -------------
//     score[c] = sum_r v_r C[c,r] (a.S1)_r (b~.S2)_r  +  sum_q (a.S1w)_q (b~.S2w)_q Cw[c,q]
// with a = state BEFORE token i (:418).  One workgroup per token.
struct Decomp0ScoreParams {
    const float *A, *Bk;            // stash [B][L+1][SP]
    const float *Vgen;              // [V][Rp]   generalized word table (NOT scaled by sum_c C)
    const float *S1, *S2;           // [S][Rp]
    const float *CT;                // [R][Kc]   C_embed^T
    const float *S1w, *S2w;         // [S][RWp]
    const float *CwT;               // [RW][Kc]  C_wildcard^T
    const float *P;                 // [K][Kc] or nullptr
    const int64_t *x, *len, *offs;
    int32_t *tags; int64_t *flat; float *scores; float *crf_scores;
    int B, L, S, SP, R, Rp, RW, RWp, K, Kp, Kc, V;
    int full, use_crf, o_idx;
    float threshold;
};

// D0_TOK tokens per workgroup: a thread owns one factor column (then one label column) for four of them, so every factor
// entry read from L2 feeds four FMAs (r02a: one workgroup per token re-read all four S x R factors -- 83 KB at the
// config-2 size -- for every token: 730 MB of L2 traffic per launch, 88 us).
constexpr int D0_TOK = 8;

__global__ void __launch_bounds__(256)
decomp0_score_kernel(const Decomp0ScoreParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, nt = blockDim.x;
    const int i0 = blockIdx.x * D0_TOK, b = blockIdx.y;
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, R = p.R, Rp = p.Rp, RW = p.RW, RWp = p.RWp, K = p.K, Kc = p.Kc;
    const int ntok = min(D0_TOK, p.L - i0);              // positions of this workgroup that exist
    const int nlive = max(0, min(D0_TOK, nsteps - i0));  // ... that were computed
    for (int t = nlive + w; t < ntok; t += 4) {          // pads (LOCAL mode)
        const int i = i0 + t;
        if (p.tags && lane == 0) p.tags[(long long)b * p.L + i] = -1;
        if (p.scores) for (int c = lane; c < K; c += WAVE) p.scores[((long long)b * p.L + i) * K + c] = 0.0f;
    }
    if (nlive <= 0) return;
    float *alpha = smem;                                 // [D0_TOK][SP]
    float *beta = alpha + D0_TOK * SP;                   // [D0_TOK][SP]
    float *ab = beta + D0_TOK * SP;                      // [D0_TOK][Rp]
    float *abw = ab + D0_TOK * Rp;                       // [D0_TOK][RWp]
    float *sc = abw + D0_TOK * RWp;                      // [D0_TOK][Kc]
    float *sc2 = sc + D0_TOK * Kc;                       // [D0_TOK][Kc]
    __shared__ int tokid[D0_TOK];

    for (int idx = tid; idx < D0_TOK * SP; idx += nt) {
        const int t = idx / SP, sidx = idx - t * SP;
        float av = 0.0f, bv = 0.0f;
        if (t < nlive) {
            const int i = i0 + t;
            const int bidx = (i + 1 <= len) ? len - (i + 1) : i + 1;
            av = p.A[((long long)b * (p.L + 1) + i) * SP + sidx];                       // state BEFORE token i (:418)
            bv = p.Bk[((long long)b * (p.L + 1) + bidx) * SP + sidx];
        }
        alpha[idx] = av; beta[idx] = bv;
    }
    if (tid < D0_TOK) tokid[tid] = tid < nlive ? clamp_tok(p.x[(long long)b * p.L + i0 + tid], p.V) : 0;
    for (int idx = tid; idx < D0_TOK * (Rp + RWp); idx += nt) ab[idx] = 0.0f;          // (the pad columns are read, times zero)
    __syncthreads();
    // ---- (a.S1)_r (b~.S2)_r per factor column, four tokens per thread ------------------------------------------
    const int th = tid >> 7, cl = tid & 127;             // token half, column lane
    for (int r = cl; r < R + RW; r += 128) {
        const bool lang = r < R;
        const int col = lang ? r : r - R, ld = lang ? Rp : RWp;
        const float *f1 = (lang ? p.S1 : p.S1w) + col, *f2 = (lang ? p.S2 : p.S2w) + col;
        const float *al = alpha + th * 4 * SP, *be = beta + th * 4 * SP;
        float a[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
        int sidx = 0;
        for (; sidx + 4 <= S; sidx += 4) {               // eight factor loads in flight; the states come as 16-byte LDS reads
            float u1[4], u2[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { u1[u] = f1[(long long)(sidx + u) * ld]; u2[u] = f2[(long long)(sidx + u) * ld]; }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float4 a4 = ld4(al + t * SP + sidx), b4 = ld4(be + t * SP + sidx);
                a[t] = fmaf(a4.x, u1[0], a[t]); a[t] = fmaf(a4.y, u1[1], a[t]);                 // :314 / :318
                a[t] = fmaf(a4.z, u1[2], a[t]); a[t] = fmaf(a4.w, u1[3], a[t]);
                bb[t] = fmaf(b4.x, u2[0], bb[t]); bb[t] = fmaf(b4.y, u2[1], bb[t]);             // :315 / :319
                bb[t] = fmaf(b4.z, u2[2], bb[t]); bb[t] = fmaf(b4.w, u2[3], bb[t]);
            }
        }
        for (; sidx < S; sidx++) {
            const float u1 = f1[(long long)sidx * ld], u2 = f2[(long long)sidx * ld];
#pragma unroll
            for (int t = 0; t < 4; t++) { a[t] = fmaf(al[t * SP + sidx], u1, a[t]); bb[t] = fmaf(be[t * SP + sidx], u2, bb[t]); }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int tk = th * 4 + t;
            if (lang) ab[tk * Rp + col] = p.Vgen[(long long)tokid[tk] * Rp + col] * (a[t] * bb[t]);   // :313,:316
            else abw[tk * RWp + col] = a[t] * bb[t];                                                  // :320
        }
    }
    __syncthreads();
    // ---- label scores, four tokens per thread ---------------------------------------------------------------------
    for (int c = cl; c < K; c += 128) {
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, w0[4] = {0.f, 0.f, 0.f, 0.f};
        const float *abp = ab + th * 4 * Rp, *abq = abw + th * 4 * RWp;
        for (int r = 0; r < R; r += 4) {                     // (the rows past R / RW are zero in CT / CwT's padded images? no: guarded)
            float cv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) cv[u] = r + u < R ? p.CT[(long long)(r + u) * Kc + c] : 0.0f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float4 x4 = ld4(abp + t * Rp + r);                                      // Rp % 4 == 0; pads of ab are finite
                s0[t] = fmaf(x4.x, cv[0], s0[t]); s0[t] = fmaf(x4.y, cv[1], s0[t]);           // :317
                s0[t] = fmaf(x4.z, cv[2], s0[t]); s0[t] = fmaf(x4.w, cv[3], s0[t]);
            }
        }
        for (int q = 0; q < RW; q += 4) {
            float cv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) cv[u] = q + u < RW ? p.CwT[(long long)(q + u) * Kc + c] : 0.0f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float4 x4 = ld4(abq + t * RWp + q);
                w0[t] = fmaf(x4.x, cv[0], w0[t]); w0[t] = fmaf(x4.y, cv[1], w0[t]);           // :321
                w0[t] = fmaf(x4.z, cv[2], w0[t]); w0[t] = fmaf(x4.w, cv[3], w0[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) sc[(th * 4 + t) * Kc + c] = s0[t] + w0[t];                        // :322
    }
    __syncthreads();
    // ---- priority, outputs, decode: a wavefront per token ----------------------------------------------------------
    const int clamp_col = p.use_crf ? K - 3 : K - 1;
    for (int t = w; t < nlive; t += 4) {
        const int i = i0 + t;
        const float *fin = sc + t * Kc;
        if (p.P) {
            for (int d = lane; d < K; d += WAVE) {
                float acc = 0.0f;
                for (int c = 0; c < K; c++) acc = fmaf(fin[c], p.P[(long long)c * Kc + d], acc);
                sc2[t * Kc + d] = acc;
            }
            __builtin_amdgcn_wave_barrier();
            fin = sc2 + t * Kc;
        }
        if (p.scores)
            for (int c = lane; c < K; c += WAVE) p.scores[((long long)b * p.L + i) * K + c] = fin[c];
        if (p.use_crf) {
            for (int c = lane; c < K; c += WAVE) {
                float vv = fin[c] + 0.0f;
                if (c == clamp_col) vv = fminf(vv, p.threshold);
                p.crf_scores[((long long)b * p.L + i) * p.Kp + c] = vv;
            }
        } else {
            float bv = -INFINITY; int bi = 0x7ffffffe;
            for (int c = lane; c < K; c += WAVE) {
                float vv = fin[c] + 0.0f;
                if (c == clamp_col) vv = fminf(vv, p.threshold);
                if (vv > bv) { bv = vv; bi = c; }
            }
            bi = wave_argmax_dpp(bv, bi);
            if (lane == 0) {
                if (bi >= K) bi = 0;
                const int tag = (bi == K - 1) ? p.o_idx : bi;
                if (p.tags) p.tags[(long long)b * p.L + i] = tag;
                if (p.flat && i < len) p.flat[p.offs[b] + i] = tag;
            }
        }
    }
}

inline size_t decomp0_score_lds_bytes(int SP, int Rp, int RWp, int Kc) {
    return (size_t)D0_TOK * ((size_t)2 * SP + Rp + RWp + 2 * (size_t)Kc) * sizeof(float);
}

// table[v][r] *= sum_c C[c][r]   (_R = V_vec * C_vec_sum, model_decompose.py:253,:393)
__global__ void scale_by_colsum_kernel(float *table, const float *C, int V, int R, int K) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)V * R) return;
    const int r = (int)(idx % R);
    float cs = 0.0f;
    for (int k = 0; k < K; k++) cs += C[(long long)k * R + r];
    table[idx] *= cs;
}

}  // namespace farnn
