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

__global__ void __launch_bounds__(256)
decomp0_score_kernel(const Decomp0ScoreParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, nt = blockDim.x;
    const int i = blockIdx.x, b = blockIdx.y;
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, R = p.R, Rp = p.Rp, RW = p.RW, RWp = p.RWp, K = p.K;
    if (i >= nsteps) {
        if (p.tags && tid == 0) p.tags[(long long)b * p.L + i] = -1;
        if (p.scores) for (int c = tid; c < K; c += nt) p.scores[((long long)b * p.L + i) * K + c] = 0.0f;
        return;
    }
    float *alpha = smem;                 // [SP]
    float *beta = alpha + SP;            // [SP]
    float *ab = beta + SP;               // [Rp]
    float *abw = ab + Rp;                // [RWp]
    float *sc = abw + RWp;               // [Kc]
    float *sc2 = sc + p.Kc;              // [Kc]

    const float *ar = p.A + ((long long)b * (p.L + 1) + i) * SP;
    const int bidx = (i + 1 <= len) ? len - (i + 1) : i + 1;
    const float *brow = p.Bk + ((long long)b * (p.L + 1) + bidx) * SP;
    const float *vg = p.Vgen + (long long)clamp_tok(p.x[(long long)b * p.L + i], p.V) * Rp;
    for (int s = tid; s < SP; s += nt) { alpha[s] = ar[s]; beta[s] = brow[s]; }
    __syncthreads();
    for (int r = tid; r < R + RW; r += nt) {
        const bool lang = r < R;
        const int col = lang ? r : r - R, ld = lang ? Rp : RWp;
        const float *f1 = (lang ? p.S1 : p.S1w) + col, *f2 = (lang ? p.S2 : p.S2w) + col;
        float a = 0.0f, bb = 0.0f;
        for (int s = 0; s < S; s++) {
            a = fmaf(alpha[s], f1[(long long)s * ld], a);                             // :314 / :318
            bb = fmaf(beta[s], f2[(long long)s * ld], bb);                            // :315 / :319
        }
        if (lang) ab[col] = vg[col] * (a * bb);                                       // :313,:316
        else abw[col] = a * bb;                                                       // :320
    }
    __syncthreads();
    for (int c = tid; c < K; c += nt) {
        float s = 0.0f, sw = 0.0f;
        for (int r = 0; r < R; r++) s = fmaf(ab[r], p.CT[(long long)r * p.Kc + c], s);       // :317
        for (int q = 0; q < RW; q++) sw = fmaf(abw[q], p.CwT[(long long)q * p.Kc + c], sw);  // :321
        sc[c] = s + sw;                                                                      // :322
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

inline size_t decomp0_score_lds_bytes(int SP, int Rp, int RWp, int Kc) {
    return ((size_t)2 * SP + Rp + RWp + 2 * (size_t)Kc) * sizeof(float);
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
