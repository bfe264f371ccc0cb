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
// bss[x] = sum_r S1 S2 v_x + W depends on the word only: BSS[V][S][SP] is built once at create time (0.48 GB
// at V=11 k, S=104).  Per token the work is then a real GEMM against weights shared by every token,
//     U[S x RO] = abw[S x S] . S2o[S x RO],   abw_ij = (a_i b~_j) bss[x]_ij,      br_q = sum_i S1o_iq U_iq
// 1.5 MFLOP on v_mfma_f32_16x16x4_f32 (exact f32 fma chains): one workgroup of 4 wavefronts per token, a
// wavefront owns whole 16-row tiles of U across all 16-column tiles (the A fragment is read once per k-step
// and used for every column tile), abw and S2o live in LDS.  16.5 ms -> see DESIGN.md for the measured time.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int D1M_MAXNT = 8;            // 16-column tiles of U: RO <= 128

struct Decomp1MfmaParams {
    Decomp1ScoreParams base;
    const float *BSS;                   // [V][S][SP]
    int ldA, ldB, MT, NT, KQ;           // LDS strides; row tiles, column tiles, k-steps of 4
};

template <int NT>
__global__ void __launch_bounds__(256)
decomp1_score_mfma_kernel(const Decomp1MfmaParams q) {
    const Decomp1ScoreParams &p = q.base;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, nt_ = blockDim.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = blockIdx.x, b = blockIdx.y;
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, RO = p.RO, K = p.K;
    if (i >= nsteps) {
        if (p.tags && tid == 0) p.tags[(long long)b * p.L + i] = -1;
        if (p.scores) for (int c = tid; c < K; c += nt_) p.scores[((long long)b * p.L + i) * K + c] = 0.0f;
        return;
    }
    const int ldA = q.ldA, ldB = q.ldB, MT = q.MT, KQ = q.KQ;
    float *alpha = smem;                 // [SP]
    float *beta = alpha + SP;            // [SP]
    float *br = beta + SP;               // [NT*16]
    float *brp = br + NT * 16;           // [4][NT*16] per-wavefront partial br
    float *sc = brp + 4 * (NT * 16 > p.Kc ? NT * 16 : p.Kc);     // [Kc]   (brp doubles as the 4 x Kc score partials)
    float *sc2 = sc + p.Kc;              // [Kc]
    float *abw = sc2 + p.Kc;             // [MT*16][ldA]
    float *s2o = abw + MT * 16 * ldA;    // [KQ*4][ldB]

    const float *ar = p.A + ((long long)b * (p.L + 1) + i) * SP;
    const int bidx = (i + 1 <= len) ? len - (i + 1) : i + 1;
    const float *brow = p.Bk + ((long long)b * (p.L + 1) + bidx) * SP;
    const float *bss = q.BSS + (long long)clamp_tok(p.x[(long long)b * p.L + i], p.V) * S * SP;
    for (int s = tid; s < SP; s += nt_) { alpha[s] = ar[s]; beta[s] = brow[s]; }
    // Everything below is staged with 16-byte loads issued in batches (8 in flight per lane): the element-
    // wise version of these loops serialised ~90 dependent L2 round trips per lane (2 ms per batch).
    for (int idx = tid; idx < (MT * 16 * ldA + KQ * 4 * ldB); idx += nt_) abw[idx] = 0.0f;   // pads (abw | s2o contiguous)
    __syncthreads();
    constexpr int UB = 8;
    const int ro4 = p.ROp >> 2, sp4 = SP >> 2;
    for (int base = 0; base < S * ro4; base += UB * nt_) {             // S2o -> LDS
        float4 v[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int e = base + u * nt_ + tid;
            v[u] = e < S * ro4 ? ld4(p.S2o + (long long)e * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int e = base + u * nt_ + tid;
            if (e < S * ro4) {
                const int k = e / ro4, c = (e - k * ro4) * 4;
                float *d = s2o + k * ldB + c;
                if (c + 0 < RO) d[0] = v[u].x;
                if (c + 1 < RO) d[1] = v[u].y;
                if (c + 2 < RO) d[2] = v[u].z;
                if (c + 3 < RO) d[3] = v[u].w;
            }
        }
    }
    for (int base = 0; base < S * sp4; base += UB * nt_) {             // abw = (a b~^T) . bss[x]   (:202-203)
        float4 v[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int e = base + u * nt_ + tid;
            v[u] = e < S * sp4 ? ld4(bss + (long long)e * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int e = base + u * nt_ + tid;
            if (e < S * sp4) {
                const int r = e / sp4, c = (e - r * sp4) * 4;
                const float a = alpha[r];
                float *d = abw + r * ldA + c;              // ldA is odd: scalar stores
                if (c + 0 < S) d[0] = (a * beta[c + 0]) * v[u].x;
                if (c + 1 < S) d[1] = (a * beta[c + 1]) * v[u].y;
                if (c + 2 < S) d[2] = (a * beta[c + 2]) * v[u].z;
                if (c + 3 < S) d[3] = (a * beta[c + 3]) * v[u].w;
            }
        }
    }
    __syncthreads();
    // ---- U = abw . S2o on the matrix cores; br partials straight from the accumulators  (:204) --------
    const int lr = lane & 15, lk = lane >> 4;
    float brl[NT];
#pragma unroll
    for (int n = 0; n < NT; n++) brl[n] = 0.0f;
    for (int mt = w; mt < MT; mt += 4) {
        f32x4_t acc[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const float *ap = abw + (mt * 16 + lr) * ldA + lk;
        const float *bp = s2o + lk * ldB + lr;
        // the operands of a k-step are read before its MFMAs and the next step's reads are issued before
        // this step's MFMAs retire (NT is a compile-time constant: no per-tile branches in the loop)
        float a = ap[0], bv[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) bv[n] = bp[n * 16];
        for (int kq = 0; kq < KQ; kq++) {
            const int kn = kq + 1 < KQ ? kq + 1 : kq;
            const float an = ap[kn * 4];
            float bn[NT];
#pragma unroll
            for (int n = 0; n < NT; n++) bn[n] = bp[kn * 4 * ldB + n * 16];
#pragma unroll
            for (int n = 0; n < NT; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[n], acc[n], 0, 0, 0);
            a = an;
#pragma unroll
            for (int n = 0; n < NT; n++) bv[n] = bn[n];
        }
        // D: column = n*16 + lr, rows = mt*16 + lk*4 + {0..3}
        float s1v[NT][4];
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = mt * 16 + lk * 4 + r, col = n * 16 + lr;
                s1v[n][r] = (row < S && col < RO) ? p.S1o[(long long)row * p.ROp + col] : 0.0f;
            }
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) brl[n] = fmaf(s1v[n][r], acc[n][r], brl[n]);
    }
#pragma unroll
    for (int n = 0; n < NT; n++) {
        float v = brl[n];
        v += __shfl_xor(v, 16, WAVE);
        v += __shfl_xor(v, 32, WAVE);
        if (lk == 0) brp[w * NT * 16 + n * 16 + lr] = v;
    }
    __syncthreads();
    for (int c = tid; c < NT * 16; c += nt_)
        br[c] = (brp[c] + brp[NT * 16 + c]) + (brp[2 * NT * 16 + c] + brp[3 * NT * 16 + c]);
    __syncthreads();
    // score = br . Cout^T (:205): the output ranks are split over the workgroup's quarter-blocks so that the
    // 70-odd loads of a column are three batches, not nine; partial sums meet in LDS (fixed order)
    {
        const int parts = nt_ / 64 >= 4 ? 4 : 1;                   // K <= 256 columns, 64-lane quarter per part
        const int part = tid >> 6, per = (RO + parts - 1) / parts;
        for (int c = lane; c < K; c += WAVE) {
            float s = 0.0f;
            const int q0 = part * per, q1 = q0 + per < RO ? q0 + per : RO;
            for (int qb = q0; qb < q1; qb += UB) {
                float cv[UB];
#pragma unroll
                for (int u = 0; u < UB; u++) cv[u] = qb + u < q1 ? p.CoutT[(long long)(qb + u) * p.Kc + c] : 0.0f;
#pragma unroll
                for (int u = 0; u < UB; u++) if (qb + u < q1) s = fmaf(br[qb + u], cv[u], s);
            }
            brp[part * p.Kc + c] = s;                               // brp (4 x NT*16 >= ?) -- see lds sizing
        }
        __syncthreads();
        for (int c = tid; c < K; c += nt_) {
            float s = brp[c];
            for (int pp = 1; pp < parts; pp++) s += brp[pp * p.Kc + c];
            sc[c] = s;
        }
    }
    __syncthreads();
    const float *fin = sc;
    if (p.P) {
        for (int d = tid; d < K; d += nt_) {
            float s = 0.0f;
            for (int c0 = 0; c0 < K; c0 += UB) {
                float pv[UB];
#pragma unroll
                for (int u = 0; u < UB; u++) pv[u] = c0 + u < K ? p.P[(long long)(c0 + u) * p.Kc + d] : 0.0f;
#pragma unroll
                for (int u = 0; u < UB; u++) if (c0 + u < K) s = fmaf(sc[c0 + u], pv[u], s);
            }
            sc2[d] = s;
        }
        __syncthreads();
        fin = sc2;
    }
    if (p.scores)
        for (int c = tid; c < K; c += nt_) p.scores[((long long)b * p.L + i) * K + c] = fin[c];
    const int clamp_col = p.use_crf ? K - 3 : K - 1;
    if (p.use_crf) {
        for (int c = tid; c < K; c += nt_) {
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

inline size_t decomp1_mfma_lds_bytes(int SP, int Kc, int ldA, int ldB, int MT, int NT, int KQ) {
    const size_t brp = 4 * (size_t)(NT * 16 > Kc ? NT * 16 : Kc);
    return ((size_t)2 * SP + (size_t)NT * 16 + brp + 2 * (size_t)Kc + (size_t)MT * 16 * ldA + (size_t)KQ * 4 * ldB) * 4;
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
