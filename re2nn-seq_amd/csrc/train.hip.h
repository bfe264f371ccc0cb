// Training step of the decomposed i-FST tagger: loss and the gradient of every tensor the
// recurrence and the scoring read, by back-propagation through time over the stashed states.
//
// Reference: FARNN_S_D_W_I_S.forward_local(train=True) (model_decompose_single.py:207-304, farnn = 0, sum
// semiring, CE1 loss = nn.CrossEntropyLoss over the flattened valid positions, model_decompose.py:79-91)
// followed by loss.backward() (train_decompose.py:190).  With Osum = C.sum(0) (:232):
//   forward chain   f_t = nl(((f_{t-1} S1 * v_t) S2^T + f_{t-1} W) * Osum)              (:169-178,:181)
//   backward chain  b_t = nl(((bb S2 * v) S1^T + bb W^T)),  bb = b_{t-1} * Osum        (:157-158,:174-178)
//   score_i = (f_{i+1} * b_{len-1-i}) C^T [. P]                                         (:200-203,:266-272)
// What is computed here is exactly what autograd computes for those expressions; the table v = Vgen[x]
// (model_decompose.py:222-241) is differentiated by the caller from dVgen.
//
// First version: one workgroup per (sequence, direction), weights read through L2, every matrix-vector product
// of a step split over the workgroup's four wavefronts.  Parameter gradients that are sums of outer products
// over tokens (dS1, dS2, dW, dC) are NOT accumulated with per-token atomics: the chain kernels store the
// per-token adjoint rows and atb_accumulate_kernel reduces them as tall-skinny A^T B products.
#pragma once
#include "common.hip.h"

namespace farnn {

struct TrainParams {
    // weights (device, row-major, unpadded) and their transposes (workspace)
    const float *Vgen, *S1, *S2, *W, *C, *h0, *hT, *P;
    const float *S1T, *S2T, *WT, *Osum;
    const int64_t *x, *len, *labels;
    // stashes and per-token adjoint rows, all [B][L+1][.] and zero outside the valid rows
    float *A, *Bk;            // states: A[b][t] after t tokens, Bk[b][t] after t reversed tokens
    float *GA, *GB;           // dL/dA, dL/dBk from the scoring
    float *Zf, *Zb;           // [.][S] pre-activation adjoints
    float *D1f, *D1b, *Tf, *Tb;   // [.][R]  u*v and v*rr
    float *BBAR;              // [.][S]  b_{t-1} * Osum
    float *DS, *AB;           // [B][L][K] d loss / d (pre-priority) scores; [B][L][S] alpha*beta
    float *dVgen, *dOsum, *dh0, *dhT, *loss;
    int32_t *tags;
    int B, L, V, S, R, K, nl, o_idx;
    float threshold, inv_tokens;
};

__device__ __forceinline__ float nl_grad_from_output(float y, int nl) {
    switch (nl) {
        case FARNN_NL_RELU:     return y > 0.0f ? 1.0f : 0.0f;
        case FARNN_NL_TANH:     return 1.0f - y * y;
        case FARNN_NL_RELUTANH: return y > 0.0f ? 1.0f - y * y : 0.0f;
        default:                return 1.0f;
    }
}

// out[j] (+)= sum_k in[k] M[k][j], M row-major [K][J] in global memory, `in` in LDS.  The k range is split over
// the workgroup's wavefronts; partial sums land in part[wave][J] (LDS) and are added by the caller after a barrier.
__device__ __forceinline__ void matvec_partial(float *part, const float *in, const float *__restrict__ M, int K, int J,
                                               int tid, int nthreads) {
    const int nw = nthreads >> 6, w = tid >> 6, lane = tid & 63;
    const int k0 = (K * w) / nw, k1 = (K * (w + 1)) / nw;
    for (int j = lane; j < J; j += WAVE) {
        float a0 = 0.0f, a1 = 0.0f;
        int k = k0;
        for (; k + 1 < k1; k += 2) {
            a0 = fmaf(in[k], M[(long long)k * J + j], a0);
            a1 = fmaf(in[k + 1], M[(long long)(k + 1) * J + j], a1);
        }
        if (k < k1) a0 = fmaf(in[k], M[(long long)k * J + j], a0);
        part[w * J + j] = a0 + a1;
    }
}
__device__ __forceinline__ float part_sum(const float *part, int J, int j, int nw) {
    float s = part[j];
    for (int w = 1; w < nw; w++) s += part[w * J + j];
    return s;
}

__global__ void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int r = idx / cols, c = idx - r * cols;
    out[(long long)c * rows + r] = in[idx];
}

__global__ void column_sum_kernel(const float *__restrict__ C, float *__restrict__ Osum, int K, int S) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    float a = 0.0f;
    for (int c = 0; c < K; c++) a += C[(long long)c * S + s];          // C_output_mat.sum(0)  (:232)
    Osum[s] = a;
}

// ---- forward chains with the stash ------------------------------------------------------------------------
// grid (B, 2): blockIdx.y = 0 forward, 1 backward.  LDS: f[S], t[R], part[4][max(S,R)], part2[4][S]
__global__ void __launch_bounds__(256)
train_forward_kernel(const TrainParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nt = blockDim.x, nw = nt >> 6;
    const int b = blockIdx.x, dir = blockIdx.y;
    const int S = p.S, R = p.R, SR = S > R ? S : R;
    float *f = smem, *tv = f + S, *part = tv + R, *part2 = part + nw * SR;
    const int len = clamp_len(p.len[b], p.L);
    float *stash = (dir == 0 ? p.A : p.Bk) + (long long)b * (p.L + 1) * S;
    for (int s = tid; s < S; s += nt) {
        const float h = dir == 0 ? p.h0[s] : p.hT[s];
        stash[s] = h;
        f[s] = dir == 0 ? h : h * p.Osum[s];                  // the backward chain masks its INPUT (:157-158)
    }
    __syncthreads();
    for (int t = 1; t <= len; t++) {
        const int tok = clamp_tok(p.x[(long long)b * p.L + (dir == 0 ? t - 1 : len - t)], p.V);
        const float *v = p.Vgen + (long long)tok * R;
        // rr = f . (S1 | S2) and the wildcard part f . (W | W^T): both depend on f only
        matvec_partial(part, f, dir == 0 ? p.S1 : p.S2, S, R, tid, nt);
        matvec_partial(part2, f, dir == 0 ? p.W : p.WT, S, S, tid, nt);
        __syncthreads();
        for (int r = tid; r < R; r += nt) tv[r] = v[r] * part_sum(part, R, r, nw);          // temp = V_vec * _RR
        __syncthreads();
        matvec_partial(part, tv, dir == 0 ? p.S2T : p.S1T, R, S, tid, nt);                   // temp . (S2^T | S1^T)
        __syncthreads();
        for (int s = tid; s < S; s += nt) {
            float pre = part_sum(part, S, s, nw) + part_sum(part2, S, s, nw);
            if (dir == 0) pre *= p.Osum[s];                                                  // (:181)
            const float h = apply_nl(pre, p.nl);
            stash[(long long)t * S + s] = h;
            f[s] = dir == 0 ? h : h * p.Osum[s];
        }
        __syncthreads();
    }
}

// ---- scores, cross-entropy, and the adjoints of alpha / beta ------------------------------------------------
// one wavefront per valid position.  LDS per wavefront: ab[S], sc[K], ds[K]
__global__ void __launch_bounds__(256)
train_loss_kernel(const TrainParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int S = p.S, K = p.K;
    float *ab = smem + w * (S + 2 * K), *sc = ab + S, *ds = sc + K;
    const long long pos = (long long)blockIdx.x * nw + w;
    if (pos >= (long long)p.B * p.L) return;
    const int b = (int)(pos / p.L), i = (int)(pos - (long long)b * p.L);
    const int len = clamp_len(p.len[b], p.L);
    if (i >= len) {
        if (lane == 0) p.tags[pos] = -1;
        return;
    }
    const float *al = p.A + ((long long)b * (p.L + 1) + i + 1) * S;                  // h0_forward_score[:, i+1]
    const float *be = p.Bk + ((long long)b * (p.L + 1) + (len - 1 - i)) * S;         // reverse(.., lengths+1)[:, i+1]
    for (int s = lane; s < S; s += WAVE) {
        const float v = al[s] * be[s];
        ab[s] = v;
        p.AB[pos * S + s] = v;
    }
    for (int c = lane; c < K; c += WAVE) {                                            // get_final_score (:200-203)
        const float *cr = p.C + (long long)c * S;
        float a = 0.0f;
        for (int s = 0; s < S; s++) a = fmaf(ab[s], cr[s], a);
        sc[c] = a;
    }
    if (p.P) {                                                                         // priority layer
        for (int d = lane; d < K; d += WAVE) {
            float a = 0.0f;
            for (int c = 0; c < K; c++) a = fmaf(sc[c], p.P[(long long)c * K + d], a);
            ds[d] = a;
        }
        for (int d = lane; d < K; d += WAVE) sc[d] = ds[d];
    }
    // softmax cross-entropy (mean over the batch's valid tokens) and the prediction (decode, argmax branch)
    float mx = -INFINITY;
    for (int c = lane; c < K; c += WAVE) mx = fmaxf(mx, sc[c]);
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, WAVE));
    float se = 0.0f;
    for (int c = lane; c < K; c += WAVE) se += expf(sc[c] - mx);
    for (int o = 32; o; o >>= 1) se += __shfl_xor(se, o, WAVE);
    const int lab = (int)p.labels[pos];
    const float lse = mx + logf(se);
    if (lane == 0) atomicAdd(p.loss, (lse - sc[lab < 0 || lab >= K ? 0 : lab]) * p.inv_tokens);
    {
        float bv = -INFINITY; int bi = 0x7ffffffe;
        for (int c = lane; c < K; c += WAVE) {
            float vv = sc[c] + 0.0f;
            if (c == K - 1) vv = fminf(vv, p.threshold);
            if (vv > bv) { bv = vv; bi = c; }
        }
        bi = wave_argmax_dpp(bv, bi);
        if (lane == 0) p.tags[pos] = (bi >= K) ? 0 : (bi == K - 1 ? p.o_idx : bi);
    }
    for (int c = lane; c < K; c += WAVE) ds[c] = (expf(sc[c] - lse) - (c == lab ? 1.0f : 0.0f)) * p.inv_tokens;
    if (p.P) {                                                                         // back through scores . P
        for (int c = lane; c < K; c += WAVE) {
            float a = 0.0f;
            for (int d = 0; d < K; d++) a = fmaf(ds[d], p.P[(long long)c * K + d], a);
            sc[c] = a;
        }
        for (int c = lane; c < K; c += WAVE) ds[c] = sc[c];
    }
    for (int c = lane; c < K; c += WAVE) p.DS[pos * K + c] = ds[c];
    float *ga = p.GA + ((long long)b * (p.L + 1) + i + 1) * S;
    float *gb = p.GB + ((long long)b * (p.L + 1) + (len - 1 - i)) * S;
    for (int s = lane; s < S; s += WAVE) {
        float d = 0.0f;
        for (int c = 0; c < K; c++) d = fmaf(ds[c], p.C[(long long)c * S + s], d);     // d(alpha*beta)
        ga[s] = d * be[s];
        gb[s] = d * al[s];
    }
}

// ---- back-propagation through time ----------------------------------------------------------------------------
// grid (B, 2).  LDS: g[S], z[S], y[S], fp[S] (f_{t-1} or bbar), u[R], rr[R], d1[R], tmpv[R], parts 4 x [4][max(S,R)], dO[S]
__global__ void __launch_bounds__(256)
train_backward_kernel(const TrainParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nt = blockDim.x, nw = nt >> 6;
    const int b = blockIdx.x, dir = blockIdx.y;
    const int S = p.S, R = p.R, SR = S > R ? S : R;
    float *g = smem, *z = g + S, *y = z + S, *fp = y + S, *dO = fp + S;
    float *u = dO + S, *rr = u + R, *d1 = rr + R, *tmpv = d1 + R;
    float *pa = tmpv + R, *pb = pa + nw * SR, *pc = pb + nw * SR, *pd = pc + nw * SR;
    const int len = clamp_len(p.len[b], p.L);
    const long long row0 = (long long)b * (p.L + 1);
    const float *stash = (dir == 0 ? p.A : p.Bk) + row0 * S;
    const float *G = (dir == 0 ? p.GA : p.GB) + row0 * S;
    float *Zo = (dir == 0 ? p.Zf : p.Zb) + row0 * S;
    float *D1o = (dir == 0 ? p.D1f : p.D1b) + row0 * R;
    float *To = (dir == 0 ? p.Tf : p.Tb) + row0 * R;
    for (int s = tid; s < S; s += nt) { g[s] = 0.0f; dO[s] = 0.0f; }
    __syncthreads();
    for (int t = len; t >= 1; t--) {
        const int tok = clamp_tok(p.x[(long long)b * p.L + (dir == 0 ? t - 1 : len - t)], p.V);
        const float *v = p.Vgen + (long long)tok * R;
        for (int s = tid; s < S; s += nt) {
            const float gt = g[s] + G[(long long)t * S + s];
            const float h = stash[(long long)t * S + s];
            const float yy = gt * nl_grad_from_output(h, p.nl);
            const float hp = stash[(long long)(t - 1) * S + s];
            if (dir == 0) { y[s] = yy; z[s] = yy * p.Osum[s]; fp[s] = hp; }       // mask on the OUTPUT of the step
            else          { z[s] = yy; fp[s] = hp * p.Osum[s]; y[s] = hp; }       // mask on the INPUT: fp = bbar, y keeps b_{t-1}
            Zo[(long long)t * S + s] = dir == 0 ? yy * p.Osum[s] : yy;
            if (dir == 1) p.BBAR[(row0 + t) * S + s] = hp * p.Osum[s];
        }
        __syncthreads();
        // four products that depend on fp and z only
        matvec_partial(pa, fp, dir == 0 ? p.S1 : p.S2, S, R, tid, nt);            // rr  = fp . (S1 | S2)
        matvec_partial(pb, z, dir == 0 ? p.S2 : p.S1, S, R, tid, nt);             // u   = z . (S2 | S1)
        matvec_partial(pc, z, dir == 0 ? p.WT : p.W, S, S, tid, nt);              // d fp through the wildcard matrix
        if (dir == 0) matvec_partial(pd, fp, p.W, S, S, tid, nt);                 // wildcard part of pre (for dOsum)
        __syncthreads();
        for (int r = tid; r < R; r += nt) {
            const float rv = part_sum(pa, R, r, nw), uv = part_sum(pb, R, r, nw), vv = v[r];
            rr[r] = rv; u[r] = uv;
            const float dd = uv * vv, tt = vv * rv;
            d1[r] = dd; tmpv[r] = tt;
            D1o[(long long)t * R + r] = dd;
            To[(long long)t * R + r] = tt;
            atomicAdd(p.dVgen + (long long)tok * R + r, uv * rv);                   // d v_t = u * rr
        }
        for (int s = tid; s < S; s += nt) g[s] = part_sum(pc, S, s, nw);
        __syncthreads();
        matvec_partial(pa, d1, dir == 0 ? p.S1T : p.S2T, R, S, tid, nt);           // d fp through the language factors
        if (dir == 0) matvec_partial(pb, tmpv, p.S2T, R, S, tid, nt);              // language part of pre (for dOsum)
        __syncthreads();
        for (int s = tid; s < S; s += nt) {
            const float dfp = g[s] + part_sum(pa, S, s, nw);
            if (dir == 0) {
                const float pre = part_sum(pb, S, s, nw) + part_sum(pd, S, s, nw);
                dO[s] = fmaf(y[s], pre, dO[s]);                                     // d Osum += y * pre_t
                g[s] = dfp;
            } else {
                dO[s] = fmaf(dfp, y[s], dO[s]);                                     // d Osum += d bbar * b_{t-1}
                g[s] = dfp * p.Osum[s];
            }
        }
        __syncthreads();
    }
    for (int s = tid; s < S; s += nt) {
        atomicAdd((dir == 0 ? p.dh0 : p.dhT) + s, g[s] + G[s]);
        atomicAdd(p.dOsum + s, dO[s]);
    }
}

// out[M][J] += sum_n A[n][M] B[n][J]   (A, B row-major with the reduction index as the row; rows that do not
// belong to a valid token are zero).  grid (ceil(M/32), ceil(J/32), splits); 256 threads, 2x2 outputs each.
__global__ void __launch_bounds__(256)
atb_accumulate_kernel(const float *__restrict__ A, const float *__restrict__ Bm, float *out, long long N, int M, int J,
                      long long chunk) {
    __shared__ float sa[32][33], sb[32][33];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const long long n0 = (long long)blockIdx.z * chunk, n1 = n0 + chunk < N ? n0 + chunk : N;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (long long nb = n0; nb < n1; nb += 32) {
        for (int e = tid; e < 32 * 32; e += 256) {
            const int rn = e >> 5, c = e & 31;
            const long long n = nb + rn;
            sa[rn][c] = (n < n1 && m0 + c < M) ? A[n * M + m0 + c] : 0.0f;
            sb[rn][c] = (n < n1 && j0 + c < J) ? Bm[n * J + j0 + c] : 0.0f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; k++) {
            const float a0 = sa[k][ty * 2], a1 = sa[k][ty * 2 + 1], b0 = sb[k][tx * 2], b1 = sb[k][tx * 2 + 1];
            acc[0][0] = fmaf(a0, b0, acc[0][0]); acc[0][1] = fmaf(a0, b1, acc[0][1]);
            acc[1][0] = fmaf(a1, b0, acc[1][0]); acc[1][1] = fmaf(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
    for (int a = 0; a < 2; a++)
        for (int c = 0; c < 2; c++) {
            const int m = m0 + ty * 2 + a, j = j0 + tx * 2 + c;
            if (m < M && j < J && acc[a][c] != 0.0f) atomicAdd(out + (long long)m * J + j, acc[a][c]);
        }
}

// dC[c][s] += dOsum[s] for every label row (Osum = C.sum(0))
__global__ void add_row_to_all_kernel(float *dC, const float *__restrict__ dOsum, int K, int S) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * S) return;
    dC[idx] += dOsum[idx % S];
}

}  // namespace farnn
