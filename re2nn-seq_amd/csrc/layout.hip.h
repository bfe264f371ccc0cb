// One-time weight re-layout kernels run by the create() entry points.
//
// The reference re-materialises `language_tensor + wildcard_mat` (19 MB at ATIS size, 2.5 GB for
// the 4-D FST) on EVERY forward call (model_onehot.py:82,87,250,366).  Weights are frozen on the
// tagging path, so the sum is taken once here, rows are padded to a multiple of 16 bytes (every
// row then starts on a dwordx4 boundary) and a transposed copy is written for the backward chain.
#pragma once
#include "common.hip.h"

namespace farnn {

// dst[c][r] = src[r][c]; dst rows padded to rows_p
__global__ void transpose_pad_kernel(const float *src, float *dst, int rows, int cols, int rows_p) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    int r = i / cols, c = i - r * cols;
    dst[(long long)c * rows_p + r] = src[i];
}

// dst[c] = sum_r src[r][c]   (ascending r, fp32)
__global__ void colsum_kernel(const float *src, float *dst, int rows, int cols) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.0f;
    for (int r = 0; r < rows; r++) s += src[(long long)r * cols + c];
    dst[c] = s;
}

// Mf[v][s][jp] = (T[v][s][j] + W[s][j]) * mask[s][j];  Mb[v][j][sp] = the same value transposed.
// grid = (ceil(S*SP/256), V)
__global__ void premix_kernel(const float *T, const float *W, const float *mask, float *Mf, float *Mb,
                              int S, int SP, int SR) {
    const long long v = blockIdx.y;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= SR * SP) return;
    int r = idx / SP, cpad = idx - r * SP;
    const float *Tv = T + v * S * S;
    float f = 0.0f, bwd = 0.0f;
    if (cpad < S && r < S) {
        f = Tv[r * S + cpad] + W[r * S + cpad];
        if (mask) f *= mask[r * S + cpad];
        bwd = Tv[cpad * S + r] + W[cpad * S + r];
        if (mask) bwd *= mask[cpad * S + r];
    }
    Mf[v * SR * SP + idx] = f;
    if (Mb) Mb[v * SR * SP + idx] = bwd;
}

// SR >= S: rows allocated per block (extra rows zero-filled; see chain.hip.h)
inline int launch_premix(const float *T, const float *W, const float *mask, float *Mf, float *Mb,
                         int V, int S, int SP, int SR) {
    dim3 grid((SR * SP + 255) / 256, V);
    premix_kernel<<<grid, 256>>>(T, W, mask, Mf, Mb, S, SP, SR);
    FARNN_HIP_TRY(hipGetLastError());
    FARNN_HIP_TRY(hipDeviceSynchronize());
    return FARNN_OK;
}

// 4-D FST (model_onehot.py:82, :87):
//   Ts[v] = sum_c T4[v][c] + sum_c W4[c]        -> chain blocks Mf / Mb
//   A4[v][c] = T4[v][c] + W4[c]                 -> scoring stream, rows padded
// grid = (ceil(S*SP/256), V)
__global__ void premix_fst4_kernel(const float *T4, const float *W4, float *Mf, float *Mb, float *A4,
                                   int C, int S, int SP, int SR) {
    const long long v = blockIdx.y;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= SR * SP) return;
    int r = idx / SP, cpad = idx - r * SP;
    if (r >= S) {                       // zero-filled padding rows of the chain blocks
        Mf[v * SR * SP + idx] = 0.0f;
        Mb[v * SR * SP + idx] = 0.0f;
        return;
    }
    const long long SS = (long long)S * S;
    float tsum = 0.0f, wsum = 0.0f, tsum_b = 0.0f, wsum_b = 0.0f;
    for (int c = 0; c < C; c++) {
        float a = 0.0f;
        if (cpad < S) {
            float t = T4[(v * C + c) * SS + r * S + cpad];
            float w = W4[c * SS + r * S + cpad];
            tsum += t; wsum += w; a = t + w;
            tsum_b += T4[(v * C + c) * SS + cpad * S + r];
            wsum_b += W4[c * SS + cpad * S + r];
        }
        A4[((v * C + c) * S) * SP + idx] = a;
    }
    Mf[v * SR * SP + idx] = tsum + wsum;
    Mb[v * SR * SP + idx] = tsum_b + wsum_b;
}

inline int launch_premix_fst4(const float *T4, const float *W4, float *Mf, float *Mb, float *A4,
                              int V, int C, int S, int SP, int SR) {
    dim3 grid((SR * SP + 255) / 256, V);
    premix_fst4_kernel<<<grid, 256>>>(T4, W4, Mf, Mb, A4, C, S, SP, SR);
    FARNN_HIP_TRY(hipGetLastError());
    FARNN_HIP_TRY(hipDeviceSynchronize());
    return FARNN_OK;
}

// ---- automaton edge list -> dense device tensors (fsa_to_tensor.py:398-615 without the host tensors) ----
// mode 0: i-FST  T[V,S,S] W[S,S] O[C,S];  mode 1: FST 4-D  T = T4[V,C,S,S], W = W4[C,S,S];
// mode 2: independent=1  T[V,S,S] W[S,S] O = Oten[C,S,S]
__global__ void scatter_edges_kernel(const int32_t *word, const int32_t *from, const int32_t *to,
                                     const int32_t *label, const float *val, long long n,
                                     float *T, float *W, float *O, int V, int S, int C, int mode, int *bad) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int w = word[e], f = from[e], t = to[e], l = label ? label[e] : -1;
    if (w >= V || f < 0 || f >= S || t < 0 || t >= S || l >= C || (mode == 1 && w >= -1 && l < 0)) {
        atomicExch(bad, 1);
        return;
    }
    const float v = val ? val[e] : 1.0f;
    if (mode == 1) {
        if (w >= 0) T[(((long long)w * C + l) * S + f) * S + t] = v;
        else if (w == -1) W[((long long)l * S + f) * S + t] = v;
        return;
    }
    if (w >= 0) { if (T) T[((long long)w * S + f) * S + t] = v; }           // (T == nullptr: labels only)
    else if (w == -1) { if (W) W[(long long)f * S + t] = v; }                // w < -1: label only
    if (l >= 0) {
        if (mode == 0) O[(long long)l * S + t] = 1.0f;
        else O[((long long)l * S + f) * S + t] = 1.0f;
    }
}

// ---- decomposed model -> dense per-word transition blocks (max-times semiring / independent=1, farnn==0) ----
// The step's transition matrix of these modes is materialised from the factors,
//     Tr[x][i][j] = (sum_r v_x[r] S1[i][r] S2[j][r] + W[i][j]) * mask[i][j]
// (model_decompose_single.py:159-166, model_decompose_independent.py:165-173), and it depends on the WORD
// only.  With 288 GB of HBM the whole table (V x S x S fp32, + its transpose: 0.96 GB at V=11k, S=104) is
// built once at create time, and the recurrence becomes the dense chain kernel: HBM-streamed blocks instead
// of S*S*R multiply-adds per step and sequence.  Same expression order as the generic kernel (fmaf over r,
// + W, * mask): identical table entries.   grid = (ceil(SR*SP/256), V)
__global__ void materialise_blocks_kernel(const float *Vgen, const float *S1, const float *S2, const float *W,
                                          const float *mask, float *Mf, float *Mb, int S, int SP, int SR,
                                          int R, int Rp) {
    const long long v = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= SR * SP) return;
    const int i = idx / SP, j = idx - i * SP;
    float tr = 0.0f;
    if (i < S && j < S) {
        const float *vg = Vgen + v * Rp, *s1 = S1 + (long long)i * Rp, *s2 = S2 + (long long)j * Rp;
        for (int r = 0; r < R; r++) tr = fmaf(vg[r] * s1[r], s2[r], tr);
        tr += W[(long long)i * SP + j];
        if (mask) tr *= mask[(long long)i * SP + j];
    }
    Mf[v * SR * SP + idx] = tr;                          // Mf[v][i][j] = Tr[i][j]
    if (Mb && i < SP && j < SR) Mb[v * SR * SP + (long long)j * SP + i] = (i < S && j < S) ? tr : 0.0f;   // Mb[v][j][i]
}

}  // namespace farnn
