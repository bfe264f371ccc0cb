// K3 -- bilinear per-token scoring for the dense FST layouts, fused with the argmax decode.
//
//   FST 4-D  (FARNN_S_O, model_onehot.py:115-127):
//       score[c] = sum_{s,j} relu( A[x_i, c, s, j] * a_i[s] * b~_{i+1}[j] )       A = T4 + W4
//   independent=1 (FARNN_S_O_I, model_onehot.py:229-233, :293-304):
//       score[c] = sum_{s,j} Oten[c,s,j] * (a_i[s] * b~_{i+1}[j]) * Tf[x_i,s,j]
//   with a_i = forward state BEFORE token i (h0_forward_score[:, i], :117 / :296).
//
// This is the genuinely HBM-streamed kernel of the repo: the 4-D layout reads C*S*S*4 bytes per
// token (2.6 MB at ATIS size) exactly once.  One workgroup per token; wavefronts split the label
// columns; a lane owns a fixed 16-byte column chunk (so its beta values live in registers) and
// walks the rows of the label's S x SP slice with coalesced dwordx4 loads.
#pragma once
#include "common.hip.h"

namespace farnn {

struct Fst4Params {
    const float *blocks;    // FST4: A4 [V][C][S][SP];  IND1: Tf [V][S][SP]
    const float *Oten;      // IND1: [C][S][SP], else nullptr
    const float *A, *Bk;    // stash [B][L+1][SP]
    const float *P;         // [C][Kp] or nullptr
    const int64_t *x, *len, *offs;
    int32_t *tags; int64_t *flat; float *scores;
    int B, L, S, SP, C, Kp, full, o_idx, V;
    int G, LPR, CPR;
    float threshold;
};

template <int NCH>
__global__ void __launch_bounds__(256)
fst4_score_kernel(const Fst4Params p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, nwaves = blockDim.x >> 6;
    const int i = blockIdx.x, b = blockIdx.y;
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, C = p.C;
    if (i >= nsteps) {
        if (p.tags && tid == 0) p.tags[(long long)b * p.L + i] = -1;
        if (p.scores) for (int c = tid; c < C; c += blockDim.x) p.scores[((long long)b * p.L + i) * C + c] = 0.0f;
        return;
    }
    float *alpha = smem;                 // [SP]
    float *sc = alpha + SP;              // [Kp]
    float *sc2 = sc + p.Kp;              // [Kp]
    float *Z = sc2 + p.Kp;               // IND1 only: [S][SP]

    const float *ar = p.A + ((long long)b * (p.L + 1) + i) * SP;
    const int bidx = (i + 1 <= len) ? len - (i + 1) : i + 1;
    const float *br = p.Bk + ((long long)b * (p.L + 1) + bidx) * SP;
    for (int s = tid; s < SP; s += blockDim.x) alpha[s] = ar[s];

    int g = lane / p.LPR;
    const int c4 = lane - g * p.LPR;
    const bool active = g < p.G;
    float4 beta[NCH];
#pragma unroll
    for (int m = 0; m < NCH; m++) {
        int cc = c4 + 64 * m;
        beta[m] = (active && cc < p.CPR) ? ld4(br + cc * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long long tok = clamp_tok(p.x[(long long)b * p.L + i], p.V);
    __syncthreads();

    if (p.Oten) {   // Z = (alpha beta^T) .* Tf[x_i]   (model_onehot.py:230-231)
        const float *tb = p.blocks + tok * S * SP;
        for (int row = w * p.G + g; row < S; row += nwaves * p.G) {
            if (active) {
                const float a = alpha[row];
#pragma unroll
                for (int m = 0; m < NCH; m++) {
                    int cc = c4 + 64 * m;
                    if (cc < p.CPR) {
                        float4 t = ld4(tb + (long long)row * SP + cc * 4);
                        st4(Z + row * SP + cc * 4, make_float4((a * beta[m].x) * t.x, (a * beta[m].y) * t.y,
                                                                (a * beta[m].z) * t.z, (a * beta[m].w) * t.w));
                    }
                }
            }
        }
        __syncthreads();
    }

    for (int c = w; c < C; c += nwaves) {
        float sum = 0.0f;
        if (p.Oten) {
            const float *ob = p.Oten + (long long)c * S * SP;
            if (active)
                for (int row = g; row < S; row += p.G) {
#pragma unroll
                    for (int m = 0; m < NCH; m++) {
                        int cc = c4 + 64 * m;
                        if (cc < p.CPR) {
                            float4 o4 = ld4(ob + (long long)row * SP + cc * 4);
                            float4 z4 = ld4(Z + row * SP + cc * 4);
                            sum = fmaf(o4.x, z4.x, sum); sum = fmaf(o4.y, z4.y, sum);
                            sum = fmaf(o4.z, z4.z, sum); sum = fmaf(o4.w, z4.w, sum);
                        }
                    }
                }
        } else {
            const float *ab = p.blocks + (tok * C + c) * S * SP;
            if (active)
                for (int row = g; row < S; row += p.G) {
                    const float a = alpha[row];
#pragma unroll
                    for (int m = 0; m < NCH; m++) {
                        int cc = c4 + 64 * m;
                        if (cc < p.CPR) {
                            float4 v = ld4(ab + (long long)row * SP + cc * 4);
                            sum += fmaxf((v.x * a) * beta[m].x, 0.0f);       // :119-121
                            sum += fmaxf((v.y * a) * beta[m].y, 0.0f);
                            sum += fmaxf((v.z * a) * beta[m].z, 0.0f);
                            sum += fmaxf((v.w * a) * beta[m].w, 0.0f);
                        }
                    }
                }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, WAVE);
        if (lane == 0) sc[c] = sum;
    }
    __syncthreads();
    const float *fin = sc;
    if (p.P) {
        for (int d = tid; d < C; d += blockDim.x) {
            float s = 0.0f;
            for (int c = 0; c < C; c++) s = fmaf(sc[c], p.P[(long long)c * p.Kp + d], s);
            sc2[d] = s;
        }
        __syncthreads();
        fin = sc2;
    }
    if (p.scores)
        for (int c = tid; c < C; c += blockDim.x) p.scores[((long long)b * p.L + i) * C + c] = fin[c];
    if (w == 0) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int c = lane; c < C; c += WAVE) {
            float v = fin[c];
            if (c == C - 1) v = fminf(v, p.threshold);
            if (v > bv) { bv = v; bi = c; }
        }
        wave_argmax(bv, bi);
        if (lane == 0) {
            if (bi >= C) bi = 0;
            int tag = (bi == C - 1) ? p.o_idx : bi;
            if (p.tags) p.tags[(long long)b * p.L + i] = tag;
            if (p.flat && i < len) p.flat[p.offs[b] + i] = tag;
        }
    }
}

inline int launch_fst4_score(const float *blocks, const float *A, const float *Bk, const float *P,
                             const int64_t *x, const int64_t *len, const int64_t *offs, int32_t *tags,
                             int64_t *flat, float *scores, int B, int L, int S, int SP, int C, int Kp,
                             int full, int o_idx, float threshold, const float *Oten, int V, hipStream_t s) {
    Fst4Params p;
    p.V = V;
    p.blocks = blocks; p.Oten = Oten; p.A = A; p.Bk = Bk; p.P = P; p.x = x; p.len = len; p.offs = offs;
    p.tags = tags; p.flat = flat; p.scores = scores;
    p.B = B; p.L = L; p.S = S; p.SP = SP; p.C = C; p.Kp = Kp; p.full = full; p.o_idx = o_idx;
    p.threshold = threshold;
    p.CPR = SP / 4;
    int nch;
    if (p.CPR <= 64) { nch = 1; p.LPR = p.CPR; p.G = 64 / p.CPR; }
    else { nch = (p.CPR + 63) / 64; p.LPR = 64; p.G = 1; }
    size_t lds = ((size_t)SP + 2 * (size_t)Kp + (Oten ? (size_t)S * SP : 0)) * sizeof(float);
    if (lds > 160 * 1024) return fail(FARNN_ERANGE, "independent=1 scoring needs S*S*4 bytes of LDS%s%s");
    dim3 grid(L, B), block(256);
#define FARNN_LAUNCH_FST4(N)                                                                          \
    do {                                                                                              \
        if (lds > 48 * 1024)                                                                          \
            FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fst4_score_kernel<N>),   \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        fst4_score_kernel<N><<<grid, block, lds, s>>>(p);                                             \
    } while (0)
    if (nch == 1) FARNN_LAUNCH_FST4(1);
    else if (nch == 2) FARNN_LAUNCH_FST4(2);
    else if (nch <= 4) FARNN_LAUNCH_FST4(4);
    else return fail(FARNN_ERANGE, "more than 1024 states%s%s");
#undef FARNN_LAUNCH_FST4
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

}  // namespace farnn
