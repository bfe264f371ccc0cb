// K2 -- per-token label scores from the stashed forward/backward states, fused with the decode.
//
// Reference: the second loop of forward_score + get_final_score + local_decode
//   i-FST     model_onehot.py:346-349, :417-426, :162-180     score[c] = sum_s O[c,s] a[i+1][s] b~[i+1][s]
//   decomposed model_decompose_single.py:202-205, :263-269    (same form with C_output_mat)
//   decode    model_decompose.py:339-371 (argmax or CRF) ; crf.py:102-195 (Viterbi)
//
// One workgroup (8 wavefronts) per sequence; per sequence this is a small GEMM
// [tokens x S] . [S x K] whose right operand (the transposed output matrix) is shared by every
// token, so it is made LDS-resident once per workgroup and register-blocked: a wavefront scores
// 4 tokens at a time, a lane owns label columns {lane, lane+64, ...}; the a*b products of a
// 32-token tile are staged in LDS with coalesced 16-byte loads.  Scores never go to HBM unless
// the caller asks for them: threshold clamp, first-index argmax and the `oo -> o_idx` mapping --
// or the whole Viterbi DP with transitions, partitions and back-pointers in LDS -- run in the
// same kernel.
#pragma once
#include "common.hip.h"

namespace farnn {

struct ScoreParams {
    const float *A, *Bk;    // stash [B][L+1][SP]
    const float *OT;        // [S][Kc] transposed output matrix, columns >= K zero
    const float *P;         // [K][Kc] priority matrix or nullptr
    const float *trT;       // [K][Kp] TRANSPOSED CRF transitions trT[j][i] = tr[i][j], or nullptr
    const int64_t *len;     // [B]
    const int64_t *offs;    // [B+1] exclusive prefix of lengths (flat output) or nullptr
    int32_t *tags;          // [B][L] or nullptr
    int64_t *flat;          // [sum len] or nullptr
    float *scores;          // [B][L][K] or nullptr
    int B, L, S, SP, K, Kp, Kc, kch;
    int full, use_crf, o_idx;
    float threshold;
};

constexpr int SCORE_KCH = 4;       // label columns per lane: K <= 256
constexpr int SCORE_WAVES = 8;
constexpr int SCORE_TT = 32;       // tokens per LDS tile (4 per wavefront)

template <bool OT_LDS, bool TR_LDS>
__global__ void __launch_bounds__(SCORE_WAVES * 64)
score_decode_kernel(const ScoreParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    constexpr int nthreads = SCORE_WAVES * 64;
    const int b = blockIdx.x;
    const int len = (int)p.len[b];
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, K = p.K, Kp = p.Kp, Kc = p.Kc, kch = p.kch;

    // ---- LDS carve ---------------------------------------------------------------------------
    float *ab = smem;                                    // [TT][SP]  alpha*beta of the tile
    float *cur = ab + SCORE_TT * SP;
    float *scw = nullptr;                                // [waves][Kc] one score row per wave (P)
    if (p.P) { scw = cur; cur += SCORE_WAVES * Kc; }
    float *otl = nullptr;
    if (OT_LDS) {
        otl = cur; cur += S * Kc;
        for (int i = tid * 4; i < S * Kc; i += nthreads * 4) st4(otl + i, ld4(p.OT + i));
    }
    float *sc_all = nullptr, *part = nullptr, *trl = nullptr;
    unsigned short *bp = nullptr;
    if (p.use_crf) {
        sc_all = cur; cur += (size_t)p.L * Kp;           // [L][Kp] clamped scores of this sequence
        part = cur; cur += 2 * Kp;                       // [2][Kp]
        if (TR_LDS) {
            trl = cur; cur += K * Kp;
            for (int i = tid * 4; i < K * Kp; i += nthreads * 4) st4(trl + i, ld4(p.trT + i));
        }
        bp = reinterpret_cast<unsigned short *>(cur);    // [L][Kp] back-pointers
    }

    const float *Ab = p.A + (long long)b * (p.L + 1) * SP;
    const float *Bb = p.Bk + (long long)b * (p.L + 1) * SP;
    const int clamp_col = p.use_crf ? K - 3 : K - 1;      // model_decompose.py:353 / :365
    const long long foff = p.offs ? p.offs[b] : 0;
    const int SP4 = SP >> 2;

    for (int t0 = 0; t0 < nsteps; t0 += SCORE_TT) {
        const int nt = min(SCORE_TT, nsteps - t0);
        // ---- phase 1: ab[tok][s] = a[i+1][s] * b~[i+1][s]; alpha = state after i+1 tokens, beta =
        // backward state before token i+1 is consumed (reversed_backward_score_x[:, i+1], :415-420)
        for (int idx = tid; idx < SCORE_TT * SP4; idx += nthreads) {
            const int tok = idx / SP4, s4 = (idx - tok * SP4) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tok < nt) {
                const int i = t0 + tok;
                const int bidx = (i + 1 <= len) ? len - (i + 1) : i + 1;
                const float4 a4 = ld4(Ab + (long long)(i + 1) * SP + s4);
                const float4 b4 = ld4(Bb + (long long)bidx * SP + s4);
                v = make_float4(a4.x * b4.x, a4.y * b4.y, a4.z * b4.z, a4.w * b4.w);
            }
            st4(ab + tok * SP + s4, v);
        }
        __syncthreads();
        // ---- phase 2: 4 tokens per wavefront, register-blocked against the output matrix ------
        const int tg = w * 4;
        if (tg < nt) {
            float acc[4][SCORE_KCH];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = 0; k < SCORE_KCH; k++) acc[j][k] = 0.0f;
            const float *abw = ab + tg * SP;
            for (int s0 = 0; s0 < S; s0 += 4) {
                float av[4][4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float4 a4 = ld4(abw + j * SP + s0);          // LDS broadcast
                    av[j][0] = a4.x; av[j][1] = a4.y; av[j][2] = a4.z; av[j][3] = a4.w;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (s0 + u < S) {
                        const float *orow = (OT_LDS ? otl : p.OT) + (long long)(s0 + u) * Kc + lane;
#pragma unroll
                        for (int k = 0; k < SCORE_KCH; k++) {
                            if (k < kch) {
                                const float ov = orow[64 * k];
#pragma unroll
                                for (int j = 0; j < 4; j++) acc[j][k] = fmaf(av[j][u], ov, acc[j][k]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = t0 + tg + j;
                if (tg + j < nt) {
                    float sc[SCORE_KCH];
#pragma unroll
                    for (int k = 0; k < SCORE_KCH; k++) sc[k] = acc[j][k];
                    if (p.P) {      // PriorityLayer: scores @ P (priority.py:20-30)
                        float *sr = scw + w * Kc;
#pragma unroll
                        for (int k = 0; k < SCORE_KCH; k++) if (k < kch) sr[lane + 64 * k] = sc[k];
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int k = 0; k < SCORE_KCH; k++) sc[k] = 0.0f;
                        for (int cc = 0; cc < K; cc++) {
                            const float sv = sr[cc];
                            const float *prow = p.P + (long long)cc * Kc + lane;
#pragma unroll
                            for (int k = 0; k < SCORE_KCH; k++)
                                if (k < kch) sc[k] = fmaf(sv, prow[64 * k], sc[k]);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (p.scores) {
                        float *so = p.scores + ((long long)b * p.L + i) * K;
#pragma unroll
                        for (int k = 0; k < SCORE_KCH; k++) {
                            const int col = lane + 64 * k;
                            if (k < kch && col < K) so[col] = sc[k];
                        }
                    }
                    // threshold clamp of the `oo` column, then decode
                    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
                    for (int k = 0; k < SCORE_KCH; k++) {
                        const int col = lane + 64 * k;
                        if (k < kch && col < K) {
                            float v = sc[k];
                            if (col == clamp_col) v = fminf(v, p.threshold);
                            if (p.use_crf) sc_all[(long long)i * Kp + col] = v;
                            if (v > bv) { bv = v; bi = col; }
                        }
                    }
                    if (!p.use_crf) {
                        wave_argmax(bv, bi);
                        if (lane == 0) {
                            if (bi >= K) bi = 0;                    // all-NaN row: torch returns 0
                            const int tag = (bi == K - 1) ? p.o_idx : bi;
                            if (p.tags) p.tags[(long long)b * p.L + i] = tag;
                            if (p.flat && i < len) p.flat[foff + i] = tag;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // positions the recurrence did not visit (LOCAL mode pads)
    for (int i = nsteps + w; i < p.L; i += SCORE_WAVES) {
        if (p.tags && lane == 0) p.tags[(long long)b * p.L + i] = -1;
        if (p.scores)
            for (int col = lane; col < K; col += WAVE) p.scores[((long long)b * p.L + i) * K + col] = 0.0f;
    }

    if (!p.use_crf) return;

    // ---- Viterbi (crf.py:102-195) over the valid positions, per sequence ----------------------
    // A quad of lanes shares one destination tag j and splits the source tags i; the quad is
    // combined with (value desc, index asc), which is torch.max's first-index rule.
    __syncthreads();
    const float *trT = TR_LDS ? trl : p.trT;
    const int START = K - 2, STOP = K - 1;
    const int n = len;
    for (int j = tid; j < K; j += nthreads)
        part[j] = sc_all[j] + trT[(long long)j * Kp + START];                 // crf.py:135
    __syncthreads();
    int pc = 0;
    const int q = tid & 3;
    for (int t = 1; t < n; t++) {
        const float *pin = part + pc * Kp;
        float *pout = part + (pc ^ 1) * Kp;
        for (int j0 = 0; j0 < K; j0 += nthreads >> 2) {
            const int j = j0 + (tid >> 2);
            float best = -INFINITY; int bi = 0x7fffffff;
            if (j < K) {
                const float f = sc_all[(long long)t * Kp + j];
                const float *trow = trT + (long long)j * Kp;
                for (int i = q; i < K; i += 4) {
                    const float v = (f + trow[i]) + pin[i];                   // crf.py:123,145
                    if (v > best) { best = v; bi = i; }
                }
            }
#pragma unroll
            for (int off = 1; off <= 2; off <<= 1) {
                const float ov = __shfl_xor(best, off, WAVE);
                const int oi = __shfl_xor(bi, off, WAVE);
                argmax_combine(best, bi, ov, oi);
            }
            if (j < K && q == 0) {
                pout[j] = best;
                bp[(long long)t * Kp + j] = (unsigned short)(bi >= K ? 0 : bi);
            }
        }
        __syncthreads();
        pc ^= 1;
    }
    if (w == 0) {
        const float *pin = part + pc * Kp;
        const float *tstop = trT + (long long)STOP * Kp;
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = lane; i < K; i += WAVE) {
            const float v = pin[i] + tstop[i];                                // crf.py:168-169
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
        for (int i = n + tid; i < nsteps; i += nthreads) p.tags[(long long)b * p.L + i] = -1;
}

inline size_t score_lds_bytes(int S, int SP, int K, int Kp, int Kc, int L, int use_crf, int has_P,
                              int ot_in_lds, int tr_in_lds) {
    size_t f = (size_t)SCORE_TT * SP;
    if (has_P) f += (size_t)SCORE_WAVES * Kc;
    if (ot_in_lds) f += (size_t)S * Kc;
    size_t bytes = f * 4;
    if (use_crf) {
        bytes += ((size_t)L * Kp + 2 * Kp) * 4;
        if (tr_in_lds) bytes += (size_t)K * Kp * 4;
        bytes += (size_t)L * Kp * 2;
    }
    return bytes;
}

// exclusive prefix sum of lengths -> offs[B+1] (one workgroup; B is a batch size, not a corpus)
__global__ void __launch_bounds__(1024)
lengths_scan_kernel(const int64_t *len, int64_t *offs, int B) {
    __shared__ long long sums[1024];
    const int tid = threadIdx.x;
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

}  // namespace farnn
