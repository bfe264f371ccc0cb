// K2 -- per-token label scores from the stashed forward/backward states, fused with the decode.
//
// Reference: the second loop of forward_score + get_final_score + local_decode
//   i-FST     model_onehot.py:346-349, :417-426, :162-180     score[c] = sum_s O[c,s] a[i+1][s] b~[i+1][s]
//   decomposed model_decompose_single.py:202-205, :263-269    (same form with C_output_mat)
//   decode    model_decompose.py:339-371 (argmax or CRF) ; crf.py:102-195 (Viterbi)
//
// One workgroup per sequence.  Each wavefront takes tokens round-robin; a lane owns label
// columns {lane, lane+64, ...}.  The output matrix is read TRANSPOSED ([S][Kp], lanes on
// consecutive labels) from LDS when it fits, else from L2.  Scores never go to HBM unless the
// caller asks for them: the threshold clamp, first-index argmax and the `oo -> o_idx` mapping
// (or the whole Viterbi DP with its transition table and back-pointers in LDS) run in the same
// kernel.
#pragma once
#include "common.hip.h"

namespace farnn {

struct ScoreParams {
    const float *A, *Bk;    // stash [B][L+1][SP]
    const float *OT;        // [S][Kp] transposed output matrix (rows >= K zero)
    const float *P;         // [K][Kp] priority matrix or nullptr
    const float *tr;        // [K][Kp] CRF transitions (use_crf) or nullptr
    const int64_t *len;     // [B]
    const int64_t *offs;    // [B+1] exclusive prefix of lengths (flat output) or nullptr
    int32_t *tags;          // [B][L] or nullptr
    int64_t *flat;          // [sum len] or nullptr
    float *scores;          // [B][L][K] or nullptr
    int B, L, S, SP, K, Kp;
    int full, use_crf, o_idx;
    int ot_in_lds, tr_in_lds;
    float threshold;
};

constexpr int SCORE_KCH = 4;       // label columns per lane: K <= 256
constexpr int SCORE_WAVES = 8;

__global__ void __launch_bounds__(SCORE_WAVES * 64)
score_decode_kernel(const ScoreParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int b = blockIdx.x;
    const int len = (int)p.len[b];
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, K = p.K, Kp = p.Kp;

    // ---- LDS carve ---------------------------------------------------------------------------
    float *abw = smem;                                   // [nwaves][SP]  alpha*beta per wave
    float *scw = abw + nwaves * SP;                      // [nwaves][Kp]  one score row per wave
    float *cur = scw + nwaves * Kp;
    const float *OT = p.OT;
    if (p.ot_in_lds) {
        float *ot = cur; cur += S * Kp;
        for (int i = tid * 4; i < S * Kp; i += nthreads * 4) st4(ot + i, ld4(p.OT + i));
        OT = ot;
    }
    float *sc_all = nullptr, *part = nullptr;
    const float *tr = p.tr;
    unsigned short *bp = nullptr;
    if (p.use_crf) {
        sc_all = cur; cur += (size_t)p.L * Kp;           // [L][Kp] clamped scores of this sequence
        part = cur; cur += 2 * Kp;                       // [2][Kp]
        if (p.tr_in_lds) {
            float *t2 = cur; cur += K * Kp;
            for (int i = tid * 4; i < K * Kp; i += nthreads * 4) st4(t2 + i, ld4(p.tr + i));
            tr = t2;
        }
        bp = reinterpret_cast<unsigned short *>(cur);    // [L][Kp] back-pointers
    }
    __syncthreads();

    const float *Ab = p.A + (long long)b * (p.L + 1) * SP;
    const float *Bb = p.Bk + (long long)b * (p.L + 1) * SP;
    const int clamp_col = p.use_crf ? K - 3 : K - 1;      // model_decompose.py:353 / :365
    const long long foff = p.offs ? p.offs[b] : 0;

    for (int i = w; i < nsteps; i += nwaves) {
        // alpha = state after i+1 tokens; beta = backward state before token i+1 is consumed
        // (reference: reversed_backward_score_x[:, i+1], model_onehot.py:415-420)
        const int bidx = (i + 1 <= len) ? len - (i + 1) : i + 1;
        const float *ar = Ab + (long long)(i + 1) * SP;
        const float *br = Bb + (long long)bidx * SP;
        float *ab = abw + w * SP;
        for (int s = lane * 4; s < SP; s += WAVE * 4) {
            float4 a4 = ld4(ar + s), b4 = ld4(br + s);
            st4(ab + s, make_float4(a4.x * b4.x, a4.y * b4.y, a4.z * b4.z, a4.w * b4.w));
        }
        __builtin_amdgcn_wave_barrier();
        float acc[SCORE_KCH];
#pragma unroll
        for (int k = 0; k < SCORE_KCH; k++) acc[k] = 0.0f;
        for (int s0 = 0; s0 < S; s0 += 4) {
            const float4 a4 = ld4(ab + s0);               // LDS broadcast; rows >= S are zero
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (s0 + u < S) {
                    const float *orow = OT + (long long)(s0 + u) * Kp;
#pragma unroll
                    for (int k = 0; k < SCORE_KCH; k++) {
                        int col = lane + 64 * k;
                        if (col < Kp) acc[k] = fmaf(av[u], orow[col], acc[k]);
                    }
                }
            }
        }
        if (p.P) {      // PriorityLayer: scores @ P (priority.py:20-30)
            float *sr = scw + w * Kp;
#pragma unroll
            for (int k = 0; k < SCORE_KCH; k++) {
                int col = lane + 64 * k;
                if (col < Kp) sr[col] = acc[k];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < SCORE_KCH; k++) acc[k] = 0.0f;
            for (int cc = 0; cc < K; cc++) {
                float sv = sr[cc];
                const float *prow = p.P + (long long)cc * Kp;
#pragma unroll
                for (int k = 0; k < SCORE_KCH; k++) {
                    int col = lane + 64 * k;
                    if (col < Kp) acc[k] = fmaf(sv, prow[col], acc[k]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (p.scores) {
            float *so = p.scores + ((long long)b * p.L + i) * K;
#pragma unroll
            for (int k = 0; k < SCORE_KCH; k++) {
                int col = lane + 64 * k;
                if (col < K) so[col] = acc[k];
            }
        }
        // threshold clamp of the `oo` column, then decode
        float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < SCORE_KCH; k++) {
            int col = lane + 64 * k;
            if (col < K) {
                float v = acc[k];
                if (col == clamp_col) v = fminf(v, p.threshold);
                if (p.use_crf) sc_all[(long long)i * Kp + col] = v;
                if (v > bv) { bv = v; bi = col; }
            }
        }
        if (!p.use_crf) {
            wave_argmax(bv, bi);
            if (lane == 0) {
                if (bi >= K) bi = 0;                        // all-NaN row: torch returns index 0
                int tag = (bi == K - 1) ? p.o_idx : bi;
                if (p.tags) p.tags[(long long)b * p.L + i] = tag;
                if (p.flat && i < len) p.flat[foff + i] = tag;
            }
        }
    }

    // positions the recurrence did not visit (LOCAL mode pads)
    for (int i = nsteps + w; i < p.L; i += nwaves) {
        if (p.tags && lane == 0) p.tags[(long long)b * p.L + i] = -1;
        if (p.scores)
            for (int col = lane; col < K; col += WAVE) p.scores[((long long)b * p.L + i) * K + col] = 0.0f;
    }

    if (!p.use_crf) return;

    // ---- Viterbi (crf.py:102-195) over the valid positions, per sequence ----------------------
    __syncthreads();
    const int START = K - 2, STOP = K - 1;
    const int n = len;
    for (int j = tid; j < K; j += nthreads)
        part[j] = sc_all[j] + tr[(long long)START * Kp + j];                 // crf.py:135
    __syncthreads();
    int pc = 0;
    for (int t = 1; t < n; t++) {
        const float *pin = part + pc * Kp;
        float *pout = part + (pc ^ 1) * Kp;
        for (int j = tid; j < K; j += nthreads) {
            const float f = sc_all[(long long)t * Kp + j];
            float best = -INFINITY; int bi = 0;
            for (int i = 0; i < K; i++) {
                float v = (f + tr[(long long)i * Kp + j]) + pin[i];           // crf.py:123,145
                if (v > best) { best = v; bi = i; }
            }
            pout[j] = best;
            bp[(long long)t * Kp + j] = (unsigned short)bi;
        }
        __syncthreads();
        pc ^= 1;
    }
    if (w == 0) {
        const float *pin = part + pc * Kp;
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = lane; i < K; i += WAVE) {
            float v = pin[i] + tr[(long long)i * Kp + STOP];                  // crf.py:168-169
            if (v > bv) { bv = v; bi = i; }
        }
        wave_argmax(bv, bi);
        if (lane == 0) {
            if (bi >= K) bi = 0;
            int ptr = bi;
            for (int t = n - 1; t >= 0; t--) {
                int tag = (ptr == K - 3) ? p.o_idx : ptr;                     // model_decompose.py:356
                if (p.tags) p.tags[(long long)b * p.L + t] = tag;
                if (p.flat) p.flat[foff + t] = tag;
                if (t > 0) ptr = bp[(long long)t * Kp + ptr];
            }
        }
    }
    if (p.tags)
        for (int i = n + tid; i < nsteps; i += nthreads) p.tags[(long long)b * p.L + i] = -1;
}

inline size_t score_lds_bytes(int S, int SP, int K, int Kp, int L, int use_crf, int ot_in_lds,
                              int tr_in_lds) {
    size_t f = (size_t)SCORE_WAVES * SP + (size_t)SCORE_WAVES * Kp;
    if (ot_in_lds) f += (size_t)S * Kp;
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
