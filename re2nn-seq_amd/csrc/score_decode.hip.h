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
#include "common.hip.h"

namespace farnn {

struct ScoreParams {
    const float *A, *Bk;    // stash [B][L+1][SP]
    const float *OT;        // [S][Kc] transposed output matrix, columns >= K zero (alloc padded to 1 KiB)
    const float *P;         // [K][Kc] priority matrix or nullptr
    const float *trT;       // [K][Kp] TRANSPOSED CRF transitions trT[j][i] = tr[i][j], or nullptr
    const int64_t *len;     // [B]
    const int64_t *offs;    // [B+1] exclusive prefix of lengths (flat output) or nullptr: with `flat` set the
                            //        kernel then sums the lengths before its sequence itself (B <= 1024)
    int32_t *tags;          // [B][L] or nullptr
    int64_t *flat;          // [sum len] or nullptr
    float *scores;          // [B][L][K] or nullptr (unclamped, what forward_score returns)
    float *crf_scores;      // [B][L][Kp] workspace: clamped scores for the Viterbi kernel
    int B, L, S, SP, K, Kp, Kc, kch;
    int full, use_crf, o_idx;
    float threshold;
    int dbg;                // diagnostic ablation mask (FARNN_DBG bits 16/32/64); 0 in production
};

constexpr int SCORE_KCH = 4;       // label columns per lane: K <= 256
constexpr int SCORE_WAVES = 8;
constexpr int SCORE_TT = 32;       // tokens per tile (4 per wavefront)

// KCH = label columns per lane (K <= 64*KCH): compile-time so that the register-blocked loops are
// fully unrolled with no per-column guards (with a runtime bound hipcc emitted one scalar branch
// per column and the GEMM loop ran 3x slower).
//
// score_tiles: the tiles tile_first, tile_first + tile_step, ... (32 tokens each) of sequence b, by the 8 wavefronts
// of the calling workgroup (every thread calls it; `smem` = 16-byte aligned LDS of score_lds_bytes()).  The output
// matrix is staged once for all the tiles.  SC1: the stash is read with agent-scope (sc1) loads -- the form the
// fused epilogue of chain_kernel needs, where another workgroup of the same launch wrote half of it.
// where score_tiles keeps the transposed output matrix inside `smem`
__device__ __forceinline__ float *score_ot_lds(float *smem, int SP, int Kc, bool has_P) {
    return smem + SCORE_TT * SP + (has_P ? SCORE_WAVES * Kc : 0);
}

// LDS-DMA of the transposed output matrix into its place (1 KiB pieces, round-robin over the eight wavefronts); the
// issuing wavefront's next `s_waitcnt vmcnt(0)` covers its pieces, a workgroup barrier behind that everybody's
__device__ __forceinline__ void score_stage_ot(const float *OT, float *otl, int S, int Kc, int w, int lane) {
    const unsigned ot_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)otl);
    const int pieces = (S * Kc * 4 + 1023) / 1024;
    const char *obase = reinterpret_cast<const char *>(OT);
    for (int k = w; k < pieces; k += SCORE_WAVES)
        lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, obase, ot_lds + (unsigned)k * 1024u);
}

// foff_pre >= 0: the sequence's offset in the flat output is already known; ot_prestaged: the caller issued
// score_stage_ot() itself (from these same wavefronts) -- both let the fused epilogue overlap them with its hand-off
template <bool OT_LDS, int KCH, bool SC1>
__device__ __forceinline__ void score_tiles(const ScoreParams &p, const int b, const int tile_first, const int tile_step,
                                            float *smem, const int tid, const long long foff_pre = -1,
                                            const bool ot_prestaged = false) {
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nthreads = SCORE_WAVES * 64;
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, K = p.K, Kc = p.Kc;
    const int ntiles = (p.L + SCORE_TT - 1) / SCORE_TT;

    // ---- LDS carve ---------------------------------------------------------------------------
    float *ab = smem;                                    // [TT][SP]  alpha*beta of the tile
    float *cur = ab + SCORE_TT * SP;
    float *scw = nullptr;                                // [waves][Kc] one score row per wave (P)
    if (p.P) { scw = cur; cur += SCORE_WAVES * Kc; }
    float *otl = cur;                                    // [S][Kc] rounded up to whole DMA pieces
    bool dma_pending = ot_prestaged;
    if (OT_LDS && !ot_prestaged && !(p.dbg & 16) && tile_first * SCORE_TT < nsteps) {
        score_stage_ot(p.OT, otl, S, Kc, w, lane);       // lands while phase 1 runs
        dma_pending = true;
    }
    // no prepared offsets: where this sequence starts in the flat output = sum of the lengths before it
    // (utils.py:153-164); B <= 1024 here, two loads per thread, hidden behind the DMA
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
        // ---- phase 1: ab[tok][s] = a[i+1][s] * b~[i+1][s]; alpha = state after i+1 tokens, beta =
        // backward state before token i+1 is consumed (reversed_backward_score_x[:, i+1], :415-420)
        for (int idx0 = 0; idx0 < SCORE_TT * SP4; idx0 += 2 * nthreads) {
            float4 a4[2], b4[2];
            int tokv[2], s4v[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int idx = idx0 + r * nthreads + tid;
                const int tok = idx / SP4;
                tokv[r] = tok; s4v[r] = (idx - tok * SP4) * 4;
                a4[r] = make_float4(0.f, 0.f, 0.f, 0.f); b4[r] = a4[r];
                if (tok < nt) {
                    const int i = t0 + tok;
                    const int bidx = (i + 1 <= len) ? len - (i + 1) : i + 1;
                    if (SC1) {
                        a4[r] = ld4_agent(Ab + (long long)(i + 1) * SP + s4v[r]);
                        b4[r] = ld4_agent(Bb + (long long)bidx * SP + s4v[r]);
                    } else {
                        a4[r] = ld4(Ab + (long long)(i + 1) * SP + s4v[r]);
                        b4[r] = ld4(Bb + (long long)bidx * SP + s4v[r]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (tokv[r] < SCORE_TT)
                    st4(ab + tokv[r] * SP + s4v[r], make_float4(a4[r].x * b4[r].x, a4[r].y * b4[r].y,
                                                                 a4[r].z * b4[r].z, a4[r].w * b4[r].w));
        }
        if (dma_pending) {                                   // this wavefront's DMA pieces landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dma_pending = false;
        }
        __syncthreads();

        // ---- phase 2: 4 tokens per wavefront, register-blocked against the output matrix ----------
        const int tg = w * 4;
        float acc[4][KCH];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < KCH; k++) acc[j][k] = 0.0f;
        if (tg < nt && !(p.dbg & 32)) {
            const float *abw = ab + tg * SP;
            const float *otb = (OT_LDS ? otl : p.OT) + lane;
            for (int s0 = 0; s0 < S; s0 += 4) {
                float av[4][4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float4 a4 = ld4(abw + j * SP + s0);              // LDS broadcast
                    av[j][0] = a4.x; av[j][1] = a4.y; av[j][2] = a4.z; av[j][3] = a4.w;
                }
                float ov[4][KCH];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int srow = (s0 + u < S) ? s0 + u : S - 1;        // ab pad columns are zero
#pragma unroll
                    for (int k = 0; k < KCH; k++)
                        ov[u][k] = otb[(long long)srow * Kc + 64 * k];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int k = 0; k < KCH; k++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[j][k] = fmaf(av[j][u], ov[u][k], acc[j][k]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = t0 + tg + j;
            if (tg + j < nt) {
                float sc[KCH];
#pragma unroll
                for (int k = 0; k < KCH; k++) sc[k] = acc[j][k];
                if (p.P) {      // PriorityLayer: scores @ P (priority.py:20-30)
                    float *sr = scw + w * Kc;
#pragma unroll
                    for (int k = 0; k < KCH; k++) sr[lane + 64 * k] = sc[k];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < KCH; k++) sc[k] = 0.0f;
                    for (int cc = 0; cc < K; cc++) {
                        const float sv = sr[cc];
                        const float *prow = p.P + (long long)cc * Kc + lane;
#pragma unroll
                        for (int k = 0; k < KCH; k++)
                            sc[k] = fmaf(sv, prow[64 * k], sc[k]);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (p.scores) {
                    float *so = p.scores + ((long long)b * p.L + i) * K;
#pragma unroll
                    for (int k = 0; k < KCH; k++) {
                        const int col = lane + 64 * k;
                        if (col < K) so[col] = sc[k];
                    }
                }
                // threshold clamp of the `oo` column, then decode
                float bv = -INFINITY; int bi = 0x7ffffffe;
#pragma unroll
                for (int k = 0; k < KCH; k++) {
                    const int col = lane + 64 * k;
                    if (col < K) {
                        float v = sc[k] + 0.0f;                      // -0.0 -> +0.0 (torch: -0 == +0)
                        if (col == clamp_col) v = fminf(v, p.threshold);
                        if (p.use_crf) p.crf_scores[((long long)b * p.L + i) * p.Kp + col] = v;
                        if (v > bv) { bv = v; bi = col; }
                    }
                }
                if (!p.use_crf) {
                    if (!(p.dbg & 64)) bi = wave_argmax_dpp(bv, bi);
                    if (lane == 0) {
                        if (bi >= K) bi = 0;                        // all-NaN row: torch returns 0
                        const int tag = (bi == K - 1) ? p.o_idx : bi;
                        if (p.tags) p.tags[(long long)b * p.L + i] = tag;
                        if (p.flat && i < len) p.flat[foff + i] = tag;
                    }
                }
            } else if (tg + j < ntL) {      // pad position inside a partly valid tile (LOCAL mode)
                if (p.tags && lane == 0) p.tags[(long long)b * p.L + i] = -1;
                if (p.scores)
                    for (int col = lane; col < K; col += WAVE) p.scores[((long long)b * p.L + i) * K + col] = 0.0f;
            }
        }
        if (tile + tile_step < ntiles) __syncthreads();      // the next tile overwrites ab
    }
    if (dma_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // never leave DMA in flight into freed LDS
}

template <bool OT_LDS, int KCH>
__global__ void __launch_bounds__(SCORE_WAVES * 64)
score_tile_kernel(const ScoreParams p) {
    extern __shared__ __align__(16) float smem[];
    score_tiles<OT_LDS, KCH, false>(p, blockIdx.y, blockIdx.x, gridDim.x, smem, threadIdx.x);
}

inline size_t score_lds_bytes(int S, int SP, int Kc, int has_P, int ot_in_lds) {
    size_t bytes = (size_t)SCORE_TT * SP * 4;
    if (has_P) bytes += (size_t)SCORE_WAVES * Kc * 4;
    if (ot_in_lds) bytes += ((size_t)S * Kc * 4 + 1023) / 1024 * 1024;
    return bytes;
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

// History variant of the DP (used when the LDS holds it): the forward pass keeps only the partition
// VALUES of every step (2 adds + 1 max per (i, j) instead of 2 adds + compare + 2 selects, and the
// four lanes of a tag combine on the DPP network instead of two ds_bpermute round trips); the
// back-pointers the reference stores (crf.py:147-149) are recomputed lazily along the ONE path the
// backtrace follows: bp_t[j] = first argmax_i ((f_t[j] + tr[i][j]) + part_{t-1}[i]) is the same f32
// expression on the same values, so the path is bit-identical.  The transposed transition table
// stays in LDS for that second pass.
// FUSED: the workgroup also computes the clamped scores of its sequence (what score_tile_kernel would
// have written to crf_scores) straight into LDS -- one kernel from stash to tags, no score round trip
// through HBM: alpha*beta products staged transposed in the (not yet used) history area, then a
// register-blocked [tokens x S].[S x K] product, 4 tokens x 4 tags per lane, against the L2-resident
// transposed output matrix.  Same fmaf chain in s order as score_tile_kernel: identical bits.
template <int IB4, bool FUSED>
__global__ void __launch_bounds__(1024)
viterbi_hist_kernel(const ScoreParams p) {
    constexpr int IB = IB4 * 4;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nthreads = blockDim.x;
    const int b = blockIdx.x;
    const int n = clamp_len(p.len[b], p.L);
    (void)p.full;
    const int K = p.K, Kp = p.Kp;
    const int PW = 4 * IB;                               // padded partition width (>= K)
    float *hist = smem;                                  // [L][PW] partitions of every step, pads -inf
    const int sc_pieces = (p.L * Kp * 4 + 1023) / 1024, tr_pieces = (K * Kp * 4 + 1023) / 1024;
    float *scl = hist + (size_t)p.L * PW;                // [L][Kp] clamped scores of this sequence (whole KiB)
    float *trl = scl + sc_pieces * 256;                  // [K][Kp] trT: trl[j][i] = transitions[i][j]
    const float *sc = p.crf_scores + (long long)b * p.L * Kp;
    const long long foff = p.offs ? p.offs[b] : (p.flat ? flat_offset_in_kernel(p.len, b, p.L, tid, nthreads) : 0);
    const int START = K - 2, STOP = K - 1;
    const float ninf = -INFINITY;
    const int wu = __builtin_amdgcn_readfirstlane(w), nwaves = nthreads >> 6;

    // set-up without a register round trip: the scores and (behind them) the transition table stream
    // into LDS by LDS-DMA; the table is only needed by the backtrace, so its pieces stay in flight
    // during the forward pass (counted vmcnt: this wavefront's table pieces are its youngest operations)
    if (!FUSED) {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)scl);
        const int need = (n * Kp * 4 + 1023) / 1024;
        for (int k = wu; k < need; k += nwaves)
            lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, reinterpret_cast<const char *>(sc), lds0 + (unsigned)k * 1024u);
    } else {
        const int S = p.S, SP = p.SP, SP4 = SP >> 2, Lq = (p.L + 3) & ~3;
        // the transposed output matrix borrows the transition table's LDS area until the scores are done
        const int ot_pieces = (S * p.Kc * 4 + 1023) / 1024;
        const bool ot_lds = ot_pieces <= tr_pieces;
        if (ot_lds) {
            const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)trl);
            for (int k = wu; k < ot_pieces; k += nwaves)
                lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, reinterpret_cast<const char *>(p.OT), lds0 + (unsigned)k * 1024u);
        }
        float *abT = hist;                               // [SP][Lq] alpha*beta, token-contiguous (aliases hist)
        const float *Ab = p.A + (long long)b * (p.L + 1) * SP;
        const float *Bb = p.Bk + (long long)b * (p.L + 1) * SP;
        for (int idx = tid; idx < Lq * SP4; idx += nthreads) {
            const int tok = idx / SP4, s4 = (idx - tok * SP4) * 4;
            float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a4;
            if (tok < n) {       // alpha = state after tok+1 tokens; beta = backward state before token tok+1 (:415-420)
                a4 = ld4(Ab + (long long)(tok + 1) * SP + s4);
                b4 = ld4(Bb + (long long)(n - (tok + 1)) * SP + s4);
            }
            abT[(s4 + 0) * Lq + tok] = a4.x * b4.x; abT[(s4 + 1) * Lq + tok] = a4.y * b4.y;
            abT[(s4 + 2) * Lq + tok] = a4.z * b4.z; abT[(s4 + 3) * Lq + tok] = a4.w * b4.w;
        }
        __syncthreads();                                 // (drains vmcnt too: the output matrix has landed)
        const int ncg = (K + 3) >> 2, ntg = (n + 3) >> 2;
        const int cg = tid % ncg, tg0 = tid / ncg, tgs = nthreads / ncg;
        const int clamp_col = K - 3;                     // model_decompose.py:353
        auto score_tiles = [&](auto op0, int ostride4) {   // op0: this lane's 4 tags in row 0 of the output matrix
            for (int tg = tg0; tg < ntg && tgs > 0; tg += tgs) {
                float acc[4][4];
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[u][c] = 0.0f;
                lds_cv4f *ap = (lds_cv4f *)((lds_cfloat *)abT + tg * 4);
#pragma unroll 4
                for (int s0 = 0; s0 < S; s0++) {
                    const v4f a4 = ap[s0 * (Lq >> 2)];
                    const v4f o4 = op0[(long long)s0 * ostride4];
                    const float av[4] = {a4.x, a4.y, a4.z, a4.w}, ov[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
                    for (int u = 0; u < 4; u++)
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[u][c] = fmaf(av[u], ov[c], acc[u][c]);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int tok = tg * 4 + u;
                    if (tok >= n) continue;
                    float v[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        v[c] = acc[u][c] + 0.0f;                      // -0.0 -> +0.0 like score_tile_kernel
                        if (cg * 4 + c == clamp_col) v[c] = fminf(v[c], p.threshold);
                    }
                    st4(scl + (long long)tok * Kp + cg * 4, make_float4(v[0], v[1], v[2], v[3]));
                }
            }
        };
        if (ot_lds) score_tiles((lds_cv4f *)((lds_cfloat *)trl + cg * 4), p.Kc >> 2);
        else score_tiles((glb_cv4f *)(p.OT + cg * 4), p.Kc >> 2);
        __syncthreads();                                 // abT (aliasing hist) is free again
    }
    const int j = tid >> 2, q = tid & 3;
    const bool owner = j < K;
    float trr[IB];                                       // tr[i][j] for this lane's block of i
    {
        const float *row = p.trT + (long long)(owner ? j : 0) * Kp + q * IB;   // 16-byte aligned; may run into
#pragma unroll                                                                  // the next row: masked below
        for (int k4 = 0; k4 < IB4; k4++) {
            const float4 v = ld4(row + k4 * 4);
            trr[k4 * 4 + 0] = v.x; trr[k4 * 4 + 1] = v.y; trr[k4 * 4 + 2] = v.z; trr[k4 * 4 + 3] = v.w;
        }
    }
    const float t_start = (owner && q == 0) ? p.trT[(long long)j * Kp + START] : 0.0f;   // before the table DMA
    int my_tr = 0;
    {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)trl);
        for (int k = wu; k < tr_pieces; k += nwaves, my_tr++)
            lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, reinterpret_cast<const char *>(p.trT), lds0 + (unsigned)k * 1024u);
    }
    if (PW > K)
        for (int i = tid; i < n * (PW - K); i += nthreads) hist[(i / (PW - K)) * PW + K + i % (PW - K)] = ninf;
    wait_vmcnt(my_tr);                                   // scores + this lane's transition block landed
#pragma unroll
    for (int k = 0; k < IB; k++) trr[k] = (owner && q * IB + k < K) ? trr[k] : ninf;
    wg_barrier_lds();                                    // (a __syncthreads would drain the table DMA)
    if (owner && q == 0) hist[j] = scl[j] + t_start;                          // crf.py:135
    wg_barrier_lds();
    for (int t = 1; t < n; t++) {
        const float *pin = hist + (size_t)(t - 1) * PW + q * IB;
        const float f = owner ? scl[(long long)t * Kp + j] : 0.0f;
        float best = ninf;
#pragma unroll
        for (int k4 = 0; k4 < IB4; k4++) {
            const float4 p4 = ld4(pin + k4 * 4);
            const float v0 = (f + trr[k4 * 4 + 0]) + p4.x, v1 = (f + trr[k4 * 4 + 1]) + p4.y;   // crf.py:123,145
            const float v2 = (f + trr[k4 * 4 + 2]) + p4.z, v3 = (f + trr[k4 * 4 + 3]) + p4.w;
            best = fmaxf(fmaxf(best, v0), v1);
            best = fmaxf(fmaxf(best, v2), v3);
        }
        best = fmaxf(best, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(best), 0xB1, 0xf, 0xf, false)));
        best = fmaxf(best, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(best), 0x4E, 0xf, 0xf, false)));
        if (owner && q == 0) hist[(size_t)t * PW + j] = best;
        wg_barrier_lds();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the transition table is in LDS
    __syncthreads();
    if (w == 0 && n > 0) {
        const float *pin = hist + (size_t)(n - 1) * PW;
        float bv = ninf; int bi = 0x7ffffffe;
        for (int i = lane; i < K; i += WAVE) {
            const float v = pin[i] + trl[(long long)STOP * Kp + i];            // crf.py:168-169
            if (v > bv) { bv = v; bi = i; }
        }
        int ptr = wave_argmax_dpp(bv, bi);
        if (ptr >= K) ptr = 0;
        for (int t = n - 1; t >= 0; t--) {
            if (lane == 0) {
                const int tag = (ptr == K - 3) ? p.o_idx : ptr;               // model_decompose.py:356
                if (p.tags) p.tags[(long long)b * p.L + t] = tag;
                if (p.flat) p.flat[foff + t] = tag;
            }
            if (t > 0) {      // the back-pointer of step t at tag ptr (crf.py:147-149), recomputed
                const float f = scl[(long long)t * Kp + ptr];
                const float *pp = hist + (size_t)(t - 1) * PW;
                const float *tr = trl + (long long)ptr * Kp;
                float v = ninf; int vi = 0x7ffffffe;
                for (int i = lane; i < K; i += WAVE) {
                    const float c = (f + tr[i]) + pp[i];
                    if (c > v) { v = c; vi = i; }
                }
                ptr = wave_argmax_dpp(v, vi);
                if (ptr >= K) ptr = 0;
            }
        }
    }
    if (p.tags)
        for (int i = n + tid; i < p.L; i += nthreads) p.tags[(long long)b * p.L + i] = -1;   // pads (LOCAL and FULL)
}

// the instantiated block size (in float4s) for K tags: ceil(ceil(K/4)/4) rounded up to a built one
inline int viterbi_ib4(int K) {
    const int need = ((K + 3) / 4 + 3) / 4;
    return need <= 2 ? 2 : need <= 4 ? 4 : need <= 9 ? 9 : need <= 13 ? 13 : 16;
}
inline size_t viterbi_lds_bytes(int K, int Kp, int L) {
    return (size_t)2 * 16 * viterbi_ib4(K) * 4 + (size_t)L * Kp * 4 + (size_t)L * Kp * 2;
}
// can the fused variant stage [SP][L] products in the history area?
inline bool viterbi_fused_fits(int K, int SP, int L) {
    return (size_t)SP * ((L + 3) & ~3) <= (size_t)L * 16 * viterbi_ib4(K);
}
inline size_t viterbi_hist_lds_bytes(int K, int Kp, int L) {
    return (size_t)L * 16 * viterbi_ib4(K) * 4 + ((size_t)L * Kp * 4 + 1023) / 1024 * 1024 +
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
