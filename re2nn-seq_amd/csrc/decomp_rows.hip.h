// K12 "rows" kernel -- the decomposed recurrence (FARNN_S_D_W_I_S.get_forward_score,
// model_decompose_single.py:138-200; FARNN_S_D_W.get_forward_score, model_decompose.py:243-307) for
// the sum semiring, any gate mode (farnn 0/1/2) and any update non-linearity.
//
// Per step and direction the state-dependent work is three dense products against weights that are
// the SAME for every sequence (fwd shown; bwd swaps S1/S2 and transposes W):
//
//   P1 (farnn==2)  [z_pre ; r_pre] = [Wss1^T ; Wss2^T] . h                    2S rows x S
//                  hb = (1-r) * h_init + r * h
//   P2             rr = v_t * (S1^T . hb)      (+ z_pre = Wss1^T . h, farnn==1)  R (+S) rows x S
//   P3             nx = [S2 | W^T] . [rr ; hb]                                   S rows x (R+S)
//                  h' = farnn ? (1-z) h + z nl(nx) : nl(nx)
//
// The token-dependent halves of the gates, v_t . Wrs + bs, do not depend on the state: they are
// folded into per-word tables Gz/Gr[V][S] when the handle is built (weights are frozen on the
// tagging path), like Vgen.  The o scaling (:156-157 input side backward, :180-181 output side
// forward) is folded into the packed rows.
//
// Every product is a set of ROW dot products.  A workgroup of 8 wavefronts owns NSEQ sequences of one
// direction; four adjacent lanes share a row and split its columns in 32-column chunks, every 16-byte
// weight read is used for the NSEQ sequences, the math is packed f32 (v_pk_fma_f32) and the quad's
// partial sums meet on the DPP network (quad_perm), never in LDS; 128 rows per pass make a phase of
// S <= 128 rows one pass.  The packed rows live in LDS as far as the 160 KiB go (a rank-50 model fits
// whole: ~105 KiB); the rest is streamed from L2 every step (a gated rank-250 model: ~390 KiB of rows
// per step).  Element-wise work (gates, non-linearity, stash) runs on the few wavefronts that own
// state entries, between raw LDS barriers.  Bound: LDS-array cycles and VALU issue per step, not HBM:
// the weights are a few hundred KiB for the whole batch ("decomposed path: not HBM-bound").
//
// Measured history (B=256, L=64, S=104, MI355X):
//   * one wavefront per sequence, factors in LDS: 3.5 us/step at R=50 (LDS-latency bound, 2 waves/CU);
//     gated or rank >= 150 models fell back to a generic kernel at ~90 us/step.
//   * 16 waves, 16..64 lanes per row, epilogue inside the row loop: 3.3 us/step -- every wavefront
//     executed the tanh/gate/store epilogue for its 1-4 rows at 6% lane utilisation.
//   * the same with row sums parked in LDS and element-wise phases on 4 waves: 2.2 us/step.
//   * the products on v_mfma_f32_4x4x1_16b_f32 (scripts/probe/): parity-green but 3.0 us/step; the
//     16-block form issues 256 MACs per 32 cycles per SIMD, a quarter of the f32 VALU rate.
//   * this version: 2.2 us/step at NSEQ=2, 1.0 us/step per sequence at NSEQ=1 (0.66 us of it the
//     barrier/element-wise skeleton); whole farnn_tag 145 us at R=50 (was 237), 374 us for the gated
//     rank-250 model (was 5960).
#pragma once
#include "common.hip.h"
#include "decomp_chain.hip.h"
#include "chain.hip.h"      // select_by_length_rank

namespace farnn {

constexpr int DR_THREADS = 512;       // 8 wavefronts: 256 VGPRs each, cheap barriers
constexpr int DR_LPR = 4;             // lanes per row
constexpr int DR_RPP = DR_THREADS / DR_LPR;   // rows per pass (128)
constexpr int DR_CHUNK = 32;          // floats of a row one quad covers per chunk (2 x 16 bytes per lane)
constexpr int DR_MAX_PF = 4;          // prefetch registers per thread for the per-token vectors

struct DecompRowsParams {
    const float *P1;              // [2S][ld2]   gate rows (farnn==2) or nullptr
    const float *P2[2];           // [n2][ld2]   per direction
    const float *P3[2];           // [S][ld3]    per direction
    int n1, n2, n3, ld2, ld3, nch2, nch3;
    int res1, res2, res3;         // leading rows of each matrix kept in LDS
    const float *Vgen, *Gz, *Gr;  // [V][Rp], [V][SP], [V][SP]
    const float *h0, *hT;
    const int64_t *x, *len;
    const int *order;             // folded launch order (batch_prep) or nullptr
    int sort;                     // 1: no order array, the workgroup selects its sequences by length rank itself
    float *A, *Bk;
    int B, L, S, SP, R, Rp, farnn, nl, full, V;
    float sig_k;
    int dbg;                      // diagnostic ablation mask (FARNN_DBG); 0 in production
    int lds_floats;               // the launch's dynamic LDS size (a whole KiB), in floats
};

typedef float v2f __attribute__((ext_vector_type(2)));

// sum over the 4 lanes of a quad, result in all four (two DPP butterflies, no LDS)
__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm:[1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm:[2,3,0,1]
    return v;
}

// out[row] = <M[row, :], x_s> for s < NSEQ.  Four adjacent lanes share a row; per chunk of 32 columns a
// lane reads two 16-byte pieces of the row and of every x_s and does packed f32 FMAs (v_pk_fma_f32), so
// the instruction stream is mostly math; the quad's partial sums meet on the DPP network.  With 128
// rows per pass a phase of S <= 128 rows is ONE pass.  Rows [0, nres) come from LDS, the rest from
// global memory (L2): two loops over pointers of explicit address spaces -- a loop over one selected
// generic pointer compiles to flat_load + a full vmcnt/lgkmcnt drain per pass.
template <int NSEQ, typename Epi>
__device__ __forceinline__ void rowdots(const float *ml, const float *mg, int nres, int nrows, int ld, int nch,
                                        const float *X, int xs, int tid, Epi &&epi) {
    const int k = tid & (DR_LPR - 1), rloc = tid >> 2;
    lds_cfloat *xl = (lds_cfloat *)X + k * 4;
    auto body = [&](auto src, int row, bool ok) {
        v2f tl[NSEQ], th[NSEQ];
#pragma unroll
        for (int s = 0; s < NSEQ; s++) { tl[s] = v2f{0.f, 0.f}; th[s] = v2f{0.f, 0.f}; }
#pragma unroll 2
        for (int c = 0; c < nch; c++) {
            const v4f a0 = src[c * 8], a1 = src[c * 8 + 4];
#pragma unroll
            for (int s = 0; s < NSEQ; s++) {
                lds_cv4f *xp = (lds_cv4f *)(xl + s * xs + c * DR_CHUNK);
                const v4f x0 = xp[0], x1 = xp[4];
                tl[s] = __builtin_elementwise_fma(v2f{a0.x, a0.y}, v2f{x0.x, x0.y}, tl[s]);
                th[s] = __builtin_elementwise_fma(v2f{a0.z, a0.w}, v2f{x0.z, x0.w}, th[s]);
                tl[s] = __builtin_elementwise_fma(v2f{a1.x, a1.y}, v2f{x1.x, x1.y}, tl[s]);
                th[s] = __builtin_elementwise_fma(v2f{a1.z, a1.w}, v2f{x1.z, x1.w}, th[s]);
            }
        }
        float acc[NSEQ];
#pragma unroll
        for (int s = 0; s < NSEQ; s++) { const v2f t = tl[s] + th[s]; acc[s] = quad_sum(t.x + t.y); }
        if (k == 0 && ok) epi(row, acc);
    };
    int row0 = 0;
    for (; row0 < nres; row0 += DR_RPP) {
        const int row = row0 + rloc;
        const bool ok = row < nrows;
        body((lds_cv4f *)((lds_cfloat *)ml + (ok ? row : nrows - 1) * ld + k * 4), row, ok);
    }
    for (; row0 < nrows; row0 += DR_RPP) {
        const int row = row0 + rloc;
        const bool ok = row < nrows;
        body((glb_cv4f *)(mg + (long long)(ok ? row : nrows - 1) * ld + k * 4), row, ok);
    }
}

// A FEW rows (the rows behind a mixed form's register passes: 6 or 12 of them), all LDS-resident: a 128-row pass would spend a
// whole pass's reads and FMAs on them.  Here a wavefront takes a row, lane l the columns 4l .. 4l+3 (+ 256): two 16-byte
// reads of the row, eight FMAs, a wave sum; the input vector's pieces are read once for all of a wavefront's rows.
template <int NSEQ, typename Epi>
__device__ __forceinline__ void rowdots_few(const float *ml, int nrows, int ld, int nch, const float *X, int xs, int tid, Epi &&epi) {
    const int lane = tid & 63, wv = tid >> 6;
    const int ncol = nch * DR_CHUNK;                            // nch <= 16: two column blocks of 256
    for (int row = wv; row < nrows; row += DR_THREADS / 64) {
        lds_cfloat *rl = (lds_cfloat *)ml + row * ld;
        float acc[NSEQ];
#pragma unroll
        for (int s = 0; s < NSEQ; s++) acc[s] = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; i++) {                           // (the input vector's pieces are re-read per row: registers are what
            const int c = 4 * lane + 256 * i;                   //  a mixed form has least of)
            if (c < ncol) {
                const v4f a = *(lds_cv4f *)(rl + c);
#pragma unroll
                for (int s = 0; s < NSEQ; s++) {
                    const v4f x = *(lds_cv4f *)((lds_cfloat *)X + s * xs + c);
                    acc[s] += (a.x * x.x + a.y * x.y) + (a.z * x.z + a.w * x.w);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NSEQ; s++) acc[s] = wave_sum_dpp(acc[s]);
        if (lane == 0) epi(row, acc);
    }
}

// The same row dot products with the rows held in REGISTERS for the whole sequence: pass i covers the rows i*128 + tid/4,
// w[i][2c], w[i][2c+1] are this lane's two 16-byte pieces of chunk c.  The chunk of every x_s is read from LDS once and used
// by all NP passes.
// NP and NCH are upper bounds: passes past the last row are not written; chunks past the model's hold zero weights, and their
// x reads run on into the LDS vectors behind x (state / gate / row floats, or past the allocation: zeros) -- finite times
// zero.  (Clamping the chunk index instead turned the reads' immediate offsets into address arithmetic: +7 % per step.)
template <int NSEQ, int NP, int NCH, typename Epi>
__device__ __forceinline__ void rowdots_regs(const v4f (&w)[NP][2 * NCH], int nrows, int /*nch*/, const float *X, int xs, int tid, Epi &&epi) {
    const int k = tid & (DR_LPR - 1), rloc = tid >> 2;
    lds_cfloat *xl = (lds_cfloat *)X + k * 4;
    v2f tl[NP][NSEQ], th[NP][NSEQ];
#pragma unroll
    for (int i = 0; i < NP; i++)
#pragma unroll
        for (int s = 0; s < NSEQ; s++) { tl[i][s] = v2f{0.f, 0.f}; th[i][s] = v2f{0.f, 0.f}; }
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int s = 0; s < NSEQ; s++) {
            lds_cv4f *xp = (lds_cv4f *)(xl + s * xs + c * DR_CHUNK);   // compile-time offsets: the reads stay `ds_read ... offset:`
            const v4f x0 = xp[0], x1 = xp[4];
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const v4f a0 = w[i][2 * c], a1 = w[i][2 * c + 1];
                tl[i][s] = __builtin_elementwise_fma(v2f{a0.x, a0.y}, v2f{x0.x, x0.y}, tl[i][s]);
                th[i][s] = __builtin_elementwise_fma(v2f{a0.z, a0.w}, v2f{x0.z, x0.w}, th[i][s]);
                tl[i][s] = __builtin_elementwise_fma(v2f{a1.x, a1.y}, v2f{x1.x, x1.y}, tl[i][s]);
                th[i][s] = __builtin_elementwise_fma(v2f{a1.z, a1.w}, v2f{x1.z, x1.w}, th[i][s]);
            }
        }
#pragma unroll
    for (int i = 0; i < NP; i++) {
        float acc[NSEQ];
#pragma unroll
        for (int s = 0; s < NSEQ; s++) { const v2f t = tl[i][s] + th[i][s]; acc[s] = quad_sum(t.x + t.y); }
        const int row = i * DR_RPP + rloc;
        if (k == 0 && row < nrows) epi(row, acc);
    }
}

// this lane's pieces of the rows of a packed matrix (rows past the last one: a copy of the last, never used)
template <int NP, int NCH>
__device__ __forceinline__ void load_rows_regs(v4f (&w)[NP][2 * NCH], const float *M, int nrows, int ld, int nch, int tid) {
    const int k = tid & (DR_LPR - 1), rloc = tid >> 2;
    const v4f zero = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int row = i * DR_RPP + rloc;
        glb_cv4f *src = (glb_cv4f *)(M + (long long)(row < nrows ? row : nrows - 1) * ld + k * 4);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            w[i][2 * c] = c < nch ? src[c * 8] : zero;
            w[i][2 * c + 1] = c < nch ? src[c * 8 + 4] : zero;
        }
    }
}

__device__ __forceinline__ float gate_sigmoid(float x, float k) { return 1.0f / (1.0f + __expf(-(x * k))); }

// update non-linearity on the hardware exponential: |error| ~1e-7 against the 1e-4 parity bar
__device__ __forceinline__ float dr_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));          // in (0, 1]: no overflow for any x
    return copysignf((1.0f - e) / (1.0f + e), x);
}
__device__ __forceinline__ float dr_nl(float x, int nl) {
    switch (nl) {
        case FARNN_NL_RELU: return fmaxf(x, 0.0f);
        case FARNN_NL_TANH: return dr_tanh(x);
        case FARNN_NL_RELUTANH: return dr_tanh(fmaxf(x, 0.0f));
        default: return x;
    }
}

// NP1R / NP2R / NP3R > 0: the matrix lives in REGISTERS (that many 128-row passes of NCH2R / NCH3R chunks) instead of LDS /
// L2: with the gates, or at rank 250, the packed rows exceed the 160 KiB of LDS and r02a streamed the rest from L2 on every
// step (the shipped example configuration's shape, S = 104, rank 250, farnn = 2: 286 KB per workgroup per step, 5.1 us per
// step).  Eight wavefronts have 512 KB of registers between them: P1 (gates) and P3 in registers + P2 in LDS hold that
// model whole, and a step touches neither L2 nor HBM for weights.
template <int NSEQ, int NP1R = 0, int NP2R = 0, int NCH2R = 0, int NP3R = 0, int NCH3R = 0, bool MIXED = false>
__global__ void __launch_bounds__(DR_THREADS)
decomp_rows_kernel(const DecompRowsParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x;
    const int dir = blockIdx.x & 1, grp = blockIdx.x >> 1;
    const int S = p.S, SP = p.SP, R = p.R, Rp = p.Rp, ld2 = p.ld2, ld3 = p.ld3;
    const int c2p = p.nch2 * DR_CHUNK, c3p = p.nch3 * DR_CHUNK;      // floats of the input vectors
    const int farnn = p.farnn;
    const int Lr = (p.L + 3) & ~3;
    const int tvl = Rp + (farnn >= 1 ? SP : 0) + (farnn == 2 ? SP : 0);

    // ---- LDS carve (every block a multiple of 16 bytes) ----------------------------------------
    int *tok = reinterpret_cast<int *>(smem);                 // [NSEQ][Lr]
    float *Hinit = smem + NSEQ * Lr;                          // [SP]
    float *H = Hinit + SP;                                    // [NSEQ][c2p]   state h (input of P1)
    float *HB = H + NSEQ * c2p;                               // [2][NSEQ][c2p]  hb: input of P2 (ping-pong)
    float *X3 = HB + 2 * NSEQ * c2p;                          // [2][NSEQ][c3p]  rr | hb: input of P3
    float *Z = X3 + 2 * NSEQ * c3p;                           // [NSEQ][SP]    update gate
    float *TV = Z + NSEQ * SP;                                // [2][NSEQ][tvl]  per-token vectors
    float *L1 = TV + 2 * NSEQ * tvl;                          // resident rows
    float *L2 = L1 + (long long)p.res1 * ld2;
    float *L3 = L2 + (long long)p.res2 * ld2;

    // ---- sequences of this workgroup -------------------------------------------------------------
    int bseq[NSEQ], nst[NSEQ], slen[NSEQ];
    int nmax = 0;
#pragma unroll
    for (int s = 0; s < NSEQ; s++) {
        const int r = grp * NSEQ + s;                         // rank by length (descending)
        const bool have = r < p.B;
        int b = 0;
        if (have) {
            const int half = p.B / 2;
            b = p.order ? p.order[r < half ? r : half + (p.B - 1 - r)] : r;     // undo the fold
        }
        if (p.sort && have) b = select_by_length_rank(p.len, p.B, p.L, r, reinterpret_cast<int *>(smem), tid, DR_THREADS);
        b = __builtin_amdgcn_readfirstlane(b);                // workgroup-uniform: keep it in SGPRs
        bseq[s] = b;
        slen[s] = have ? __builtin_amdgcn_readfirstlane(clamp_len(p.len[b], p.L)) : 0;
        nst[s] = have ? (p.full ? p.L : slen[s]) : -1;        // -1: no sequence in this slot
        nmax = nst[s] > nmax ? nst[s] : nmax;
    }
    const float *hinit = dir == 0 ? p.h0 : p.hT;
    float *stash_base = dir == 0 ? p.A : p.Bk;
    auto pick = [&](const int (&arr)[NSEQ], int s) { int v = arr[0];
#pragma unroll
        for (int q = 1; q < NSEQ; q++) v = s == q ? arr[q] : v;
        return v; };

    // ---- set-up ------------------------------------------------------------------------------------
    // everything a vector read can touch is initialised: the register forms read whole (upper-bound) chunk counts and run on
    // into the arrays behind the vector -- with zero weights, but LDS keeps what earlier workgroups left there (-inf pads of the
    // Viterbi kernel: -inf x 0 = NaN).  Vectors and per-token buffers are zeroed here, the tail behind the resident rows below.
    for (int i = tid; i < (int)(L1 - Hinit); i += DR_THREADS) Hinit[i] = 0.0f;
#pragma unroll
    for (int s = 0; s < NSEQ; s++)
        for (int k = tid; k < nst[s]; k += DR_THREADS) {
            const int idx = (dir == 0) ? k : (k < slen[s] ? slen[s] - 1 - k : k);
            tok[s * Lr + k] = clamp_tok(p.x[(long long)bseq[s] * p.L + idx], p.V);
        }
    // A register form may hold only a matrix's first NPxR passes (mixed forms: S in 129..160 needs three passes of gate rows or
    // two of output rows; the registers hold the first passes, the few rows behind them are LDS-resident like any other
    // matrix's, or streamed): R1 / R2 / R3 = rows held in registers, the resident / streamed rows start there.
    constexpr int R1 = NP1R * DR_RPP, R2 = NP2R * DR_RPP, R3 = NP3R * DR_RPP;
    {   // resident rows: global -> LDS
        const float *src[3] = {p.P1 + (long long)R1 * ld2, p.P2[dir] + (long long)R2 * ld2, p.P3[dir] + (long long)R3 * ld3};
        float *dst[3] = {L1, L2, L3};
        const long long cnt[3] = {(long long)p.res1 * ld2, (long long)p.res2 * ld2, (long long)p.res3 * ld3};
#pragma unroll
        for (int q = 0; q < 3; q++)
            for (long long i = (long long)tid * 4; i < cnt[q]; i += DR_THREADS * 4) st4(dst[q] + i, ld4(src[q] + i));
        float *tail = L3 + cnt[2];
        for (int i = tid; tail + i < smem + p.lds_floats; i += DR_THREADS) tail[i] = 0.0f;
    }
    v4f w1[NP1R > 0 ? NP1R : 1][2 * (NCH2R > 0 ? NCH2R : 1)], w2[NP2R > 0 ? NP2R : 1][2 * (NCH2R > 0 ? NCH2R : 1)];
    v4f w3[NP3R > 0 ? NP3R : 1][2 * (NCH3R > 0 ? NCH3R : 1)];
    if constexpr (NP1R > 0) load_rows_regs<NP1R, NCH2R>(w1, p.P1, p.n1, ld2, p.nch2, tid);
    if constexpr (NP2R > 0) load_rows_regs<NP2R, NCH2R>(w2, p.P2[dir], p.n2, ld2, p.nch2, tid);
    if constexpr (NP3R > 0) load_rows_regs<NP3R, NCH3R>(w3, p.P3[dir], p.n3, ld3, p.nch3, tid);
    __syncthreads();
    for (int j = tid; j < S; j += DR_THREADS) {
        const float hv = hinit[j];
        Hinit[j] = hv;
#pragma unroll
        for (int s = 0; s < NSEQ; s++) {
            if (nst[s] < 0) continue;
            H[s * c2p + j] = hv;
            HB[s * c2p + j] = hv;                              // buffer 0 = step 0's "cur"
            X3[s * c3p + Rp + j] = hv;
            stash_base[(long long)bseq[s] * (p.L + 1) * SP + j] = hv;
        }
    }
    // pad columns of every row the chains will write (the workspace is strided with the call's L and not re-zeroed)
    if (SP > S) {
#pragma unroll
        for (int s = 0; s < NSEQ; s++)
            for (int q = tid; q < (nst[s] + 1) * (SP - S); q += DR_THREADS)
                stash_base[((long long)bseq[s] * (p.L + 1) + q / (SP - S)) * SP + S + q % (SP - S)] = 0.0f;
    }

    // per-token vectors: element e of sequence s at step t
    auto tv_load = [&](int s, int e, int t) -> float {
        const int n = pick(nst, s);
        if (n <= 0) return 0.0f;
        const int tk = tok[s * Lr + (t < n ? t : n - 1)];
        if (e < Rp) return p.Vgen[(long long)tk * Rp + e];
        if (e < Rp + SP) return p.Gz[(long long)tk * SP + (e - Rp)];
        return p.Gr[(long long)tk * SP + (e - Rp - SP)];
    };
    const int ntv = NSEQ * tvl;
    int pf_s[DR_MAX_PF], pf_e[DR_MAX_PF];                     // loop-invariant split of the prefetch slots
#pragma unroll
    for (int i = 0; i < DR_MAX_PF; i++) {
        const int e = tid + i * DR_THREADS;
        pf_s[i] = e < ntv ? e / tvl : 0;
        pf_e[i] = e < ntv ? e % tvl : -1;
    }
    for (int i = 0; i < DR_MAX_PF; i++)
        if (pf_e[i] >= 0) TV[tid + i * DR_THREADS] = tv_load(pf_s[i], pf_e[i], 0);
    __syncthreads();

    const float sig_k = p.sig_k;
    const int nl_mode = p.nl;
    // element-wise phases: element e -> (sequence e % NSEQ, state entry e / NSEQ)
    for (int t = 0; t < nmax; t++) {
        const int cur = t & 1, nxt = cur ^ 1;
        const float *TVc = TV + cur * ntv;
        int act[NSEQ];
#pragma unroll
        for (int s = 0; s < NSEQ; s++) act[s] = t < nst[s];
        // next step's per-token vectors: parked in registers until after P2.  Their loads (a token id from LDS, an address, a
        // global load each) are issued behind P1's products when there is a P1 phase -- at the step's top they sat between the
        // barrier and the step's first LDS reads (the placement that paid 5 % in decomp_regs8_kernel)
        float pf[DR_MAX_PF];
#define FARNN_DR_ISSUE_PREFETCH()                                                                                  \
        _Pragma("unroll")                                                                                          \
        for (int i = 0; i < DR_MAX_PF; i++)                                                                        \
            pf[i] = (pf_e[i] >= 0 && t + 1 < nmax && !(p.dbg & 8)) ? tv_load(pf_s[i], pf_e[i], t + 1) : 0.0f;
        if (farnn != 2) { FARNN_DR_ISSUE_PREFETCH() }
        // The element-wise work rides in the row epilogues: the lane that finishes a row sum turns it into
        // a gate, an rr entry or the new state right away, so a step is 2 barriers (3 with farnn==2).
        // hb and [rr | hb] ping-pong because P3's epilogue writes the next step's hb while other
        // wavefronts still read this step's.
        float *HBc = HB + cur * NSEQ * c2p, *HBn = HB + nxt * NSEQ * c2p;
        float *X3c = X3 + cur * NSEQ * c3p, *X3n = X3 + nxt * NSEQ * c3p;
        if (farnn == 2) {
            // ---- P1: z, r from h; hb = (1-r) h_init + r h  (:143-151) -----------------------------------
            auto epi1 = [&](int row, const float (&acc)[NSEQ]) {
#pragma unroll
                for (int s = 0; s < NSEQ; s++) {
                    if (!act[s]) continue;
                    const float *tv = TVc + s * tvl;
                    if (row < S) {
                        Z[s * SP + row] = gate_sigmoid(acc[s] + tv[Rp + row], sig_k);
                    } else {
                        const int j = row - S;
                        const float rg = gate_sigmoid(acc[s] + tv[Rp + SP + j], sig_k);
                        const float hb = (1.0f - rg) * Hinit[j] + rg * H[s * c2p + j];
                        HBc[s * c2p + j] = hb;
                        X3c[s * c3p + Rp + j] = hb;
                    }
                }
            };
            if (!(p.dbg & 1)) {
                if constexpr (NP1R > 0) {
                    rowdots_regs<NSEQ, NP1R, NCH2R>(w1, p.n1, p.nch2, H, c2p, tid, epi1);
                    if (MIXED && p.n1 > R1) {                   // (mixed form: the rows behind the register passes)
                        auto shifted = [&](int row, const float (&acc)[NSEQ]) { epi1(row + R1, acc); };
                        rowdots_few<NSEQ>(L1, p.n1 - R1, ld2, p.nch2, H, c2p, tid, shifted);   // (the plan keeps them LDS-resident)
                    }
                } else rowdots<NSEQ>(L1, p.P1, p.res1, p.n1, ld2, p.nch2, H, c2p, tid, epi1);
            }
            FARNN_DR_ISSUE_PREFETCH()
            wg_barrier_lds();
        }
#undef FARNN_DR_ISSUE_PREFETCH
        {   // ---- P2: rr = v * (Sa^T . hb)  (:169-170 / :174-175); farnn==1: z from the same h ------------
            auto epi2 = [&](int row, const float (&acc)[NSEQ]) {
#pragma unroll
                for (int s = 0; s < NSEQ; s++) {
                    if (row < R) X3c[s * c3p + row] = acc[s] * TVc[s * tvl + row];
                    else Z[s * SP + (row - R)] = gate_sigmoid(acc[s] + TVc[s * tvl + Rp + (row - R)], sig_k);
                }
            };
            if (!(p.dbg & 1)) {
                if constexpr (NP2R > 0) {
                    rowdots_regs<NSEQ, NP2R, NCH2R>(w2, p.n2, p.nch2, HBc, c2p, tid, epi2);
                    if (MIXED && p.n2 > R2) {                   // (mixed form: the rows behind the register passes)
                        auto shifted = [&](int row, const float (&acc)[NSEQ]) { epi2(row + R2, acc); };
                        rowdots_few<NSEQ>(L2, p.n2 - R2, ld2, p.nch2, HBc, c2p, tid, shifted);   // (the plan keeps them LDS-resident)
                    }
                } else rowdots<NSEQ>(L2, p.P2[dir], p.res2, p.n2, ld2, p.nch2, HBc, c2p, tid, epi2);
            }
        }
        {   // park the prefetched per-token vectors BEFORE this step's stash stores are issued: vmcnt retires
            // in order, so waiting for these loads later would also wait for every younger store
            float *TVn = TV + nxt * ntv;
#pragma unroll
            for (int i = 0; i < DR_MAX_PF; i++)
                if (pf_e[i] >= 0) TVn[tid + i * DR_THREADS] = pf[i];
        }
        wg_barrier_lds();
        {   // ---- P3: nx = Sb . rr + W(^T) . hb, non-linearity, gate mix, stash  (:171-196) -----------------
            auto epi3 = [&](int row, const float (&acc)[NSEQ]) {
#pragma unroll
                for (int s = 0; s < NSEQ; s++) {
                    if (!act[s]) continue;
                    const float nx = (p.dbg & 2) ? acc[s] : dr_nl(acc[s], nl_mode);
                    float hn = nx;
                    if (farnn == 2) {
                        const float z = Z[s * SP + row];
                        hn = (1.0f - z) * H[s * c2p + row] + z * nx;
                        H[s * c2p + row] = hn;
                    } else {
                        if (farnn == 1) {
                            const float z = Z[s * SP + row];
                            hn = (1.0f - z) * HBc[s * c2p + row] + z * nx;
                        }
                        HBn[s * c2p + row] = hn;
                        X3n[s * c3p + Rp + row] = hn;
                    }
                    if (!(p.dbg & 4)) stash_base[((long long)bseq[s] * (p.L + 1) + t + 1) * SP + row] = hn;
                }
            };
            if (!(p.dbg & 1)) {
                if constexpr (NP3R > 0) {
                    rowdots_regs<NSEQ, NP3R, NCH3R>(w3, p.n3, p.nch3, X3c, c3p, tid, epi3);
                    if (MIXED && p.n3 > R3) {                   // (mixed form: the rows behind the register passes)
                        auto shifted = [&](int row, const float (&acc)[NSEQ]) { epi3(row + R3, acc); };
                        rowdots_few<NSEQ>(L3, p.n3 - R3, ld3, p.nch3, X3c, c3p, tid, shifted);   // (the plan keeps them LDS-resident)
                    }
                } else rowdots<NSEQ>(L3, p.P3[dir], p.res3, p.n3, ld3, p.nch3, X3c, c3p, tid, epi3);
            }
        }
        wg_barrier_lds();
    }
}

// ---- packing (create time) ------------------------------------------------------------------------
// P2[dir] row r < R: Sa[:, r] (fwd Sa = S1, bwd Sa = S2 with the backward input scaling o folded in);
//          row R + j (farnn==1): Wss1[:, j].     P1 row j: Wss1[:, j]; row S + j: Wss2[:, j].
// P3[dir] row j: [Sb[j, :] | pad to Rp | Wd[:, j]] (fwd Sb = S2, Wd[s] = W[s][j], row scaled by o[j];
//          bwd Sb = S1, Wd[s] = W[j][s] * o[s]).
struct PackSrc {
    const float *S1, *S2, *W;     // [S][Rp], [S][Rp], [S][SP]
    const float *Wss1, *Wss2;     // [S][SP] or nullptr
    const float *o;               // [SP]
    int S, SP, R, Rp, farnn;
};

__global__ void pack_p2_kernel(PackSrc q, float *out, int nrows, int ld, int dir) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nrows * ld) return;
    const int row = (int)(idx / ld), s = (int)(idx % ld);
    float v = 0.0f;
    if (s < q.S) {
        if (row < q.R) v = dir == 0 ? q.S1[(long long)s * q.Rp + row] : q.S2[(long long)s * q.Rp + row] * q.o[s];
        else v = q.Wss1[(long long)s * q.SP + (row - q.R)];
    }
    out[idx] = v;
}

__global__ void pack_p1_kernel(PackSrc q, float *out, int ld) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2LL * q.S * ld) return;
    const int row = (int)(idx / ld), s = (int)(idx % ld);
    float v = 0.0f;
    if (s < q.S) v = row < q.S ? q.Wss1[(long long)s * q.SP + row] : q.Wss2[(long long)s * q.SP + (row - q.S)];
    out[idx] = v;
}

__global__ void pack_p3_kernel(PackSrc q, float *out, int ld, int dir) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)q.S * ld) return;
    const int j = (int)(idx / ld), c = (int)(idx % ld);
    float v = 0.0f;
    if (c < q.R) {
        v = dir == 0 ? q.S2[(long long)j * q.Rp + c] * q.o[j] : q.S1[(long long)j * q.Rp + c];
    } else if (c >= q.Rp && c < q.Rp + q.S) {
        const int s = c - q.Rp;
        v = dir == 0 ? q.W[(long long)s * q.SP + j] * q.o[j] : q.W[(long long)j * q.SP + s] * q.o[s];
    }
    out[idx] = v;
}

// G[v][j] = sum_r Vgen[v][r] Wrs[r][j] + bs[j]   (the token half of a gate, :144-148)
__global__ void gate_table_kernel(const float *Vgen, const float *Wrs, const float *bs, float *G,
                                  int V, int R, int Rp, int S, int SP) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)V * SP) return;
    const int v = (int)(idx / SP), j = (int)(idx % SP);
    float acc = 0.0f;
    if (j < S) {
        for (int r = 0; r < R; r++) acc = fmaf(Vgen[(long long)v * Rp + r], Wrs[(long long)r * SP + j], acc);
        acc += bs[j];
    }
    G[idx] = acc;
}

// everything the launcher needs, filled by build_rows_pack() in farnn_hip.hip
struct DecompRowsPack {
    bool ok = false;
    float *P1 = nullptr, *P2[2] = {nullptr, nullptr}, *P3[2] = {nullptr, nullptr}, *Gz = nullptr, *Gr = nullptr;
    int n1 = 0, n2 = 0, n3 = 0, ld2 = 0, ld3 = 0, nch2 = 0, nch3 = 0;
};

// row length in floats for `cols` columns: whole 32-column chunks, +16 so that the four rows a 16-lane
// LDS access group touches (64 contiguous bytes each) start in different quarters of the 64 banks
inline int rows_ld(int cols) { return (cols + DR_CHUNK - 1) / DR_CHUNK * DR_CHUNK + 16; }

struct RowsPlan { int nseq, res1, res2, res3; size_t lds; int form; };   // form: which matrices live in registers (0: none)

// register-resident forms that are instantiated (one sequence per workgroup): UPPER BOUNDS on the passes of P1 / P2, the
// chunks of their rows and the passes / chunks of P3 (in order of cost; the first that holds the model is taken).
// 1: farnn = 2, S <= 128, Rp + SP <= 224 (rank ~100-120); 2: farnn = 1 with R + S <= 256; 3: farnn = 2, S <= 128,
// Rp + SP <= 384 (rank 150 / 250: the shipped example configurations).
// 4: farnn = 2 at 128 < S <= 160 (the shipped configurations with --additional_states 30 at this automaton size): the gate
// rows alone in registers, P2 / P3 in LDS as far as it goes, the rest streamed.
// 5 / 6 (round 3, MIXED: the registers hold a matrix's first passes, the rows behind them are LDS-resident): farnn = 2 at
// 128 < S <= 160 with the output rows in registers too: two passes of gate rows + the 12 behind them in LDS, one pass of output
// rows + the 6 behind it in LDS, P2 in LDS -- 5: Rp + SP <= 288 (rank 150 + additional states: P2 fits whole, nothing is
// streamed); 6: Rp + SP <= 416 (rank 250 + additional states: P2 as far as the LDS goes, 128 of its 250 rows).  (Three passes
// of gate rows beside a pass of output rows do not fit the 256 VGPRs of a lane: 46 spilled.)
#define FARNN_ROWS_FORMS(X) X(1, 2, 0, 4, 1, 7) X(2, 0, 2, 4, 0, 0) X(3, 2, 0, 4, 1, 12) X(5, 2, 0, 5, 1, 9) X(6, 2, 0, 5, 1, 13) X(4, 3, 0, 5, 0, 0)
constexpr bool rows_form_mixed(int form) { return form == 5 || form == 6; }

// one attempt at `nseq` sequences per workgroup; forms: may a register-resident form be chosen
inline bool rows_plan_try(const DecompRowsPack &k, const DecompWeights &w, int L, int nseq, bool forms, RowsPlan &pl) {
    const int Lr = (L + 3) & ~3;
    const int tvl = w.Rp + (w.farnn >= 1 ? w.SP : 0) + (w.farnn == 2 ? w.SP : 0);
    if (nseq * tvl > DR_MAX_PF * DR_THREADS) return false;
    const size_t fixed = 4 * ((size_t)nseq * Lr + w.SP + 3ull * nseq * k.nch2 * DR_CHUNK +
                              2ull * nseq * k.nch3 * DR_CHUNK + (size_t)nseq * w.SP + 2ull * nseq * tvl);
    const size_t cap = 159 * 1024;
    if (fixed + 16 * 1024 > cap) return false;               // leave room for at least some resident rows
    size_t left = cap - fixed;
    auto take = [&](int nrows, int ld) {
        if (nrows == 0) return 0;
        long long fit = (long long)(left / ((size_t)ld * 4));
        int res = fit >= nrows ? nrows : (int)(fit / DR_RPP) * DR_RPP;
        left -= (size_t)res * ld * 4;
        return res;
    };
    pl.form = 0;
    if (forms && nseq == 1) {      // (two sequences per workgroup on these forms: measured slower, spills)
        const int np1 = (k.n1 + DR_RPP - 1) / DR_RPP, np2 = (k.n2 + DR_RPP - 1) / DR_RPP, np3 = (k.n3 + DR_RPP - 1) / DR_RPP;
#define FARNN_ROWS_MATCH(F_, A_, B_, C_, D_, E_)                                                                   \
        if (!pl.form && (A_ == 0 || (k.n1 > 0 && (np1 <= A_ || rows_form_mixed(F_)) && k.nch2 <= C_)) &&              \
            (B_ == 0 || (np2 <= B_ && k.nch2 <= C_)) &&                                                           \
            (D_ == 0 || ((np3 <= D_ || rows_form_mixed(F_)) && k.nch3 <= E_)) && (A_ > 0 || k.n1 == 0) &&         \
            (!rows_form_mixed(F_) || (np1 > 2 && np3 > 1))) {     /* (mixed forms: only where forms 1-3 do not reach) */ \
            /* the matrices left outside the registers go to the LDS as far as it holds them (whole, for forms 1-3 at   \
               the sizes they were made for), the remainder is streamed as before; a mixed form's rows behind its     \
               register passes first: they are few */                                                             \
            pl.form = F_;                                                                                          \
            const size_t left0 = left;                                                                             \
            const int t3 = k.n3 > D_ * DR_RPP ? k.n3 - D_ * DR_RPP : 0, t1 = k.n1 > A_ * DR_RPP ? k.n1 - A_ * DR_RPP : 0; \
            pl.res3 = take(t3, k.ld3);                                                                             \
            pl.res1 = take(t1, k.ld2);                                                                             \
            if (rows_form_mixed(F_) && (pl.res3 < t3 || pl.res1 < t1 || k.nch2 > 16 || k.nch3 > 16)) {             \
                pl.form = 0; left = left0;              /* its rows behind the register passes must all be LDS-resident */ \
            } else pl.res2 = take(k.n2 > B_ * DR_RPP ? k.n2 - B_ * DR_RPP : 0, k.ld2);                             \
        }
        FARNN_ROWS_FORMS(FARNN_ROWS_MATCH)
#undef FARNN_ROWS_MATCH
    }
    if (!pl.form) {
        pl.res3 = take(k.n3, k.ld3);
        pl.res2 = take(k.n2, k.ld2);
        pl.res1 = take(k.n1, k.ld2);
    }
    pl.nseq = nseq;
    // Register forms read whole vector chunks, up to a chunk past the model's width: everything up to the end of the ALLOCATION
    // must be initialised.  The kernel zeroes its tail up to lds_floats; round that up to a multiple of both 1 KiB (LDS-DMA
    // pieces) and the LDS allocation granule (1280 B on gfx950: what the hardware hands out beyond the request), capped by
    // the 160 KiB of the CU, so that no allocated-but-unzeroed slack can sit behind it
    pl.lds = (cap - left + 5119) / 5120 * 5120;
    if (pl.lds > 160 * 1024) pl.lds = 160 * 1024;
    return true;
}

inline bool rows_plan(const DecompRowsPack &k, const DecompWeights &w, int B, int L, RowsPlan &pl) {
    const bool forms = !tun(TUN_ROWS_NOREGS);
    if (const int v = tun(TUN_ROWS_NSEQ)) {
        if (v == 1 || v == 2 || v == 4) {
            for (int n = v; n >= 1; n /= 2)
                if (rows_plan_try(k, w, L, n, forms, pl)) return true;
            return false;
        }
    }
    // a register-resident form (one sequence per workgroup, no weight leaves the CU after set-up) beats sharing streamed
    // weights between sequences for any batch that is not huge: 210 vs 354 us at B = 256 for the gated rank-250 shape
    if (forms && B <= 1024 && rows_plan_try(k, w, L, 1, true, pl) && pl.form) return true;
    // measured (B=256, S=104): one sequence per workgroup in two rounds beats two per workgroup in one
    // (145 vs 168 us at R=50, 374 vs 450 us gated R=250): the x reads and FMAs scale with NSEQ, only the
    // weight reads are shared
    int nseq = 1;
    while (nseq < 4 && 2 * ((B + nseq - 1) / nseq) > 512 * nseq) nseq *= 2;
    for (; nseq >= 1; nseq /= 2)
        if (rows_plan_try(k, w, L, nseq, forms, pl)) return true;
    return false;
}

template <int NSEQ, int A_ = 0, int B_ = 0, int C_ = 0, int D_ = 0, int E_ = 0, bool MIXED = false>
inline int launch_rows_n(const DecompRowsParams &p, int groups, size_t lds, hipStream_t s) {
    static int raised = -1;     // per process and instantiation; the attribute is per (device, function) but monotone in lds
    if ((int)lds > raised) {
        FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(decomp_rows_kernel<NSEQ, A_, B_, C_, D_, E_, MIXED>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = 160 * 1024;
    }
    decomp_rows_kernel<NSEQ, A_, B_, C_, D_, E_, MIXED><<<dim3(2 * groups), dim3(DR_THREADS), lds, s>>>(p);
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

inline int launch_decomp_rows(const DecompRowsPack &k, const DecompWeights &w, const RowsPlan &pl,
                              const int64_t *x, const int64_t *len, const int *order, int sort_in_kernel,
                              float *A, float *Bk, int B, int L, int full, hipStream_t s) {
    DecompRowsParams p;
    p.P1 = k.P1; p.P2[0] = k.P2[0]; p.P2[1] = k.P2[1]; p.P3[0] = k.P3[0]; p.P3[1] = k.P3[1];
    p.n1 = k.n1; p.n2 = k.n2; p.n3 = k.n3; p.ld2 = k.ld2; p.ld3 = k.ld3; p.nch2 = k.nch2; p.nch3 = k.nch3;
    p.res1 = pl.res1; p.res2 = pl.res2; p.res3 = pl.res3;
    p.Vgen = w.Vgen; p.Gz = k.Gz; p.Gr = k.Gr; p.h0 = w.h0; p.hT = w.hT;
    p.x = x; p.len = len; p.order = order; p.sort = sort_in_kernel; p.A = A; p.Bk = Bk;
    p.B = B; p.L = L; p.S = w.S; p.SP = w.SP; p.R = w.R; p.Rp = w.Rp; p.farnn = w.farnn; p.nl = w.nl;
    p.full = full; p.V = w.V; p.sig_k = w.sig_k;
    p.dbg = tun(TUN_DBG);
    p.lds_floats = (int)(pl.lds / 4);
    const int groups = (B + pl.nseq - 1) / pl.nseq;
#define FARNN_ROWS_LAUNCH(F_, A_, B_, C_, D_, E_) if (pl.form == F_) return launch_rows_n<1, A_, B_, C_, D_, E_, rows_form_mixed(F_)>(p, groups, pl.lds, s);
    FARNN_ROWS_FORMS(FARNN_ROWS_LAUNCH)
#undef FARNN_ROWS_LAUNCH
    if (pl.nseq == 4) return launch_rows_n<4>(p, groups, pl.lds, s);
    if (pl.nseq == 2) return launch_rows_n<2>(p, groups, pl.lds, s);
    return launch_rows_n<1>(p, groups, pl.lds, s);
}

}  // namespace farnn
