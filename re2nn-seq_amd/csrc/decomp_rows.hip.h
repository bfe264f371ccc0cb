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
// Every product is a set of ROW dot products.  A workgroup of 16 wavefronts owns NSEQ sequences of
// one direction; LPR adjacent lanes share a row and split its columns, so a lane keeps its column
// slice of the NSEQ input vectors in registers for the whole phase and every 16-byte weight read is
// used NSEQ times.  Partial sums meet on the DPP network (row_shr / row_bcast), never in LDS.  The
// packed rows live in LDS as far as the 160 KiB go (a rank-50 model fits whole: ~105 KiB); the rest
// is streamed from L2 every step (a rank-250 gated model: ~390 KiB of rows per step, shared by the
// NSEQ sequences).  Bound: LDS / L2->CU bandwidth per step, not HBM: the weights are a few hundred
// KiB for the whole batch ("decomposed path: not HBM-bound").
//
// Measured (profiles/): the previous one-wavefront-per-sequence kernel ran 2 wavefronts per CU and
// was LDS-latency bound at 3.5 us per step (R=50); gated or rank>=150 models fell back to the
// generic kernel.
#pragma once
#include "common.hip.h"
#include "decomp_chain.hip.h"

namespace farnn {

#ifndef FARNN_DR_THREADS
#define FARNN_DR_THREADS 1024
#endif
constexpr int DR_THREADS = FARNN_DR_THREADS;
constexpr int DR_MAX_PF = 2;          // prefetch registers per thread for the per-token vectors

struct DecompRowsParams {
    const float *P1;              // [2S][ld2]   gate rows (farnn==2) or nullptr
    const float *P2[2];           // [n2][ld2]   per direction
    const float *P3[2];           // [S][ld3]    per direction
    int n1, n2, n3, ld2, ld3, lpr2, m2, lpr3, m3;
    int res1, res2, res3;         // leading rows of each matrix kept in LDS
    int vbs;                      // floats per [rr | hb] vector
    const float *Vgen, *Gz, *Gr;  // [V][Rp], [V][SP], [V][SP]
    const float *h0, *hT;
    const int64_t *x, *len;
    const int *order;             // folded launch order (batch_prep) or nullptr
    float *A, *Bk;
    int B, L, S, SP, R, Rp, farnn, nl, full;
    float sig_k;
    int dbg;                      // diagnostic ablation mask (FARNN_DBG); 0 in production
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float x) {
    // lanes without a source (row start / rows outside ROW_MASK) add 0
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    v = dpp_add<0x111, 0xf>(v);                      // row_shr:1
    v = dpp_add<0x112, 0xf>(v);                      // row_shr:2
    v = dpp_add<0x114, 0xf>(v);                      // row_shr:4
    v = dpp_add<0x118, 0xf>(v);                      // row_shr:8
    if (LPR >= 32) v = dpp_add<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    if (LPR >= 64) v = dpp_add<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3
    return v;        // complete in the last lane of every LPR-lane group
}

// out[row] = <M[row, :], vec_s> for s < NSEQ; epi(row, acc) runs in the last lane of the row's group.
// Rows [0, nres) are read from LDS, the rest from global memory (L2); the two parts are separate
// loops over pointers of explicit address spaces -- one loop over a selected generic pointer
// compiles to flat_load + a full vmcnt/lgkmcnt drain per row pass.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef __attribute__((address_space(3))) const v4f lds_cv4f;
typedef __attribute__((address_space(1))) const v4f glb_cv4f;

template <int LPR, int M, int NSEQ, typename Vec, typename Epi>
__device__ __forceinline__ void rowdots_v(const float *ml, const float *mg, int nres, int nrows, int ld,
                                          Vec &&vecf, int tid, Epi &&epi) {
    constexpr int RPP = DR_THREADS / LPR;
    const int k = tid & (LPR - 1), rloc = tid / LPR;
    float4 xv[NSEQ][M];
#pragma unroll
    for (int s = 0; s < NSEQ; s++)
#pragma unroll
        for (int m = 0; m < M; m++) xv[s][m] = vecf(s, (m * LPR + k) * 4);
    auto finish = [&](int row, bool ok, const float4 (&a)[M]) {
        float acc[NSEQ];
#pragma unroll
        for (int s = 0; s < NSEQ; s++) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int m = 0; m < M; m++) {
                t.x = fmaf(a[m].x, xv[s][m].x, t.x); t.y = fmaf(a[m].y, xv[s][m].y, t.y);
                t.z = fmaf(a[m].z, xv[s][m].z, t.z); t.w = fmaf(a[m].w, xv[s][m].w, t.w);
            }
            acc[s] = group_sum<LPR>((t.x + t.y) + (t.z + t.w));
        }
        if (k == LPR - 1 && ok) epi(row, acc);
    };
    int row0 = 0;
    {   // ---- LDS-resident rows ---------------------------------------------------------------------
        lds_cfloat *mlds = (lds_cfloat *)ml;
        for (; row0 < nres; row0 += RPP) {
            const int row = row0 + rloc;
            const bool ok = row < nrows;
            const int rc = ok ? row : nrows - 1;
            lds_cv4f *src = (lds_cv4f *)(mlds + rc * ld + k * 4);
            float4 a[M];
#pragma unroll
            for (int m = 0; m < M; m++) { const v4f q = src[m * LPR]; a[m] = make_float4(q.x, q.y, q.z, q.w); }
            finish(row, ok, a);
        }
    }
    if (row0 < nrows) {   // ---- streamed rows: the next pass is in flight while this one is reduced -----
        auto load = [&](int r0, float4 (&a)[M]) {
            const int row = r0 + rloc;
            const int rc = row < nrows ? row : nrows - 1;
            glb_cv4f *src = (glb_cv4f *)(mg + (long long)rc * ld + k * 4);
#pragma unroll
            for (int m = 0; m < M; m++) { const v4f q = src[m * LPR]; a[m] = make_float4(q.x, q.y, q.z, q.w); }
        };
        float4 a[M], an[M];
        load(row0, a);
        for (; row0 < nrows; row0 += RPP) {
            const bool more = row0 + RPP < nrows;
            if (more) load(row0 + RPP, an);
            finish(row0 + rloc, row0 + rloc < nrows, a);
            if (more) {
#pragma unroll
                for (int m = 0; m < M; m++) a[m] = an[m];
            }
        }
    }
}

template <int LPR, int M, int NSEQ, typename Epi>
__device__ __forceinline__ void rowdots(const float *ml, const float *mg, int nres, int nrows, int ld,
                                        const float *vec, int vstride, int tid, Epi &&epi) {
    rowdots_v<LPR, M, NSEQ>(ml, mg, nres, nrows, ld,
                            [&](int s, int c) -> float4 { return ld4(vec + s * vstride + c); }, tid, epi);
}

#define FARNN_ROWS_CASE(LPRV, MV, ...)                                                         \
    case (LPRV) * 8 + (MV): rowdots<LPRV, MV, NSEQ>(__VA_ARGS__); break;
#define FARNN_ROWS_DISPATCH(lpr, mm, ...)                                                      \
    switch ((lpr) * 8 + (mm)) {                                                                \
        FARNN_ROWS_CASE(16, 1, __VA_ARGS__) FARNN_ROWS_CASE(16, 2, __VA_ARGS__)                \
        FARNN_ROWS_CASE(16, 3, __VA_ARGS__)                                                    \
        FARNN_ROWS_CASE(32, 1, __VA_ARGS__) FARNN_ROWS_CASE(32, 2, __VA_ARGS__)                \
        FARNN_ROWS_CASE(32, 3, __VA_ARGS__)                                                    \
        FARNN_ROWS_CASE(64, 1, __VA_ARGS__) FARNN_ROWS_CASE(64, 2, __VA_ARGS__)                \
        FARNN_ROWS_CASE(64, 3, __VA_ARGS__)                                                    \
        default: break;                                                                        \
    }

#define FARNN_ROWS_CASE_V(LPRV, MV, ...)                                                       \
    case (LPRV) * 8 + (MV): rowdots_v<LPRV, MV, NSEQ>(__VA_ARGS__); break;
#define FARNN_ROWS_DISPATCH_V(lpr, mm, ...)                                                    \
    switch ((lpr) * 8 + (mm)) {                                                                \
        FARNN_ROWS_CASE_V(16, 1, __VA_ARGS__) FARNN_ROWS_CASE_V(16, 2, __VA_ARGS__)            \
        FARNN_ROWS_CASE_V(16, 3, __VA_ARGS__)                                                  \
        FARNN_ROWS_CASE_V(32, 1, __VA_ARGS__) FARNN_ROWS_CASE_V(32, 2, __VA_ARGS__)            \
        FARNN_ROWS_CASE_V(32, 3, __VA_ARGS__)                                                  \
        FARNN_ROWS_CASE_V(64, 1, __VA_ARGS__) FARNN_ROWS_CASE_V(64, 2, __VA_ARGS__)            \
        FARNN_ROWS_CASE_V(64, 3, __VA_ARGS__)                                                  \
        default: break;                                                                        \
    }

__device__ __forceinline__ float gate_sigmoid(float x, float k) { return 1.0f / (1.0f + expf(-(x * k))); }

template <int NSEQ>
__global__ void __launch_bounds__(DR_THREADS)
decomp_rows_kernel(const DecompRowsParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x;
    const int dir = blockIdx.x & 1, grp = blockIdx.x >> 1;
    const int S = p.S, SP = p.SP, R = p.R, Rp = p.Rp, ld2 = p.ld2, ld3 = p.ld3, vbs = p.vbs;
    const int farnn = p.farnn;
    const int Lr = (p.L + 3) & ~3;
    const int tvl = Rp + (farnn >= 1 ? SP : 0) + (farnn == 2 ? SP : 0);
    const int s2l = Rp + SP;                                  // floats per SUM2 vector

    // ---- LDS carve (every block a multiple of 16 bytes) ----------------------------------------
    int *tok = reinterpret_cast<int *>(smem);                 // [NSEQ][Lr]
    float *Hinit = smem + NSEQ * Lr;                          // [ld2]
    float *H = Hinit + ld2;                                   // [NSEQ][ld2]   state (farnn==2 only)
    float *HB = H + NSEQ * ld2;                               // [NSEQ][vbs]   hb: what the factors see
    float *Z = HB + NSEQ * vbs;                               // [NSEQ][SP]    update gate (farnn==2)
    float *SUM1 = Z + NSEQ * SP;                              // [NSEQ][2*SP]  P1 row sums: z | r
    float *SUM2 = SUM1 + NSEQ * 2 * SP;                       // [NSEQ][s2l]   P2 row sums: rr | z (farnn==1)
    float *SUM3 = SUM2 + NSEQ * s2l;                          // [NSEQ][SP]    P3 row sums
    float *TV = SUM3 + NSEQ * SP;                             // [2][NSEQ][tvl]  per-token vectors
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
        b = __builtin_amdgcn_readfirstlane(b);                // workgroup-uniform: keep it in SGPRs
        bseq[s] = b;
        slen[s] = have ? __builtin_amdgcn_readfirstlane((int)p.len[b]) : 0;
        nst[s] = have ? (p.full ? p.L : slen[s]) : -1;        // -1: no sequence in this slot
        nmax = nst[s] > nmax ? nst[s] : nmax;
    }
    const float *hinit = dir == 0 ? p.h0 : p.hT;
    float *stash_base = dir == 0 ? p.A : p.Bk;

    // ---- set-up ------------------------------------------------------------------------------------
    for (int i = tid; i < (int)(TV - Hinit); i += DR_THREADS) Hinit[i] = 0.0f;
#pragma unroll
    for (int s = 0; s < NSEQ; s++)
        for (int k = tid; k < nst[s]; k += DR_THREADS) {
            const int idx = (dir == 0) ? k : (k < slen[s] ? slen[s] - 1 - k : k);
            tok[s * Lr + k] = (int)p.x[(long long)bseq[s] * p.L + idx];
        }
    {   // resident rows: global -> LDS
        const float *src[3] = {p.P1, p.P2[dir], p.P3[dir]};
        float *dst[3] = {L1, L2, L3};
        const long long cnt[3] = {(long long)p.res1 * ld2, (long long)p.res2 * ld2, (long long)p.res3 * ld3};
#pragma unroll
        for (int q = 0; q < 3; q++)
            for (long long i = (long long)tid * 4; i < cnt[q]; i += DR_THREADS * 4) st4(dst[q] + i, ld4(src[q] + i));
    }
    __syncthreads();
    for (int j = tid; j < S; j += DR_THREADS) {
        const float hv = hinit[j];
        Hinit[j] = hv;
#pragma unroll
        for (int s = 0; s < NSEQ; s++) {
            if (nst[s] < 0) continue;
            H[s * ld2 + j] = hv;
            HB[s * vbs + j] = hv;
            stash_base[(long long)bseq[s] * (p.L + 1) * SP + j] = hv;
        }
    }
    for (int j = S + tid; j < SP; j += DR_THREADS)
#pragma unroll
        for (int s = 0; s < NSEQ; s++)
            if (nst[s] >= 0) stash_base[(long long)bseq[s] * (p.L + 1) * SP + j] = 0.0f;

    // per-token vectors: element e of sequence s at step t
    auto tv_load = [&](int s, int e, int t) -> float {
        if (nst[s] <= 0) return 0.0f;
        const int tk = tok[s * Lr + (t < nst[s] ? t : nst[s] - 1)];
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
    // element-wise phases: thread (es, ej) owns state entry ej of sequence es
    const int es = tid < NSEQ * S ? tid / S : 0, ej = tid < NSEQ * S ? tid % S : -1;
    int e_nst = -1;
    long long e_stash = 0;
#pragma unroll
    for (int s = 0; s < NSEQ; s++)
        if (s == es) { e_nst = nst[s]; e_stash = (long long)bseq[s] * (p.L + 1) * SP; }
    __syncthreads();

    const float sig_k = p.sig_k;
    const int nl_mode = p.nl;
    for (int t = 0; t < nmax; t++) {
        const int cur = t & 1, nxt = cur ^ 1;
        const float *TVc = TV + cur * ntv;
        // next step's per-token vectors: loads issued now, parked in registers until after P3
        float pf[DR_MAX_PF];
#pragma unroll
        for (int i = 0; i < DR_MAX_PF; i++)
            pf[i] = (pf_e[i] >= 0 && t + 1 < nmax && !(p.dbg & 8)) ? tv_load(pf_s[i], pf_e[i], t + 1) : 0.0f;
        if (farnn == 2) {
            // ---- P1: both gate pre-activations from h  (:143-148) -----------------------------------
            auto epi1 = [&](int row, const float (&acc)[NSEQ]) {
                const int o = row < S ? row : SP + (row - S);
#pragma unroll
                for (int s = 0; s < NSEQ; s++) SUM1[s * 2 * SP + o] = acc[s];
            };
            FARNN_ROWS_DISPATCH(p.lpr2, p.m2, L1, p.P1, p.res1, p.n1, ld2, H, ld2, tid, epi1)
            wg_barrier_lds();
            // ---- E1: z, r; hb = (1-r) h_init + r h  (:149-151) ---------------------------------------
            if (ej >= 0 && t < e_nst) {
                const float *tv = TVc + es * tvl;
                Z[es * SP + ej] = gate_sigmoid(SUM1[es * 2 * SP + ej] + tv[Rp + ej], sig_k);
                const float rg = gate_sigmoid(SUM1[es * 2 * SP + SP + ej] + tv[Rp + SP + ej], sig_k);
                HB[es * vbs + ej] = (1.0f - rg) * Hinit[ej] + rg * H[es * ld2 + ej];
            }
            wg_barrier_lds();
        }
        {   // ---- P2: Sa^T . hb  (:169 / :174); farnn==1: the z pre-activation from the same h -----------
            auto epi2 = [&](int row, const float (&acc)[NSEQ]) {
                const int o = row < R ? row : Rp + (row - R);
#pragma unroll
                for (int s = 0; s < NSEQ; s++) SUM2[s * s2l + o] = acc[s];
            };
            if (!(p.dbg & 1)) FARNN_ROWS_DISPATCH(p.lpr2, p.m2, L2, p.P2[dir], p.res2, p.n2, ld2, HB, vbs, tid, epi2)
            wg_barrier_lds();
        }
        {   // ---- P3: nx = Sb . (v * sums) + W(^T) . hb  (:170-173 / :175-178) ---------------------------
            auto epi3 = [&](int row, const float (&acc)[NSEQ]) {
#pragma unroll
                for (int s = 0; s < NSEQ; s++) SUM3[s * SP + row] = acc[s];
            };
            // the input vector [rr | hb] is formed on the fly: rr = v * (P2 sums)
            auto vec3 = [&](int s, int c) -> float4 {
                if (c < Rp) {
                    const float4 a = ld4(SUM2 + s * s2l + c), v = ld4(TVc + s * tvl + c);
                    return make_float4(a.x * v.x, a.y * v.y, a.z * v.z, a.w * v.w);
                }
                return ld4(HB + s * vbs + (c - Rp));
            };
            if (!(p.dbg & 1)) FARNN_ROWS_DISPATCH_V(p.lpr3, p.m3, L3, p.P3[dir], p.res3, p.n3, ld3, vec3, tid, epi3)
        }
        {   // park the prefetched per-token vectors BEFORE this step's stash stores are issued: vmcnt retires
            // in order, so waiting for these loads later would also wait for every younger store
            float *TVn = TV + nxt * ntv;
#pragma unroll
            for (int i = 0; i < DR_MAX_PF; i++)
                if (pf_e[i] >= 0) TVn[tid + i * DR_THREADS] = pf[i];
        }
        wg_barrier_lds();
        // ---- E3: non-linearity, gate mix, stash  (:183-196) ------------------------------------------
        if (ej >= 0 && t < e_nst) {
            const float nx = (p.dbg & 2) ? SUM3[es * SP + ej] : apply_nl(SUM3[es * SP + ej], nl_mode);
            float hn = nx;
            if (farnn == 2) {
                const float z = Z[es * SP + ej];
                hn = (1.0f - z) * H[es * ld2 + ej] + z * nx;
                H[es * ld2 + ej] = hn;
            } else if (farnn == 1) {
                const float z = gate_sigmoid(SUM2[es * s2l + Rp + ej] + TVc[es * tvl + Rp + ej], sig_k);
                hn = (1.0f - z) * HB[es * vbs + ej] + z * nx;
                HB[es * vbs + ej] = hn;
            } else {
                HB[es * vbs + ej] = hn;
            }
            if (!(p.dbg & 4)) stash_base[e_stash + (long long)(t + 1) * SP + ej] = hn;
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

struct RowsCfg { int LPR, M; };
inline RowsCfg rows_cfg(int cols) {
    const int c4 = (cols + 3) / 4;
    RowsCfg best{0, 0};
    int bestsz = 1 << 30;
    const int lprs[3] = {16, 32, 64};
    for (int i = 0; i < 3; i++) {
        const int mm = (c4 + lprs[i] - 1) / lprs[i];
        if (mm <= 3 && lprs[i] * mm < bestsz) { best = RowsCfg{lprs[i], mm}; bestsz = lprs[i] * mm; }
    }
    return best;
}

// everything the launcher needs, filled by build_rows_pack() in farnn_hip.hip
struct DecompRowsPack {
    bool ok = false;
    float *P1 = nullptr, *P2[2] = {nullptr, nullptr}, *P3[2] = {nullptr, nullptr}, *Gz = nullptr, *Gr = nullptr;
    int n1 = 0, n2 = 0, n3 = 0, ld2 = 0, ld3 = 0, lpr2 = 0, m2 = 0, lpr3 = 0, m3 = 0, vbs = 0;
};

struct RowsPlan { int nseq, res1, res2, res3; size_t lds; };

inline bool rows_plan(const DecompRowsPack &k, const DecompWeights &w, int B, int L, RowsPlan &pl) {
    const int Lr = (L + 3) & ~3;
    const int tvl = w.Rp + (w.farnn >= 1 ? w.SP : 0) + (w.farnn == 2 ? w.SP : 0);
    int nseq = 1;
    while (nseq < 4 && 2 * ((B + nseq - 1) / nseq) > 256) nseq *= 2;       // about one workgroup per CU
    if (const char *e = getenv("FARNN_ROWS_NSEQ")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4) nseq = v; }
    if (nseq == 4 && (k.m2 > 2 || k.m3 > 2)) nseq = 2;      // 4 x 3 float4 of vector slices would spill
    for (; nseq >= 1; nseq /= 2) {
        if (nseq * tvl > DR_MAX_PF * DR_THREADS) continue;
        const size_t fixed = 4 * ((size_t)nseq * Lr + k.ld2 + (size_t)nseq * k.ld2 + (size_t)nseq * k.vbs +
                                  (size_t)nseq * (w.SP + 2 * w.SP + (w.Rp + w.SP) + w.SP) + 2ull * nseq * tvl);
        const size_t cap = 159 * 1024;
        if (fixed + 16 * 1024 > cap) continue;               // leave room for at least some resident rows
        size_t left = cap - fixed;
        auto take = [&](int nrows, int ld, int lpr) {
            if (nrows == 0) return 0;
            const int rpp = DR_THREADS / lpr;
            long long fit = (long long)(left / ((size_t)ld * 4));
            int res = fit >= nrows ? nrows : (int)(fit / rpp) * rpp;
            left -= (size_t)res * ld * 4;
            return res;
        };
        pl.res3 = take(k.n3, k.ld3, k.lpr3);
        pl.res2 = take(k.n2, k.ld2, k.lpr2);
        pl.res1 = take(k.n1, k.ld2, k.lpr2);
        pl.nseq = nseq;
        pl.lds = cap - left;
        return true;
    }
    return false;
}

template <int NSEQ>
inline int launch_rows_n(const DecompRowsParams &p, int groups, size_t lds, hipStream_t s) {
    static int raised = -1;     // per process; the attribute is per (device, function) but monotone in lds
    if ((int)lds > raised) {
        FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(decomp_rows_kernel<NSEQ>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = 160 * 1024;
    }
    decomp_rows_kernel<NSEQ><<<dim3(2 * groups), dim3(DR_THREADS), lds, s>>>(p);
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

inline int launch_decomp_rows(const DecompRowsPack &k, const DecompWeights &w, const RowsPlan &pl,
                              const int64_t *x, const int64_t *len, const int *order, float *A, float *Bk,
                              int B, int L, int full, hipStream_t s) {
    DecompRowsParams p;
    p.P1 = k.P1; p.P2[0] = k.P2[0]; p.P2[1] = k.P2[1]; p.P3[0] = k.P3[0]; p.P3[1] = k.P3[1];
    p.n1 = k.n1; p.n2 = k.n2; p.n3 = k.n3; p.ld2 = k.ld2; p.ld3 = k.ld3;
    p.lpr2 = k.lpr2; p.m2 = k.m2; p.lpr3 = k.lpr3; p.m3 = k.m3;
    p.res1 = pl.res1; p.res2 = pl.res2; p.res3 = pl.res3; p.vbs = k.vbs;
    p.Vgen = w.Vgen; p.Gz = k.Gz; p.Gr = k.Gr; p.h0 = w.h0; p.hT = w.hT;
    p.x = x; p.len = len; p.order = order; p.A = A; p.Bk = Bk;
    p.B = B; p.L = L; p.S = w.S; p.SP = w.SP; p.R = w.R; p.Rp = w.Rp; p.farnn = w.farnn; p.nl = w.nl;
    p.full = full; p.sig_k = w.sig_k;
    { const char *e = getenv("FARNN_DBG"); p.dbg = e ? atoi(e) : 0; }
    const int groups = (B + pl.nseq - 1) / pl.nseq;
    if (pl.nseq == 4) return launch_rows_n<4>(p, groups, pl.lds, s);
    if (pl.nseq == 2) return launch_rows_n<2>(p, groups, pl.lds, s);
    return launch_rows_n<1>(p, groups, pl.lds, s);
}

}  // namespace farnn
