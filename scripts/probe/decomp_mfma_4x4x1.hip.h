// EXPERIMENT (not built): the decomposed recurrence on v_mfma_f32_4x4x1_16b_f32.  Parity-green, but
// slower than the VALU rows kernel (215 vs 159 us at rank 50, 533 vs 437 us gated rank 250): the
// 16-block 4x4x1 form issues once per 32 cycles per SIMD for 256 MACs, a quarter of the f32 VALU rate.

// K12 -- the decomposed recurrence (FARNN_S_D_W_I_S.get_forward_score,
// model_decompose_single.py:138-200; FARNN_S_D_W.get_forward_score, model_decompose.py:243-307) for
// the sum semiring, any gate mode (farnn 0/1/2) and any update non-linearity, on the f32 matrix
// cores.
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
// A workgroup owns up to FOUR sequences of one direction and runs them in lock step, so every
// product is a [rows x K] . [K x 4] GEMM -- exactly the shape of v_mfma_f32_4x4x1_16b_f32 (16
// blocks of 4x4x1, exact f32 fma chains): lane l feeds A = M[row0 + l][k] and B = x_{l%4}[k], and
// after the K loop holds rows row0 + 4*(l/4) .. +3 of sequence l%4 (lane map probed on gfx950:
// scripts/probe/mfma4x4.hip).  One instruction = 64 rows x 4 sequences, no cross-lane reduction.
// The weights are packed at create time in that operand order, [tile][k/4][lane][4], so a
// wavefront's read of one piece is 1 KiB contiguous: conflict-free from LDS, coalesced from L2.
// Pieces live in LDS as far as the 160 KiB go (a rank-50 model fits whole), the rest is streamed
// from L2 every step.  Tiles x K-slices are dealt over the wavefronts; K-slice partial sums meet in
// LDS and are finished by element-wise phases (gates, rr = v * sums, non-linearity, stash).
//
// History (profiles/): one wavefront per sequence: 3.5 us/step at R=50 (LDS-latency bound, 2 waves
// per CU).  Row dots on the VALU with DPP reductions, 16 waves: 1.45 us/step -- issue bound: 20% of
// the instructions were FMAs, the rest reductions and per-wave overhead repeated 16 times.
#pragma once
#include "common.hip.h"
#include "decomp_chain.hip.h"

namespace farnn {

constexpr int DM_MAX_PF = 4;          // prefetch registers per thread for the per-token vectors
constexpr int DM_NS = 4;              // sequence slots per workgroup = N of the 4x4x1 MFMA
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;
typedef __attribute__((address_space(1))) const f32x4 glb_cf32x4;

struct DmPhase {
    const float *pk;      // packed [tiles][nq][64][4]
    int nt, nq, nres;     // tiles, k-quads, LDS-resident pieces (prefix of the tile-major piece list)
    int ks, qps;          // K-slices per tile, quads per slice
    int tp;               // nt * 64: floats per sequence in the partial-sum buffers
};

struct DecompMfmaParams {
    DmPhase ph[3][2];             // [P1,P2,P3][dir]   (P1 unused unless farnn==2)
    const float *Vgen, *Gz, *Gr;  // [V][Rp], [V][SP], [V][SP]
    const float *h0, *hT;
    const int64_t *x, *len;
    const int *order;             // folded launch order (batch_prep) or nullptr
    float *A, *Bk;
    int B, L, S, SP, R, Rp, farnn, nl, full, nseq;
    int xs2, xs3;                 // floats per sequence of the P1/P2 and the P3 input vectors
    float sig_k;
    int dbg;
};

// one (tile, K-slice) job of a phase: partial[seq][row] = sum_{k in slice} M[row][k] x_seq[k]
__device__ __forceinline__ void dm_job(const DmPhase &ph, const float *pk_lds, int tile, int q0, int q1,
                                       const float *X, int xs, float *part, int lane) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const float *xb = X + (lane & 3) * xs;
    const long long base = (long long)tile * ph.nq;
    int qres = ph.nres - (int)base;                 // quads of this tile that are LDS resident
    qres = qres < q0 ? q0 : (qres > q1 ? q1 : qres);
    {   // resident pieces: five quads (10 LDS reads) in flight ahead of their 20 MFMAs -- a wavefront
        // that waits for each quad's operands pays the LDS latency once per 4 MFMAs
        lds_cf32x4 *src = (lds_cf32x4 *)((__attribute__((address_space(3))) const float *)pk_lds) + base * 64 + lane;
        constexpr int U = 5;
        for (int q = q0; q < qres; q += U) {
            f32x4 a[U];
            float4 b[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int qq = q + u < qres ? q + u : qres - 1;
                a[u] = src[(long long)qq * 64];
                b[u] = ld4(xb + 4 * qq);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (q + u < qres) {
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].x, b[u].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].y, b[u].y, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].z, b[u].z, acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].w, b[u].w, acc3, 0, 0, 0);
                }
            }
        }
    }
    if (qres < q1) {          // streamed pieces: eight 1 KiB loads in flight per wavefront
        glb_cf32x4 *src = (glb_cf32x4 *)ph.pk + base * 64 + lane;
        constexpr int U = 8;
        for (int q = qres; q < q1; q += U) {
            f32x4 a[U];
#pragma unroll
            for (int u = 0; u < U; u++) a[u] = src[(long long)(q + u < q1 ? q + u : q1 - 1) * 64];
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (q + u < q1) {
                    const float4 b = ld4(xb + 4 * (q + u));
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].x, b.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].y, b.y, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].z, b.z, acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].w, b.w, acc3, 0, 0, 0);
                }
            }
        }
    }
    const f32x4 r = (acc0 + acc1) + (acc2 + acc3);
    st4(part + (lane & 3) * ph.tp + tile * 64 + 4 * (lane >> 2), make_float4(r.x, r.y, r.z, r.w));
}

// all jobs of a phase, dealt over the workgroup's wavefronts; part = [ks][4][tp]
__device__ __forceinline__ void dm_phase(const DmPhase &ph, const float *pk_lds, const float *X, int xs,
                                         float *part, int wave, int nwaves, int lane) {
    if (ph.ks == 1) {
        for (int tile = wave; tile < ph.nt; tile += nwaves) dm_job(ph, pk_lds, tile, 0, ph.nq, X, xs, part, lane);
    } else {
        const int tile = wave % ph.nt, slice = wave / ph.nt;
        if (slice < ph.ks) {
            const int q0 = slice * ph.qps, q1 = q0 + ph.qps < ph.nq ? q0 + ph.qps : ph.nq;
            dm_job(ph, pk_lds, tile, q0 < q1 ? q0 : q1, q1, X, xs, part + (long long)slice * DM_NS * ph.tp, lane);
        }
    }
}

__device__ __forceinline__ float dm_sum(const float *part, const DmPhase &ph, int s, int row) {
    float acc = part[s * ph.tp + row];
    for (int k = 1; k < ph.ks; k++) acc += part[(k * DM_NS + s) * ph.tp + row];
    return acc;
}

__device__ __forceinline__ float gate_sigmoid(float x, float k) { return 1.0f / (1.0f + __expf(-(x * k))); }

// update non-linearity with the hardware exponential: |error| ~1e-7, against the 1e-4 parity bar
__device__ __forceinline__ float dm_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));          // in (0, 1]: no overflow for any x
    return copysignf((1.0f - e) / (1.0f + e), x);
}
__device__ __forceinline__ float dm_nl(float x, int nl) {
    switch (nl) {
        case FARNN_NL_RELU: return fmaxf(x, 0.0f);
        case FARNN_NL_TANH: return dm_tanh(x);
        case FARNN_NL_RELUTANH: return dm_tanh(fmaxf(x, 0.0f));
        default: return x;
    }
}

__global__ void __launch_bounds__(512)
decomp_mfma_kernel(const DecompMfmaParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthreads >> 6;
    const int dir = blockIdx.x & 1, grp = blockIdx.x >> 1;
    const int S = p.S, SP = p.SP, R = p.R, Rp = p.Rp, farnn = p.farnn, NSEQ = p.nseq;
    const int xs2 = p.xs2, xs3 = p.xs3;
    const int Lr = (p.L + 3) & ~3;
    const int tvl = Rp + (farnn >= 1 ? SP : 0) + (farnn == 2 ? SP : 0);
    const DmPhase &P1 = p.ph[0][dir], &P2 = p.ph[1][dir], &P3 = p.ph[2][dir];

    // ---- LDS carve (every block a multiple of 16 bytes) ----------------------------------------
    int *tok = reinterpret_cast<int *>(smem);                 // [4][Lr]
    float *Hinit = smem + DM_NS * Lr;                         // [SP]
    float *X1 = Hinit + SP;                                   // [4][xs2]  h        (input of P1; farnn==2)
    float *X2 = X1 + DM_NS * xs2;                             // [4][xs2]  hb       (input of P2)
    float *X3 = X2 + DM_NS * xs2;                             // [4][xs3]  rr | hb  (input of P3)
    float *Z = X3 + DM_NS * xs3;                              // [4][SP]   update gate (farnn==2)
    float *TV = Z + DM_NS * SP;                               // [2][4][tvl]  per-token vectors
    float *PS1 = TV + 2 * DM_NS * tvl;                        // [ks][4][tp]  partial sums
    float *PS2 = PS1 + (farnn == 2 ? P1.ks * DM_NS * P1.tp : 0);
    float *PS3 = PS2 + P2.ks * DM_NS * P2.tp;
    float *K1 = PS3 + P3.ks * DM_NS * P3.tp;                  // resident pieces
    float *K2 = K1 + (farnn == 2 ? (long long)P1.nres * 256 : 0);
    float *K3 = K2 + (long long)P2.nres * 256;

    // ---- sequences of this workgroup -------------------------------------------------------------
    int bseq[DM_NS], nst[DM_NS], slen[DM_NS];
    int nmax = 0;
#pragma unroll
    for (int s = 0; s < DM_NS; s++) {
        const int r = grp * NSEQ + s;                         // rank by length (descending)
        const bool have = s < NSEQ && r < p.B;
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
    for (int i = tid; i < (int)(PS1 - Hinit); i += nthreads) Hinit[i] = 0.0f;
#pragma unroll
    for (int s = 0; s < DM_NS; s++)
        for (int k = tid; k < nst[s]; k += nthreads) {
            const int idx = (dir == 0) ? k : (k < slen[s] ? slen[s] - 1 - k : k);
            tok[s * Lr + k] = (int)p.x[(long long)bseq[s] * p.L + idx];
        }
    {   // resident pieces: global -> LDS
        const float *src[3] = {P1.pk, P2.pk, P3.pk};
        float *dst[3] = {K1, K2, K3};
        const long long cnt[3] = {farnn == 2 ? (long long)P1.nres * 256 : 0, (long long)P2.nres * 256,
                                  (long long)P3.nres * 256};
#pragma unroll
        for (int q = 0; q < 3; q++)
            for (long long i = (long long)tid * 4; i < cnt[q]; i += (long long)nthreads * 4) st4(dst[q] + i, ld4(src[q] + i));
    }
    __syncthreads();
    for (int j = tid; j < S; j += nthreads) {
        const float hv = hinit[j];
        Hinit[j] = hv;
#pragma unroll
        for (int s = 0; s < DM_NS; s++) {
            if (nst[s] < 0) continue;
            X1[s * xs2 + j] = hv;
            X2[s * xs2 + j] = hv;
            X3[s * xs3 + Rp + j] = hv;
            stash_base[(long long)bseq[s] * (p.L + 1) * SP + j] = hv;
        }
    }
    for (int j = S + tid; j < SP; j += nthreads)
#pragma unroll
        for (int s = 0; s < DM_NS; s++)
            if (nst[s] >= 0) stash_base[(long long)bseq[s] * (p.L + 1) * SP + j] = 0.0f;

    // per-token vectors: element e of sequence s at step t
    auto tv_load = [&](int s, int e, int t) -> float {
        int n = nst[0];
#pragma unroll
        for (int q = 1; q < DM_NS; q++) n = s == q ? nst[q] : n;
        if (n <= 0) return 0.0f;
        const int tk = tok[s * Lr + (t < n ? t : n - 1)];
        if (e < Rp) return p.Vgen[(long long)tk * Rp + e];
        if (e < Rp + SP) return p.Gz[(long long)tk * SP + (e - Rp)];
        return p.Gr[(long long)tk * SP + (e - Rp - SP)];
    };
    const int ntv = DM_NS * tvl;
    int pf_s[DM_MAX_PF], pf_e[DM_MAX_PF];                     // loop-invariant split of the prefetch slots
#pragma unroll
    for (int i = 0; i < DM_MAX_PF; i++) {
        const int e = tid + i * nthreads;
        pf_s[i] = e < ntv ? e / tvl : 0;
        pf_e[i] = e < ntv ? e % tvl : -1;
    }
    for (int i = 0; i < DM_MAX_PF; i++)
        if (pf_e[i] >= 0) TV[tid + i * nthreads] = tv_load(pf_s[i], pf_e[i], 0);
    __syncthreads();

    const float sig_k = p.sig_k;
    const int nl_mode = p.nl;
    // element-wise phases: element e -> (sequence e & 3, index e >> 2)
    for (int t = 0; t < nmax; t++) {
        const int cur = t & 1, nxt = cur ^ 1;
        const float *TVc = TV + cur * ntv;
        bool act[DM_NS];
#pragma unroll
        for (int s = 0; s < DM_NS; s++) act[s] = t < nst[s];
        auto active = [&](int s) { bool a = act[0];
#pragma unroll
            for (int q = 1; q < DM_NS; q++) a = s == q ? act[q] : a;
            return a; };
        // next step's per-token vectors: loads issued now, parked in registers until after P3
        float pf[DM_MAX_PF];
#pragma unroll
        for (int i = 0; i < DM_MAX_PF; i++)
            pf[i] = (pf_e[i] >= 0 && t + 1 < nmax) ? tv_load(pf_s[i], pf_e[i], t + 1) : 0.0f;
        if (farnn == 2) {
            // ---- P1: both gate pre-activations from h  (:143-148) -----------------------------------
            if (!(p.dbg & 1)) dm_phase(P1, K1, X1, xs2, PS1, wave, nwaves, lane);
            wg_barrier_lds();
            // ---- E1: z, r; hb = (1-r) h_init + r h  (:149-151) ---------------------------------------
            for (int e = tid; e < DM_NS * S; e += nthreads) {
                const int s = e & 3, j = e >> 2;
                if (!active(s)) continue;
                const float *tv = TVc + s * tvl;
                Z[s * SP + j] = gate_sigmoid(dm_sum(PS1, P1, s, j) + tv[Rp + j], sig_k);
                const float rg = gate_sigmoid(dm_sum(PS1, P1, s, S + j) + tv[Rp + SP + j], sig_k);
                const float hb = (1.0f - rg) * Hinit[j] + rg * X1[s * xs2 + j];
                X2[s * xs2 + j] = hb;
                X3[s * xs3 + Rp + j] = hb;
            }
            wg_barrier_lds();
        }
        // ---- P2: Sa^T . hb  (:169 / :174); farnn==1: the z pre-activation rows ride along ---------
        if (!(p.dbg & 1)) dm_phase(P2, K2, X2, xs2, PS2, wave, nwaves, lane);
        wg_barrier_lds();
        // ---- E2: rr = v * sums  (:170 / :175) ------------------------------------------------------------
        for (int e = tid; e < DM_NS * R; e += nthreads) {
            const int s = e & 3, r = e >> 2;
            X3[s * xs3 + r] = TVc[s * tvl + r] * dm_sum(PS2, P2, s, r);
        }
        wg_barrier_lds();
        // ---- P3: nx = Sb . rr + W(^T) . hb  (:171-173 / :176-178) ---------------------------------------
        if (!(p.dbg & 1)) dm_phase(P3, K3, X3, xs3, PS3, wave, nwaves, lane);
        {   // park the prefetched per-token vectors BEFORE this step's stash stores are issued: vmcnt retires
            // in order, so waiting for these loads later would also wait for every younger store
            float *TVn = TV + nxt * ntv;
#pragma unroll
            for (int i = 0; i < DM_MAX_PF; i++)
                if (pf_e[i] >= 0) TVn[tid + i * nthreads] = pf[i];
        }
        wg_barrier_lds();
        // ---- E3: non-linearity, gate mix, stash  (:183-196) ------------------------------------------
        for (int e = tid; e < DM_NS * S; e += nthreads) {
            const int s = e & 3, j = e >> 2;
            if (!active(s)) continue;
            const float nx = (p.dbg & 2) ? dm_sum(PS3, P3, s, j) : dm_nl(dm_sum(PS3, P3, s, j), nl_mode);
            float hn = nx;
            if (farnn == 2) {
                const float z = Z[s * SP + j];
                hn = (1.0f - z) * X1[s * xs2 + j] + z * nx;
                X1[s * xs2 + j] = hn;
            } else {
                if (farnn == 1) {
                    const float z = gate_sigmoid(dm_sum(PS2, P2, s, R + j) + TVc[s * tvl + Rp + j], sig_k);
                    hn = (1.0f - z) * X2[s * xs2 + j] + z * nx;
                }
                X2[s * xs2 + j] = hn;
                X3[s * xs3 + Rp + j] = hn;
            }
            long long sb = (long long)bseq[0];
#pragma unroll
            for (int q = 1; q < DM_NS; q++) sb = s == q ? (long long)bseq[q] : sb;
            stash_base[(sb * (p.L + 1) + t + 1) * SP + j] = hn;
        }
        wg_barrier_lds();
    }
}

// ---- packing (create time) ------------------------------------------------------------------------
// Logical matrices (fwd shown):  P2 row r < R: Sa[:, r] (bwd: the backward input scaling o folded
// in); row R + j (farnn==1): Wss1[:, j].  P1 row j: Wss1[:, j]; row S + j: Wss2[:, j].
// P3 row j: [Sb[j, :] | pad to Rp | Wd[:, j]] (fwd: row scaled by o[j]; bwd: Wd[s] = W[j][s] o[s]).
struct PackSrc {
    const float *S1, *S2, *W;     // [S][Rp], [S][Rp], [S][SP]
    const float *Wss1, *Wss2;     // [S][SP] or nullptr
    const float *o;               // [SP]
    int S, SP, R, Rp, farnn;
};

__device__ __forceinline__ float pack_elem(const PackSrc &q, int which, int dir, int row, int c) {
    if (which == 0) {             // P1 [2S][S]
        if (row >= 2 * q.S || c >= q.S) return 0.0f;
        return row < q.S ? q.Wss1[(long long)c * q.SP + row] : q.Wss2[(long long)c * q.SP + (row - q.S)];
    }
    if (which == 1) {             // P2 [R (+S)][S]
        const int nrows = q.R + (q.farnn == 1 ? q.S : 0);
        if (row >= nrows || c >= q.S) return 0.0f;
        if (row < q.R) return dir == 0 ? q.S1[(long long)c * q.Rp + row] : q.S2[(long long)c * q.Rp + row] * q.o[c];
        return q.Wss1[(long long)c * q.SP + (row - q.R)];
    }
    if (row >= q.S) return 0.0f;  // P3 [S][Rp + S]
    if (c < q.R) return dir == 0 ? q.S2[(long long)row * q.Rp + c] * q.o[row] : q.S1[(long long)row * q.Rp + c];
    if (c >= q.Rp && c < q.Rp + q.S) {
        const int s = c - q.Rp;
        return dir == 0 ? q.W[(long long)s * q.SP + row] * q.o[row] : q.W[(long long)row * q.SP + s] * q.o[s];
    }
    return 0.0f;
}

// out[((tile * nq + kq) * 64 + lane) * 4 + j] = M[tile * 64 + lane][4 kq + j]
__global__ void pack_pieces_kernel(PackSrc q, float *out, int which, int dir, int nt, int nq) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nt * nq * 256) return;
    const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const long long piece = idx >> 8;
    const int tile = (int)(piece / nq), kq = (int)(piece % nq);
    out[idx] = pack_elem(q, which, dir, tile * 64 + lane, 4 * kq + j);
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

// everything the launcher needs, filled by build_mfma_pack() in farnn_hip.hip
struct DecompMfmaPack {
    bool ok = false;
    float *pk[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    int nt[3] = {0, 0, 0}, nq[3] = {0, 0, 0};
    float *Gz = nullptr, *Gr = nullptr;
    int xs2 = 0, xs3 = 0;
};

struct MfmaPlan { int nseq, nwaves, ks[3], qps[3], nres[3]; size_t lds; };

inline bool mfma_plan(const DecompMfmaPack &k, const DecompWeights &w, int B, int L, MfmaPlan &pl) {
    const int Lr = (L + 3) & ~3;
    const int tvl = w.Rp + (w.farnn >= 1 ? w.SP : 0) + (w.farnn == 2 ? w.SP : 0);
    int nseq = 1;
    while (nseq < DM_NS && 2 * ((B + nseq - 1) / nseq) > 256) nseq *= 2;   // about one workgroup per CU
    if (const char *e = getenv("FARNN_ROWS_NSEQ")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4) nseq = v; }
    int nwaves = 8;
    if (const char *e = getenv("FARNN_ROWS_WAVES")) { int v = atoi(e); if (v >= 1 && v <= 8) nwaves = v; }
    if (DM_NS * tvl > DM_MAX_PF * nwaves * 64) return false;
    size_t fixed = (size_t)DM_NS * Lr + w.SP + 2ull * DM_NS * k.xs2 + (size_t)DM_NS * k.xs3 +
                   (size_t)DM_NS * w.SP + 2ull * DM_NS * tvl;
    for (int f = (w.farnn == 2 ? 0 : 1); f < 3; f++) {
        const int nt = k.nt[f], nq = k.nq[f];
        int ks = nt >= nwaves ? 1 : nwaves / nt;
        if (ks > nq) ks = nq;
        pl.ks[f] = ks;
        pl.qps[f] = (nq + ks - 1) / ks;
        fixed += (size_t)ks * DM_NS * nt * 64;
    }
    if (w.farnn != 2) { pl.ks[0] = 1; pl.qps[0] = 0; }
    fixed *= 4;
    const size_t cap = 159 * 1024;
    if (fixed + 8 * 1024 > cap) return false;
    size_t left = cap - fixed;
    const int order3[3] = {2, 1, 0};                 // residency priority: P3, P2, P1
    pl.nres[0] = pl.nres[1] = pl.nres[2] = 0;
    for (int i = 0; i < 3; i++) {
        const int f = order3[i];
        if (f == 0 && w.farnn != 2) continue;
        const long long pieces = (long long)k.nt[f] * k.nq[f];
        const long long fit = (long long)(left / 1024);
        pl.nres[f] = (int)(fit >= pieces ? pieces : fit);
        left -= (size_t)pl.nres[f] * 1024;
    }
    pl.nseq = nseq; pl.nwaves = nwaves;
    pl.lds = cap - left;
    return true;
}

inline int launch_decomp_mfma(const DecompMfmaPack &k, const DecompWeights &w, const MfmaPlan &pl,
                              const int64_t *x, const int64_t *len, const int *order, float *A, float *Bk,
                              int B, int L, int full, hipStream_t s) {
    DecompMfmaParams p;
    for (int f = 0; f < 3; f++)
        for (int d = 0; d < 2; d++) {
            DmPhase &ph = p.ph[f][d];
            ph.pk = k.pk[f][d]; ph.nt = k.nt[f]; ph.nq = k.nq[f]; ph.nres = pl.nres[f];
            ph.ks = pl.ks[f]; ph.qps = pl.qps[f]; ph.tp = k.nt[f] * 64;
        }
    p.Vgen = w.Vgen; p.Gz = k.Gz; p.Gr = k.Gr; p.h0 = w.h0; p.hT = w.hT;
    p.x = x; p.len = len; p.order = order; p.A = A; p.Bk = Bk;
    p.B = B; p.L = L; p.S = w.S; p.SP = w.SP; p.R = w.R; p.Rp = w.Rp; p.farnn = w.farnn; p.nl = w.nl;
    p.full = full; p.nseq = pl.nseq; p.xs2 = k.xs2; p.xs3 = k.xs3; p.sig_k = w.sig_k;
    { const char *e = getenv("FARNN_DBG"); p.dbg = e ? atoi(e) : 0; }
    static bool raised = false;
    if (!raised) {
        FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(decomp_mfma_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    const int groups = (B + pl.nseq - 1) / pl.nseq;
    decomp_mfma_kernel<<<dim3(2 * groups), dim3(pl.nwaves * 64), pl.lds, s>>>(p);
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

}  // namespace farnn
