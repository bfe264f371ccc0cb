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
// One workgroup per pair of sequences and direction; the direction's matrices live in LDS when they fit, every
// matrix-vector product of a step is split over the workgroup's wavefronts, everything a step reads from global
// memory is fetched one step ahead, and the step barriers wait for LDS only (a __syncthreads would drain the
// prefetches and the stash stores: one L2 round trip per barrier).  Parameter gradients that are sums of outer products
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
    float *PRE;               // [.][S]  forward chain's pre-activation before the output mask
    float *DS, *AB;           // [B][L][K] d loss / d (pre-priority) scores; [B][L][S] alpha*beta
    float *SC;                // [B][L][K] scores after the priority layer (CRF mode: the emissions)
    // GRU-style gates (farnn 1/2, model_decompose_single.py:143-154,193-198)
    int farnn;
    float sig_k;
    const float *Wss1, *Wrs1, *bs1, *Wss2, *Wrs2, *bs2;     // [S][S], [R][S], [S]
    const float *Wss1T, *Wss2T, *Wrs1T, *Wrs2T;             // [S][S], [S][R]
    float *ZGf, *ZGb, *RGf, *RGb, *CDf, *CDb;               // [.][S] gates z, r and the candidate state of every step
    float *DAZf, *DAZb, *DARf, *DARb;                       // [.][S] adjoints of the gate pre-activations
    float *VRf, *VRb;                                       // [.][R] v_t rows (for d Wrs)
    float *HBARf;                                           // [.][S] forward chain input hbar_t (= f_{t-1} without gates)
    const float *GV1, *GV2;                                 // [V][S] Vgen Wrs1, Vgen Wrs2: the input halves of the gates, hoisted out of the recurrences
    float *dGV1, *dGV2;                                     // [V][S] their adjoints (rows of the words that occur)
    int nss_f, nss_b;         // through-L2 chain kernels: how many of their S x S matrices still fit in LDS (0..3)
    const float *trans;       // [K][K] CRF transitions (CRF mode) and the per-sequence partials of their gradient
    float *dtrans_part;       // [B][K][K]
    float *dVgen, *dOsum, *dh0, *dhT, *loss;
    int32_t *tags;
    int *err;                 // sticky device flag: bit 0 = a label outside 0..K-1 at a valid position
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

constexpr int TR_THREADS = 512;  // chain kernels: 8 wavefronts split every matrix-vector product's reduction index
constexpr int TR_VPT = 2;        // per-thread slots covering 2 R (R <= 512) and 2 S (S <= 512) values
constexpr int TR_NSEQ = 2;      // sequences per workgroup of the chain kernels when the matrices sit in LDS (every element read
                                 // feeds both); TR_NSEQ_L2 when they are read through L2 every step (that mode is L2-rate bound)
constexpr int TR_NSEQ_L2 = 4;

// part[(wave*2+q)*J + j] = sum over this wavefront's share of k of in[q][k] M[k][j]  (M row-major [K][J], in LDS
// or global memory; `in` = two vectors of stride ldin in LDS).  The caller adds the wavefronts' shares after a barrier.
// Wavefront w works on column block jb = w % njb (64 columns) and reduction share ks = w / njb of nks = 8 / njb.
__device__ __forceinline__ int mv_nks(int J) {
    const int njb = (J + 63) >> 6, n = (TR_THREADS / 64) / njb;
    return n < 1 ? 1 : n;
}
__device__ __forceinline__ int mv_pad(int n) { return ((n + 3) & ~3) + 8; }     // stride of a matvec input vector in LDS

// part[(ks*2+q)*J + j] = sum over share ks of k of in[q][k] M[k][j]   (M row-major [K][J] in LDS or global memory; `in` =
// two vectors of stride ldin = mv_pad(K) in LDS, 16-byte aligned, zero beyond K).  Measured on the first version (one
// LDS read per operand): 5.8 k cycles for the two products of a step, 2/3 of the LDS instructions being broadcast reads
// of `in` -- those are 16-byte reads here, and the shares start at multiples of 4.
template <bool INLDS, int NS>
__device__ __forceinline__ void matvec2_partial(float *part, const float *in, int ldin, const float *__restrict__ M, int K,
                                                int J, int tid, int /*nthreads*/) {
    const int njb = (J + 63) >> 6, nks = mv_nks(J);
    const int w = tid >> 6, lane = tid & 63;
    const int jb = w % njb, ks = w / njb;
    if (ks >= nks) return;
    int j = jb * 64 + lane;
    asm volatile("" : "+v"(j));       // (opaque: the offsets derived from it are per-call-site loop invariants -- see part2_sum)
    const int K4 = (K + 3) >> 2;
    const int k0 = 4 * ((K4 * ks) / nks), k1r = 4 * ((K4 * (ks + 1)) / nks), k1 = k1r < K ? k1r : K;
    const bool jok = j < J;
    const int jc = jok ? j : J - 1;
    float a[NS][4];
#pragma unroll
    for (int q = 0; q < NS; q++)
#pragma unroll
        for (int e = 0; e < 4; e++) a[q][e] = 0.0f;
    // 32-bit offsets stepped by J (64-bit index arithmetic per operand cost more than the reads).  A round past the
    // end of the share is masked after the reads; the matrix rows it touches are read unguarded when the matrix is
    // in LDS (they stay inside the allocation) and clamped to the last row otherwise.  Every matrix element read
    // feeds the NS sequences of the workgroup.
    int off = k0 * J + jc;
    const int last = (K - 1) * J + jc;
    for (int k = k0; k < k1; k += 8, off += 8 * J) {
        float m[8];
        v4f xa[NS], xb[NS];
#pragma unroll
        for (int q = 0; q < NS; q++) { xa[q] = *(const v4f *)(in + q * ldin + k); xb[q] = *(const v4f *)(in + q * ldin + k + 4); }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int o = off + u * J;
            m[u] = M[INLDS ? o : (o < last ? o : last)];
        }
        const int left = k1 - k;                          // wave-uniform
        if (left < 8) {                                   // whatever sits past the share (another share's inputs, LDS beyond
#pragma unroll                                            // the matrix -- possibly NaN) must not count: the matrix side is zeroed
            for (int u = 0; u < 8; u++) if (u >= left) m[u] = 0.0f;
#pragma unroll
            for (int q = 0; q < NS; q++)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (u >= left) xa[q][u] = 0.0f;
                    if (4 + u >= left) xb[q][u] = 0.0f;
                }
        }
#pragma unroll
        for (int q = 0; q < NS; q++) {
#pragma unroll
            for (int u = 0; u < 4; u++) a[q][u] = fmaf(xa[q][u], m[u], a[q][u]);
#pragma unroll
            for (int u = 0; u < 4; u++) a[q][u] = fmaf(xb[q][u], m[4 + u], a[q][u]);
        }
    }
    if (jok) {
#pragma unroll
        for (int q = 0; q < NS; q++) part[(ks * NS + q) * J + j] = (a[q][0] + a[q][1]) + (a[q][2] + a[q][3]);
    }
}
template <int NS>
__device__ __forceinline__ float part2_sum(const float *part, int J, int q, int j, int /*nw*/) {
    // all shares are read before any is added (at most 8): a read-add chain over LDS costs ~105 cycles per element
    // (scripts/probe/lds_rate.hip), independent reads ~12
    constexpr int NW = TR_THREADS / 64;
    const int nks = mv_nks(J);
    float v[NW];
    // (the element's index is opaque to the optimiser: as a loop invariant each of the eight addresses of each call site and slot
    // was hoisted out of the time loop into a register of its own -- ~100 of them in the gated two-slot kernels, spilled)
    int idx = q * J + j;
    asm volatile("" : "+v"(idx));
#pragma unroll
    for (int w = 0; w < NW; w++) v[w] = w < nks ? part[(w < nks ? w : 0) * NS * J + idx] : 0.0f;
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; w++) s += v[w];
    return s;
}
// copy a [rows*cols] matrix from global memory into LDS (16-byte pieces when the size allows)
__device__ __forceinline__ void stage_matrix(float *dst, const float *__restrict__ src, int n, int tid, int nt) {
    int i = tid;
    for (; i + 7 * nt < n; i += 8 * nt) {              // eight loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[i + u * nt];
#pragma unroll
        for (int u = 0; u < 8; u++) dst[i + u * nt] = v[u];
    }
    for (; i < n; i += nt) dst[i] = src[i];
}

__global__ void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int r = idx / cols, c = idx - r * cols;
    out[(long long)c * rows + r] = in[idx];
}

// Osum[s] = sum_c C[c][s] (C_output_mat.sum(0), :232): one thread per state, rows read coalesced across the block
__global__ void column_sum_kernel(const float *__restrict__ C, float *__restrict__ Osum, int K, int S) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int c = 0;
    for (; c + 4 <= K; c += 4) {
        a0 += C[(long long)c * S + s]; a1 += C[(long long)(c + 1) * S + s];
        a2 += C[(long long)(c + 2) * S + s]; a3 += C[(long long)(c + 3) * S + s];
    }
    for (; c < K; c++) a0 += C[(long long)c * S + s];
    Osum[s] = (a0 + a1) + (a2 + a3);
}

// ---- step preparation in one launch: zero the gradient outputs, transpose the matrices, column-sum C -----------
// (fifteen-odd memsets and tiny kernels cost ~5 us of launch each on the stream)
constexpr int PREP_MAX_JOBS = 28;
struct PrepJob { const float *src; float *dst; int rows, cols, kind, e0; };   // kind 0 zero, 1 transpose, 2 column sum
struct PrepJobs { PrepJob j[PREP_MAX_JOBS]; int n, total; };

__global__ void __launch_bounds__(256)
train_prep_kernel(const PrepJobs jobs) {
    __shared__ float tile[32][33];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= jobs.total) return;
    int ji = 0;
    while (ji + 1 < jobs.n && (int)(blockIdx.x * blockDim.x) >= jobs.j[ji + 1].e0) ji++;   // jobs start on block boundaries
    const PrepJob &jb = jobs.j[ji];
    const int e = idx - jb.e0;
    if (jb.kind == 1) {
        // transpose through a 32x32 LDS tile: reads and writes both run along rows (a direct transpose writes 64
        // different cache lines per wavefront: 54 us for the 11 k x 250 table)
        const int tiles_c = (jb.cols + 31) >> 5, t = e >> 8, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
        if (r0 < jb.rows) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = r0 + ty + 8 * k, c = c0 + tx;
                tile[ty + 8 * k][tx] = (r < jb.rows && c < jb.cols) ? jb.src[(long long)r * jb.cols + c] : 0.0f;
            }
        }
        __syncthreads();
        if (r0 < jb.rows) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c = c0 + ty + 8 * k, r = r0 + tx;
                if (r < jb.rows && c < jb.cols) jb.dst[(long long)c * jb.rows + r] = tile[tx][ty + 8 * k];
            }
        }
        return;
    }
    const int n = jb.kind == 2 ? jb.cols : jb.rows * jb.cols;
    if (e >= n) return;
    if (jb.kind == 0) jb.dst[e] = 0.0f;
    else {                                                             // e = column; rows summed in four chains
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int c = 0;
        for (; c + 4 <= jb.rows; c += 4) {
            a0 += jb.src[(long long)c * jb.cols + e]; a1 += jb.src[(long long)(c + 1) * jb.cols + e];
            a2 += jb.src[(long long)(c + 2) * jb.cols + e]; a3 += jb.src[(long long)(c + 3) * jb.cols + e];
        }
        for (; c < jb.rows; c++) a0 += jb.src[(long long)c * jb.cols + e];
        jb.dst[e] = (a0 + a1) + (a2 + a3);
    }
}

// ---- forward chains with the stash ------------------------------------------------------------------------
// grid (ceil(B/2), 2): blockIdx.y = 0 forward, 1 backward chain; TR_NSEQ sequences per workgroup.  LDSW: the three
// matrices of this direction (2 S R + S S floats) are staged in LDS once; otherwise they are read through L2.
// LDS: [M1 | M2 | M3] f[2][S] tv[2][R] part[4][2][max(S,R)] part2[4][2][S]
template <bool LDSW, bool GATED, int VPS, int VPR, int NS>
__global__ void __launch_bounds__(TR_THREADS)
train_forward_kernel(const TrainParams p) {
    constexpr int VPT = VPS > VPR ? VPS : VPR;       // slots per thread: VPS cover 2 S states, VPR cover 2 R ranks
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nt = blockDim.x, nw = nt >> 6;
    const int b0 = blockIdx.x * NS, dir = blockIdx.y;
    const int S = p.S, R = p.R, SR = S > R ? S : R;
    float *wl = smem;
    const int SP = mv_pad(S), RP = mv_pad(R);                  // strides of the two sequences' vectors (zero pads)
    float *f = smem + (LDSW ? ((2 * S * R + S * S + 3) & ~3) : 0), *tv = f + NS * SP, *part = tv + NS * RP, *part2 = part + nw * NS * SR;
    const float *M1 = dir == 0 ? p.S1 : p.S2, *M2 = dir == 0 ? p.W : p.WT, *M3 = dir == 0 ? p.S2T : p.S1T;
    if (LDSW) {
        stage_matrix(wl, M1, S * R, tid, nt);
        stage_matrix(wl + S * R, M2, S * S, tid, nt);
        stage_matrix(wl + S * R + S * S, M3, R * S, tid, nt);
        M1 = wl; M2 = wl + S * R; M3 = wl + S * R + S * S;
    }
    int len[NS], maxlen = 0;
    for (int q = 0; q < NS; q++) {
        len[q] = b0 + q < p.B ? clamp_len(p.len[b0 + q], p.L) : 0;
        maxlen = len[q] > maxlen ? len[q] : maxlen;
    }
    // the token of every step, in step order, so that no step waits for an index load
    int *toks = (int *)(part2 + nw * NS * S);                  // [NS][L]
    // gated steps: raw state and v_t as matvec inputs, four more partial buffers
    float *hv = (float *)(toks + NS * p.L), *pg = hv + NS * SP;
    const int farnn = GATED ? p.farnn : 0;             // the ungated instantiation carries none of the gate state
    // through-L2 instantiation: the S x S matrices that still fit stay in LDS (W first, then the gates' Wss)
    float *ssl = smem + (((int)((GATED ? pg + 2 * nw * NS * S : hv) - smem) + 3) & ~3);  // offsets in floats: keeps the LDS address space
    const int nss = LDSW ? 0 : p.nss_f;
    const float *M2l = ssl, *G1l = ssl + S * S, *G2l = ssl + 2 * S * S;
    const bool l_m2 = nss >= 1, l_g1 = GATED && nss >= 2, l_g2 = GATED && nss >= 3;
    if (l_m2) stage_matrix(ssl, M2, S * S, tid, nt);
    if (l_g1 && farnn) stage_matrix(ssl + S * S, p.Wss1, S * S, tid, nt);
    if (l_g2 && farnn == 2) stage_matrix(ssl + 2 * S * S, p.Wss2, S * S, tid, nt);
    for (int e = tid; e < NS * p.L; e += nt) {
        const int q = e / p.L, i = e - q * p.L;
        toks[e] = i < len[q] ? clamp_tok(p.x[(long long)(b0 + q) * p.L + (dir == 0 ? i : len[q] - 1 - i)], p.V) : 0;
    }
    for (int e = tid; e < NS * SP + NS * RP; e += nt) f[e] = 0.0f;          // f | tv contiguous: pads stay zero
    if (farnn) for (int e = tid; e < NS * SP; e += nt) hv[e] = 0.0f;
    __syncthreads();
    for (int e = tid; e < NS * S; e += nt) {
        const int q = e / S, s = e - q * S;
        const float h = dir == 0 ? p.h0[s] : p.hT[s];
        if (b0 + q < p.B) (dir == 0 ? p.A : p.Bk)[(long long)(b0 + q) * (p.L + 1) * S + s] = h;
        f[q * SP + s] = dir == 0 ? h : h * p.Osum[s];                  // the backward chain masks its INPUT (:157-158)
        if (farnn) hv[q * SP + s] = h;
    }
    __syncthreads();
    // per-thread slots: which (sequence, state) and (sequence, rank) element this thread owns -- fixed for the whole
    // kernel, so the time loop has no divisions and steps its stash rows by S
    int sq[VPT], ss[VPT], rq[VPT], rr_[VPT];
    bool sv[VPT], rv[VPT];
    unsigned srow[VPT];                                // (32-bit element offsets: train_backward_kernel)
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const int e = tid + k * nt;
        // slot k of a kind that needs fewer slots than the other is dead at compile time (VPS, VPR): its registers vanish
        sv[k] = k < VPS && e < NS * S; sq[k] = sv[k] ? e / S : 0; ss[k] = sv[k] ? e - sq[k] * S : 0;
        rv[k] = k < VPR && e < NS * R; rq[k] = rv[k] ? e / R : 0; rr_[k] = rv[k] ? e - rq[k] * R : 0;
        srow[k] = (unsigned)(b0 + sq[k]) * (unsigned)((p.L + 1) * S) + (unsigned)ss[k];
    }
    float *stash_out = dir == 0 ? p.A : p.Bk;
    // v_t = Vgen[token] is fetched one step ahead into registers (VPT values per thread cover 2 R)
    float vcur[VPT], vnext[VPT], osum[VPT], hk[VPT], hin[VPT], zk[VPT], rk[VPT];
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        vcur[k] = (rv[k] && maxlen >= 1) ? p.Vgen[(unsigned)(toks[rq[k] * p.L] * R + rr_[k])] : 0.0f;
        osum[k] = sv[k] ? p.Osum[ss[k]] : 0.0f;
        hin[k] = sv[k] ? (dir == 0 ? p.h0[ss[k]] : p.hT[ss[k]]) : 0.0f;
        hk[k] = hin[k]; zk[k] = 1.0f; rk[k] = 1.0f;
    }
    float *ZG = dir == 0 ? p.ZGf : p.ZGb, *RG = dir == 0 ? p.RGf : p.RGb, *CD = dir == 0 ? p.CDf : p.CDb;
    float g1c[VPT], g2c[VPT], g1n[VPT], g2n[VPT];
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const bool ok = farnn && sv[k] && maxlen >= 1;
        const unsigned go = ok ? (unsigned)(toks[sq[k] * p.L] * S + ss[k]) : 0u;
        g1c[k] = ok ? p.GV1[go] : 0.0f;
        g2c[k] = (ok && farnn == 2) ? p.GV2[go] : 0.0f;
        g1n[k] = g2n[k] = 0.0f;
    }
    for (int t = 1; t <= maxlen; t++) {
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            vnext[k] = (rv[k] && t < maxlen) ? p.Vgen[(unsigned)(toks[rq[k] * p.L + t] * R + rr_[k])] : 0.0f;
            if (farnn) {
                const bool ok = sv[k] && t < maxlen;
                const unsigned go = ok ? (unsigned)(toks[sq[k] * p.L + t] * S + ss[k]) : 0u;
                g1n[k] = ok ? p.GV1[go] : 0.0f;
                g2n[k] = (ok && farnn == 2) ? p.GV2[go] : 0.0f;
            }
        }
        if (farnn) {
            // z = sigma(k (h Wss1 + v Wrs1 + bs1)), r likewise (:146-149); hbar = (1-r) h_init + r h (:150-151).  The input
            // halves v Wrs do not depend on the state: they are rows of GV = Vgen Wrs (one product per step for the whole
            // vocabulary, before the recurrences), fetched one step ahead like v itself -- the two largest products of
            // the gate phase (K = R) leave the sequential loop
            if (l_g1) matvec2_partial<true, NS>(pg, hv, SP, G1l, S, S, tid, nt);
            else matvec2_partial<false, NS>(pg, hv, SP, p.Wss1, S, S, tid, nt);
            if (farnn == 2) {
                if (l_g2) matvec2_partial<true, NS>(pg + nw * NS * S, hv, SP, G2l, S, S, tid, nt);
                else matvec2_partial<false, NS>(pg + nw * NS * S, hv, SP, p.Wss2, S, S, tid, nt);
            }
            wg_barrier_lds();
#pragma unroll
            for (int k = 0; k < VPT; k++) {
                if (sv[k]) {
                    const float az = part2_sum<NS>(pg, S, sq[k], ss[k], nw) + g1c[k] + p.bs1[ss[k]];
                    zk[k] = 1.0f / (1.0f + expf(-p.sig_k * az));
                    float hbar = hk[k];
                    if (farnn == 2) {
                        const float ar = part2_sum<NS>(pg + nw * NS * S, S, sq[k], ss[k], nw) + g2c[k] + p.bs2[ss[k]];
                        rk[k] = 1.0f / (1.0f + expf(-p.sig_k * ar));
                        hbar = (1.0f - rk[k]) * hin[k] + rk[k] * hk[k];
                    }
                    f[sq[k] * SP + ss[k]] = dir == 0 ? hbar : hbar * osum[k];
                }
            }
            wg_barrier_lds();
        }
        // rr = f . (S1 | S2) and the wildcard part f . (W | W^T): both depend on f only
        matvec2_partial<LDSW, NS>(part, f, SP, M1, S, R, tid, nt);
        if (l_m2) matvec2_partial<true, NS>(part2, f, SP, M2l, S, S, tid, nt);
        else matvec2_partial<LDSW, NS>(part2, f, SP, M2, S, S, tid, nt);
        wg_barrier_lds();
#pragma unroll
        for (int k = 0; k < VPT; k++)
            if (rv[k]) tv[rq[k] * RP + rr_[k]] = vcur[k] * part2_sum<NS>(part, R, rq[k], rr_[k], nw);     // temp = V_vec * _RR
        wg_barrier_lds();
        matvec2_partial<LDSW, NS>(part, tv, RP, M3, R, S, tid, nt);                              // temp . (S2^T | S1^T)
        wg_barrier_lds();
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            if (sv[k] && t <= len[sq[k]]) {
                const float pre = part2_sum<NS>(part, S, sq[k], ss[k], nw) + part2_sum<NS>(part2, S, sq[k], ss[k], nw);
                const unsigned row = srow[k] + (unsigned)(t * S);
                float h;
                if (dir == 0) { p.PRE[row] = pre; h = apply_nl(pre * osum[k], p.nl); }      // (:181)
                else          { h = apply_nl(pre, p.nl); }
                if (farnn) {                                                                 // (:193-196)
                    CD[row] = h; ZG[row] = zk[k]; RG[row] = rk[k];
                    h = (1.0f - zk[k]) * hk[k] + zk[k] * h;
                    hk[k] = h;
                    hv[sq[k] * SP + ss[k]] = h;
                } else {
                    f[sq[k] * SP + ss[k]] = dir == 0 ? h : h * osum[k];
                }
                stash_out[row] = h;
            }
        }
#pragma unroll
        for (int k = 0; k < VPT; k++) { vcur[k] = vnext[k]; g1c[k] = g1n[k]; g2c[k] = g2n[k]; }
        wg_barrier_lds();
    }
}

// ---- scores, cross-entropy, and the adjoints of alpha / beta ------------------------------------------------
// Persistent workgroups of 8 wavefronts; a wavefront takes positions round-robin and does one position at a time.
// CLDS: C_output_mat staged once per workgroup in LDS with row stride S+1 (lanes = labels read it conflict-free).
// LDS: [Cs[K][S+1]] then per wavefront ab[mv_pad(S)], sc[K], ds[K]
// PHASE 0: fused cross-entropy (scores, loss, adjoints).  CRF mode splits it around train_crf_kernel: PHASE 1 writes the
// emissions SC (and alpha*beta), PHASE 2 reads d loss / d emissions from DS and produces the adjoints.
template <bool CLDS, int PHASE>
__global__ void __launch_bounds__(512)
train_loss_kernel(const TrainParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int S = p.S, K = p.K, SP = mv_pad(S), ldc = CLDS ? S + 1 : S;
    float *Cs = smem;
    float *ab = smem + (CLDS ? ((K * (S + 1) + 3) & ~3) : 0) + w * (SP + 2 * K), *sc = ab + SP, *ds = sc + K;
    if (CLDS) {
        for (int e = tid; e < K * S; e += blockDim.x) { const int c = e / S, s_ = e - c * S; Cs[c * (S + 1) + s_] = p.C[e]; }
        __syncthreads();
    }
    const float *Cm = CLDS ? Cs : p.C;
    float loss_acc = 0.0f;
    for (long long pos = (long long)blockIdx.x * nw + w; pos < (long long)p.B * p.L; pos += (long long)gridDim.x * nw) {
        const int b = (int)(pos / p.L), i = (int)(pos - (long long)b * p.L);
        const int len = clamp_len(p.len[b], p.L);
        if (i >= len) {
            if (lane == 0) p.tags[pos] = -1;
            continue;
        }
        const float *al = p.A + ((long long)b * (p.L + 1) + i + 1) * S;                  // h0_forward_score[:, i+1]
        const float *be = p.Bk + ((long long)b * (p.L + 1) + (len - 1 - i)) * S;         // reverse(.., lengths+1)[:, i+1]
        if (PHASE != 2) {
            for (int s = lane; s < S; s += WAVE) {
                const float v = al[s] * be[s];
                ab[s] = v;
                p.AB[pos * S + s] = v;
            }
            for (int c = lane; c < K; c += WAVE) {                                        // get_final_score (:200-203)
                const float *cr = Cm + (long long)c * ldc;
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                int s = 0;
                for (; s + 8 <= S; s += 8) {
                    float cv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) cv[u] = cr[s + u];
                    const v4f x0 = *(const v4f *)(ab + s), x1 = *(const v4f *)(ab + s + 4);
                    a0 = fmaf(x0[0], cv[0], a0); a1 = fmaf(x0[1], cv[1], a1); a2 = fmaf(x0[2], cv[2], a2); a3 = fmaf(x0[3], cv[3], a3);
                    a0 = fmaf(x1[0], cv[4], a0); a1 = fmaf(x1[1], cv[5], a1); a2 = fmaf(x1[2], cv[6], a2); a3 = fmaf(x1[3], cv[7], a3);
                }
                for (; s < S; s++) a0 = fmaf(ab[s], cr[s], a0);
                sc[c] = (a0 + a1) + (a2 + a3);
            }
            if (p.P) {                                                                     // priority layer
                for (int d = lane; d < K; d += WAVE) {
                    float a = 0.0f;
                    for (int c = 0; c < K; c++) a = fmaf(sc[c], p.P[(long long)c * K + d], a);
                    ds[d] = a;
                }
                for (int d = lane; d < K; d += WAVE) sc[d] = ds[d];
            }
        }
        if (PHASE == 1) {
            for (int c = lane; c < K; c += WAVE) p.SC[pos * K + c] = sc[c];
            continue;
        }
        if (PHASE == 0) {
            // softmax cross-entropy (mean over the batch's valid tokens) and the prediction (decode, argmax branch)
            float mx = -INFINITY;
            for (int c = lane; c < K; c += WAVE) mx = fmaxf(mx, sc[c]);
            for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, WAVE));
            float se = 0.0f;
            for (int c = lane; c < K; c += WAVE) se += expf(sc[c] - mx);
            for (int o = 32; o; o >>= 1) se += __shfl_xor(se, o, WAVE);
            // a label outside 0..K-1 (torch's CrossEntropyLoss raises on it): counted as label 0 in the loss AND in the
            // gradient, and reported through the sticky flag (the next call returns FARNN_EINVAL)
            const long long lab64 = p.labels[pos];
            const int lab = (lab64 < 0 || lab64 >= K) ? 0 : (int)lab64;
            if (lane == 0 && lab64 != lab) atomicOr(p.err, 1);
            const float lse = mx + logf(se);
            loss_acc += lse - sc[lab];
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
        } else {
            for (int c = lane; c < K; c += WAVE) ds[c] = p.DS[pos * K + c];                 // d loss / d emissions (CRF)
        }
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
            float d0 = 0.0f, d1 = 0.0f;
            int c = 0;
            for (; c + 8 <= K; c += 8) {
                float cv[8], dv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { cv[u] = Cm[(long long)(c + u) * ldc + s]; dv[u] = ds[c + u]; }
#pragma unroll
                for (int u = 0; u < 8; u += 2) { d0 = fmaf(dv[u], cv[u], d0); d1 = fmaf(dv[u + 1], cv[u + 1], d1); }
            }
            for (; c < K; c++) d0 = fmaf(ds[c], Cm[(long long)c * ldc + s], d0);           // d(alpha*beta)
            const float d = d0 + d1;
            ga[s] = d * be[s];
            gb[s] = d * al[s];
        }
    }
    if (lane == 0 && loss_acc != 0.0f) atomicAdd(p.loss, loss_acc * p.inv_tokens);
}

// ---- CRF negative log-likelihood on the emissions (reference baselines/crf.py:48-99, 202-260) -------------------
// One workgroup per sequence.  loss += log Z - score(gold path) (a SUM over the batch, :250-260); d loss / d emissions
// = posterior marginals - gold one-hot (into DS); d loss / d transitions = expected - gold transition counts (per
// sequence into dtrans_part, reduced afterwards); the decoded tags are the Viterbi path of the emissions with
// column K-3 clamped (model_decompose.py:351-356).  Tags K-2 / K-1 are START / STOP.
// Scaled (exp-domain) messages: with etr = exp(tr) kept in LDS, a step of either recursion is a plain dot product,
//   alpha_t[j] = amax + log(sum_i exp(alpha_{t-1}[i] - amax) etr[i][j]) + f_t[j],   amax = max_i alpha_{t-1}[i],
// and the expected transition counts are products of the stored factors -- no exp in any K x K loop (the first
// version evaluated expf K^2 times per step: 1.85 ms for 256 sequences).
// LDS: tr, etr, ex [K][K+1]; al[L][K] log-messages; ea[L][K] = exp(al - amax_t); am[L]; bt[2][K]; eb[K]; red[8];
//      vit[2][K]; fl[L][K] emissions; bp[L][K] bytes
template <bool BIG>
__global__ void __launch_bounds__(256)
train_crf_kernel(const TrainParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
    const int K = p.K, K1 = K + 1, START = K - 2, STOP = K - 1;
    const int n = clamp_len(p.len[b], p.L);
    // BIG (tag sets whose three tables + messages exceed the 160 KiB of LDS, e.g. K = 130 at L = 64): only
    // exp(transitions) and the messages stay in LDS; the transitions are read from global memory (L2-resident, the
    // Viterbi half and the gold score use them), the expected counts accumulate in this sequence's slice of
    // dtrans_part (every element has one owning thread) and the emissions are read where the loss kernel left them.
    const int TS = BIG ? K : K1;                                               // row stride of tr / ex
    float *etr = smem, *lds_tr = etr + K * K1, *lds_ex = lds_tr + (BIG ? 0 : K * K1);
    float *al = lds_ex + (BIG ? 0 : K * K1), *ea = al + (long long)p.L * K;
    float *am = ea + (long long)p.L * K, *bt = am + p.L, *eb = bt + 2 * K, *red = eb + K, *vit = red + 8;
    float *fl = vit + 2 * K;                                                   // [L][K] this sequence's emissions (!BIG)
    unsigned char *bp = (unsigned char *)(fl + (BIG ? 0 : (long long)p.L * K));
    float *dpart = p.dtrans_part + (long long)b * K * K;
    const float *tr;
    float *ex;
    if constexpr (BIG) { tr = p.trans; ex = dpart; } else { tr = lds_tr; ex = lds_ex; }
    for (int e = tid; e < K * K; e += nt) {
        const int o = (e / K) * K1 + e % K;
        const float t = p.trans[e];
        etr[o] = __expf(t);
        if constexpr (BIG) { dpart[e] = 0.0f; } else { lds_tr[o] = t; lds_ex[o] = 0.0f; }
    }
    __syncthreads();
    if (n == 0) {
        if constexpr (!BIG)
            for (int e = tid; e < K * K; e += nt) dpart[e] = 0.0f;
        return;
    }
    const float *F;
    if constexpr (BIG) {
        F = p.SC + (long long)b * p.L * K;
    } else {
        const float *Fg = p.SC + (long long)b * p.L * K;
        for (int e = tid; e < n * K; e += nt) fl[e] = Fg[e];
        F = fl;
    }
    __syncthreads();
    const int64_t *y = p.labels + (long long)b * p.L;
    // forward messages and the Viterbi recursion on the clamped emissions (association as in :123,:145)
    // red[4..7]: per-wavefront maxima of the newest forward (later backward) message: the next step's scale comes from
    // four broadcast reads instead of a pass over the K messages by every thread
    const int wv = tid >> 6, lane_ = tid & 63;
    {
        float vmax = -INFINITY;
        for (int j = tid; j < K; j += nt) {
            const float f0 = F[j], a0 = f0 + tr[START * TS + j];
            al[j] = a0;
            vit[j] = (j == K - 3 ? fminf(f0, p.threshold) : f0) + tr[START * TS + j];
            vmax = fmaxf(vmax, a0);
        }
        for (int o = 32; o; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, WAVE));
        if (lane_ == 0) red[4 + wv] = vmax;
    }
    __syncthreads();
    for (int t = 0; t < n; t++) {
        // scale of step t: amax_t (from the wavefront maxima) and ea_t = exp(al_t - amax_t)
        const float *at = al + (long long)t * K;
        const float mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        for (int j = tid; j < K; j += nt) ea[(long long)t * K + j] = __expf(at[j] - mx);
        if (tid == 0) am[t] = mx;
        __syncthreads();
        if (t + 1 < n) {
            // threads 0..127 do the sum recursion, 128..255 the max recursion (Viterbi): half the work per thread.
            // Four independent chains per loop: a single dependent chain over LDS reads runs at ~105 cycles per element,
            // four chains at ~12 (scripts/probe/lds_rate.hip)
            const float *et = ea + (long long)t * K, *vp = vit + (t & 1) * K;
            const int half = tid >> 7, lt = tid & 127;
            float vmax = -INFINITY;
            if (half == 0) {
                for (int j = lt; j < K; j += 128) {
                    float s4[4] = {0.f, 0.f, 0.f, 0.f};
                    int i = 0;
#pragma unroll 4
                    for (; i + 4 <= K; i += 4) {
#pragma unroll
                        for (int u = 0; u < 4; u++) s4[u] = fmaf(et[i + u], etr[(i + u) * K1 + j], s4[u]);
                    }
                    for (; i < K; i++) s4[0] = fmaf(et[i], etr[i * K1 + j], s4[0]);
                    const float an = mx + __logf((s4[0] + s4[1]) + (s4[2] + s4[3])) + F[(long long)(t + 1) * K + j];
                    al[(long long)(t + 1) * K + j] = an;
                    vmax = fmaxf(vmax, an);
                }
            } else {
                for (int j = lt; j < K; j += 128) {
                    const float ft = F[(long long)(t + 1) * K + j], fc = j == K - 3 ? fminf(ft, p.threshold) : ft;
                    float b4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    int i4[4] = {0, 0, 0, 0};
                    int i = 0;
#pragma unroll 4
                    for (; i + 4 <= K; i += 4) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const float cand = (fc + tr[(i + u) * TS + j]) + vp[i + u];
                            if (cand > b4[u]) { b4[u] = cand; i4[u] = i + u; }
                        }
                    }
                    for (; i < K; i++) {
                        const float cand = (fc + tr[i * TS + j]) + vp[i];
                        if (cand > b4[0]) { b4[0] = cand; i4[0] = i; }
                    }
                    float bv = b4[0];
                    int bi = i4[0];
#pragma unroll
                    for (int u = 1; u < 4; u++)                              // first maximum over i, like torch.max (:149)
                        if (b4[u] > bv || (b4[u] == bv && i4[u] < bi)) { bv = b4[u]; bi = i4[u]; }
                    vit[((t + 1) & 1) * K + j] = bv;
                    bp[(long long)(t + 1) * K + j] = (unsigned char)bi;
                }
            }
            for (int o = 32; o; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, WAVE));
            if (lane_ == 0) red[4 + wv] = vmax;                               // -inf from the Viterbi wavefronts
            __syncthreads();
        }
    }
    // log Z, the gold score and the Viterbi backtrace (one thread: n steps)
    if (tid == 0) {
        const float *et = ea + (long long)(n - 1) * K, *vp = vit + ((n - 1) & 1) * K;
        float se = 0.0f;
        for (int i = 0; i < K; i++) se = fmaf(et[i], etr[i * K1 + STOP], se);
        const float logZ = am[n - 1] + logf(se);
        int prev = START;
        float gold = 0.0f;
        for (int t = 0; t < n; t++) {
            int yt = (int)y[t];
            if (y[t] < 0 || y[t] >= K) { yt = 0; atomicOr(p.err, 1); }         // same clamp as the marginals below
            gold += F[(long long)t * K + yt] + tr[prev * TS + yt];
            ex[prev * TS + yt] -= 1.0f;                                         // gold transition counts
            prev = yt;
        }
        gold += tr[prev * TS + STOP];
        ex[prev * TS + STOP] -= 1.0f;
        red[0] = logZ;
        atomicAdd(p.loss, logZ - gold);
        float bv = -INFINITY; int ptr = 0;
        for (int i = 0; i < K; i++) { const float c = vp[i] + tr[i * TS + STOP]; if (c > bv) { bv = c; ptr = i; } }
        for (int t = n - 1; t >= 0; t--) {
            p.tags[(long long)b * p.L + t] = ptr == K - 3 ? p.o_idx : ptr;
            if (t > 0) ptr = bp[(long long)t * K + ptr];
        }
    }
    for (int t = n + tid; t < p.L; t += nt) p.tags[(long long)b * p.L + t] = -1;
    {
        float vmax = -INFINITY;
        for (int i = tid; i < K; i += nt) {
            const float bv = tr[i * TS + STOP];
            bt[((n - 1) & 1) * K + i] = bv;                                      // backward message at the last token
            vmax = fmaxf(vmax, F[(long long)(n - 1) * K + i] + bv);
        }
        for (int o = 32; o; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, WAVE));
        __syncthreads();                                                         // the forward loop's last reads of red[4..7]
        if (lane_ == 0) red[4 + wv] = vmax;
    }
    __syncthreads();
    const float logZ = red[0];
    // expected transition counts: thread (row group r0, column cj) owns rows r0, r0 + rstep, ... of column cj.  When that
    // is at most CRF_XR rows, its counts and its exp(transition) entries live in registers for the whole backward pass
    // (the LDS read-modify-write form costs four LDS instructions per element and step)
    constexpr int CRF_XR = 32;
    const int xcj = tid % K1, xr0 = tid / K1, xrstep = nt / K1 > 0 ? nt / K1 : 1;
    const bool xreg = K1 <= nt && (K + xrstep - 1) / xrstep <= CRF_XR;
    const bool xmine = xreg && xcj < K && xr0 < xrstep;
    float exr[CRF_XR], etrr[CRF_XR];
#pragma unroll
    for (int u = 0; u < CRF_XR; u++) {
        const int i = xr0 + u * xrstep;
        exr[u] = 0.0f;
        etrr[u] = (xmine && i < K) ? etr[i * K1 + xcj] : 0.0f;
    }
    // backward messages, marginals and expected transition counts
    for (int t = n - 1; t >= 0; t--) {
        const float *bc = bt + (t & 1) * K;
        float *bn = bt + ((t + 1) & 1) * K;                                     // becomes beta_{t-1}
        const float *at = al + (long long)t * K;
        // scale of the backward side at step t: bmax over f_t + beta_t, eb = exp(f_t + beta_t - bmax)
        const float bmx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        for (int j = tid; j < K; j += nt) {
            eb[j] = __expf(F[(long long)t * K + j] + bc[j] - bmx);
            const float m = __expf(at[j] + bc[j] - logZ);
            int yt = (int)y[t];
            yt = yt < 0 || yt >= K ? 0 : yt;
            p.DS[((long long)b * p.L + t) * K + j] = m - (j == yt ? 1.0f : 0.0f);
            if (t == 0) ex[START * TS + j] += m;
            if (t == n - 1) ex[j * TS + STOP] += m;
        }
        __syncthreads();                                                        // eb ready; START row / STOP column settled
        if (t > 0) {
            const float *ep = ea + (long long)(t - 1) * K;
            const float scale = __expf(am[t - 1] + bmx - logZ);                 // xi_{t-1}(i,j) = ea[i] etr[i][j] eb[j] scale
            // thread (row group r0, column j): rows r0, r0 + rstep, ... -- no divisions, all reads of a round independent
            if (xreg) {
                if (xmine) {
                    const float se = scale * eb[xcj];
#pragma unroll
                    for (int u = 0; u < CRF_XR; u++) {
                        const int i = xr0 + u * xrstep;
                        exr[u] = fmaf(ep[i < K ? i : 0] * se, etrr[u], exr[u]);          // etrr is 0 past the last row
                    }
                }
            } else if (K1 <= nt) {
                const int cj = tid % K1, r0 = tid / K1, rstep = nt / K1;        // K1 columns per row in LDS (the pad column is skipped)
                if (cj < K && r0 < rstep) {
                    const float ebj = eb[cj];
#pragma unroll 4
                    for (int i = r0; i < K; i += rstep)
                        ex[i * TS + cj] = fmaf(ep[i] * scale, etr[i * K1 + cj] * ebj, ex[i * TS + cj]);
                }
            } else {
                for (int e = tid; e < K * K; e += nt) {
                    const int i = e / K, j = e - i * K;
                    ex[i * TS + j] = fmaf(ep[i] * scale, etr[i * K1 + j] * eb[j], ex[i * TS + j]);
                }
            }
            float vmax = -INFINITY;
            for (int i = tid; i < K; i += nt) {                                 // beta_{t-1}[i] = bmax + log sum_j etr[i][j] eb[j]
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
                int j = 0;
#pragma unroll 4
                for (; j + 4 <= K; j += 4) {
                    s0 = fmaf(etr[i * K1 + j], eb[j], s0); s1 = fmaf(etr[i * K1 + j + 1], eb[j + 1], s1);
                    s2 = fmaf(etr[i * K1 + j + 2], eb[j + 2], s2); s3 = fmaf(etr[i * K1 + j + 3], eb[j + 3], s3);
                }
                for (; j < K; j++) s0 = fmaf(etr[i * K1 + j], eb[j], s0);
                const float se = (s0 + s1) + (s2 + s3);
                const float bnew = bmx + __logf(se);
                bn[i] = bnew;
                vmax = fmaxf(vmax, F[(long long)(t - 1) * K + i] + bnew);
            }
            for (int o = 32; o; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, WAVE));
            if (lane_ == 0) red[4 + wv] = vmax;                                   // read after the barrier below
        }
        __syncthreads();
    }
    if (xmine) {
#pragma unroll
        for (int u = 0; u < CRF_XR; u++) {
            const int i = xr0 + u * xrstep;
            if (i < K) ex[i * TS + xcj] += exr[u];                              // each element has exactly one owner
        }
    }
    __syncthreads();
    for (int e = tid; e < K * K; e += nt) dpart[e] = ex[(e / K) * TS + e % K];
}

// LDS bytes of train_crf_kernel<BIG> (its carve, above)
inline size_t train_crf_lds_bytes(size_t K, size_t L, bool big) {
    const size_t K1 = K + 1;
    const size_t floats = (big ? 1 : 3) * K * K1 + (big ? 2 : 3) * L * K + L + 5 * K + 8;
    return floats * sizeof(float) + ((L * K + 3) & ~(size_t)3);
}

// dtrans[e] = sum_b dtrans_part[b][e]
__global__ void crf_reduce_kernel(const float *__restrict__ part, float *dtrans, int B, int KK) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= KK) return;
    float s0 = 0.0f, s1 = 0.0f;
    int b = 0;
    for (; b + 1 < B; b += 2) { s0 += part[(long long)b * KK + e]; s1 += part[(long long)(b + 1) * KK + e]; }
    if (b < B) s0 += part[(long long)b * KK + e];
    dtrans[e] = s0 + s1;
}

// ---- back-propagation through time ----------------------------------------------------------------------------
// grid (ceil(B/2), 2), TR_NSEQ sequences per workgroup.  LDSW: the four matrices of this direction
// (3 S R + S S floats) live in LDS.  The forward chain's pre-activation is read from the stash (PRE), not recomputed.
// LDS: [Ma | Mb | Mc | Md] z fp [2][mv_pad(S)], d1 [2][mv_pad(R)], pa pb [8][2][max(S,R)], pc [8][2][S], toks [2][L]
template <bool LDSW, bool GATED, int VPS, int VPR, int NS>
__global__ void __launch_bounds__(TR_THREADS)
train_backward_kernel(const TrainParams p) {
    constexpr int VPT = VPS > VPR ? VPS : VPR;       // slots per thread: VPS cover 2 S states, VPR cover 2 R ranks
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nt = blockDim.x, nw = nt >> 6;
    const int b0 = blockIdx.x * NS, dir = blockIdx.y;
    const int S = p.S, R = p.R, SR = S > R ? S : R;
    float *wl = smem;
    const int SP = mv_pad(S), RP = mv_pad(R);                  // strides of the matvec inputs z, fp, d1 (zero pads)
    float *z = smem + (LDSW ? ((3 * S * R + S * S + 3) & ~3) : 0), *fp = z + NS * SP, *d1 = fp + NS * SP;
    float *pa = d1 + NS * RP, *pb = pa + nw * NS * SR, *pc = pb + nw * NS * SR;
    // rr = fp . Ma, u = z . Mb, d fp = z . Mc + d1 . Md
    const float *Ma = dir == 0 ? p.S1 : p.S2, *Mb = dir == 0 ? p.S2 : p.S1, *Mc = dir == 0 ? p.WT : p.W,
                *Md = dir == 0 ? p.S1T : p.S2T;
    if (LDSW) {
        stage_matrix(wl, Ma, S * R, tid, nt);
        stage_matrix(wl + S * R, Mb, S * R, tid, nt);
        stage_matrix(wl + 2 * S * R, Mc, S * S, tid, nt);
        stage_matrix(wl + 2 * S * R + S * S, Md, R * S, tid, nt);
        Ma = wl; Mb = wl + S * R; Mc = wl + 2 * S * R; Md = wl + 2 * S * R + S * S;
    }
    int len[NS], maxlen = 0;
    for (int q = 0; q < NS; q++) {
        len[q] = b0 + q < p.B ? clamp_len(p.len[b0 + q], p.L) : 0;
        maxlen = len[q] > maxlen ? len[q] : maxlen;
    }
    int *toks = (int *)(pc + nw * NS * S);                     // [NS][L] tokens in step order
    // gated steps: the gate pre-activation adjoints as matvec inputs and a fourth partial buffer
    float *dazv = (float *)(toks + NS * p.L), *darv = dazv + NS * SP;
    const int farnn = GATED ? p.farnn : 0;             // the ungated instantiation carries none of the gate state
    // through-L2 instantiation: the S x S matrices that still fit stay in LDS (W^T|W first, then the gates' Wss^T)
    float *ssl = smem + (((int)((GATED ? darv + NS * SP : dazv) - smem) + 3) & ~3);
    const int nss = LDSW ? 0 : p.nss_b;
    const float *Mcl = ssl, *G1l = ssl + S * S, *G2l = ssl + 2 * S * S;
    const bool l_mc = nss >= 1, l_g1 = GATED && nss >= 2, l_g2 = GATED && nss >= 3;
    if (l_mc) stage_matrix(ssl, Mc, S * S, tid, nt);
    if (l_g1 && farnn) stage_matrix(ssl + S * S, p.Wss1T, S * S, tid, nt);
    if (l_g2 && farnn == 2) stage_matrix(ssl + 2 * S * S, p.Wss2T, S * S, tid, nt);
    for (int e = tid; e < NS * p.L; e += nt) {
        const int q = e / p.L, i = e - q * p.L;
        toks[e] = i < len[q] ? clamp_tok(p.x[(long long)(b0 + q) * p.L + (dir == 0 ? i : len[q] - 1 - i)], p.V) : 0;
    }
    for (int e = tid; e < 2 * NS * SP + NS * RP; e += nt) z[e] = 0.0f;          // z fp d1 contiguous: pads stay zero
    if (farnn) for (int e = tid; e < 2 * NS * SP; e += nt) dazv[e] = 0.0f;       // dazv | darv contiguous
    __syncthreads();
    // per-thread slots, fixed for the whole kernel (no divisions in the time loop); g, y and dOsum of a slot are only
    // ever touched by its owner, so they live in registers
    int sq[VPT], ss[VPT], rq[VPT], rr_[VPT];
    bool sv[VPT], rv[VPT];
    // (32-bit element offsets against the scalar base pointers -- farnn_train_create bounds the arrays: 64-bit rows made every
    // stash array of a slot its own 64-bit pointer over the loop, and the gated two-slot kernels spilled 28 registers)
    unsigned srow[VPT], rrow[VPT];
    float osum[VPT], gacc[VPT], dOacc[VPT], yk[VPT];
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const int e = tid + k * nt;
        // slot k of a kind that needs fewer slots than the other is dead at compile time (VPS, VPR): its registers vanish
        sv[k] = k < VPS && e < NS * S; sq[k] = sv[k] ? e / S : 0; ss[k] = sv[k] ? e - sq[k] * S : 0;
        rv[k] = k < VPR && e < NS * R; rq[k] = rv[k] ? e / R : 0; rr_[k] = rv[k] ? e - rq[k] * R : 0;
        sv[k] = sv[k] && b0 + sq[k] < p.B;
        rv[k] = rv[k] && b0 + rq[k] < p.B;
        srow[k] = (unsigned)(b0 + (sv[k] ? sq[k] : 0)) * (unsigned)((p.L + 1) * S) + (unsigned)ss[k];
        rrow[k] = (unsigned)(b0 + (rv[k] ? rq[k] : 0)) * (unsigned)((p.L + 1) * R) + (unsigned)rr_[k];
        osum[k] = sv[k] ? p.Osum[ss[k]] : 0.0f;
        gacc[k] = 0.0f; dOacc[k] = 0.0f; yk[k] = 0.0f;
    }
    float hin[VPT], dhin[VPT], zc[VPT], zn[VPT], rc[VPT], rn[VPT], cc[VPT], cn[VPT], dhk[VPT];
    const float *ZG = dir == 0 ? p.ZGf : p.ZGb, *RG = dir == 0 ? p.RGf : p.RGb, *CD = dir == 0 ? p.CDf : p.CDb;
    float *DAZ = dir == 0 ? p.DAZf : p.DAZb, *DAR = dir == 0 ? p.DARf : p.DARb;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        hin[k] = (farnn && sv[k]) ? (dir == 0 ? p.h0[ss[k]] : p.hT[ss[k]]) : 0.0f;
        dhin[k] = 0.0f; zc[k] = zn[k] = rc[k] = rn[k] = 1.0f; cc[k] = cn[k] = 0.0f; dhk[k] = 0.0f;
    }
    // everything a step reads from the stash is fetched one step ahead into registers (the rows of a sequence
    // beyond its length are zero, so the reads need no guard): h_t, h_{t-1}, dL/dh_t from the scoring, pre_t, v_t
    const float *stash_base = dir == 0 ? p.A : p.Bk, *G_base = dir == 0 ? p.GA : p.GB;
    float *Zo = dir == 0 ? p.Zf : p.Zb, *D1o = dir == 0 ? p.D1f : p.D1b, *To = dir == 0 ? p.Tf : p.Tb;
    float hcur[VPT], hprev[VPT], hpp[VPT], gs[VPT], gsn[VPT], pr[VPT], prn[VPT], vcur[VPT], vnext[VPT];
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const bool ok = sv[k] && maxlen >= 1;
        const unsigned row = srow[k] + (unsigned)(maxlen * S);
        hcur[k] = ok ? stash_base[row] : 0.0f;
        hprev[k] = ok ? stash_base[row - S] : 0.0f;
        gs[k] = ok ? G_base[row] : 0.0f;
        pr[k] = (ok && dir == 0) ? p.PRE[row] : 0.0f;
        if (farnn) { zc[k] = ok ? ZG[row] : 1.0f; rc[k] = ok ? RG[row] : 1.0f; cc[k] = ok ? CD[row] : 0.0f; }
        vcur[k] = (rv[k] && maxlen >= 1) ? p.Vgen[(unsigned)(toks[rq[k] * p.L + (maxlen - 1 < len[rq[k]] ? maxlen - 1 : 0)] * R + rr_[k])] : 0.0f;
    }
    for (int t = maxlen; t >= 1; t--) {
#pragma unroll
        for (int k = 0; k < VPT; k++) {                        // prefetch for step t-1
            const bool ok = sv[k] && t >= 2;
            const unsigned row = srow[k] + (unsigned)((t - 1) * S);
            hpp[k] = ok ? stash_base[row - S] : 0.0f;
            gsn[k] = ok ? G_base[row] : 0.0f;
            prn[k] = (ok && dir == 0) ? p.PRE[row] : 0.0f;
            if (farnn) { zn[k] = ok ? ZG[row] : 1.0f; rn[k] = ok ? RG[row] : 1.0f; cn[k] = ok ? CD[row] : 0.0f; }
            vnext[k] = (rv[k] && t >= 2) ? p.Vgen[(unsigned)(toks[rq[k] * p.L + (t - 2 < len[rq[k]] ? t - 2 : 0)] * R + rr_[k])] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            if (sv[k]) {
                const int li = sq[k] * SP + ss[k];
                if (t <= len[sq[k]]) {
                    const unsigned row = srow[k] + (unsigned)(t * S);
                    const float gt = gacc[k] + gs[k];
                    float yy, hp = hprev[k];
                    if (farnn) {                                      // h_t = (1-z) h_{t-1} + z cand (:193-196)
                        dhk[k] = gt * (1.0f - zc[k]);                  // direct path to h_{t-1}
                        yy = (gt * zc[k]) * nl_grad_from_output(cc[k], p.nl);
                        // d z pre-activation: dz = gt (cand - h_{t-1}); sigma'(k a) = k z (1-z)
                        const float daz = gt * (cc[k] - hp) * p.sig_k * zc[k] * (1.0f - zc[k]);
                        dazv[li] = daz;
                        DAZ[row] = daz;
                        atomicAdd(p.dGV1 + (unsigned)(toks[sq[k] * p.L + t - 1] * S + ss[k]), daz);   // az = h Wss1 + GV1[token] + bs1
                        if (farnn == 2) hp = (1.0f - rc[k]) * hin[k] + rc[k] * hp;          // hbar (:150-151)
                    } else {
                        yy = gt * nl_grad_from_output(hcur[k], p.nl);
                    }
                    if (dir == 0) {                                   // mask on the OUTPUT of the step
                        const float zz = yy * osum[k];
                        z[li] = zz; fp[li] = hp;
                        dOacc[k] = fmaf(yy, pr[k], dOacc[k]);                         // d Osum += y * pre_t
                        Zo[row] = zz;
                        if (farnn) p.HBARf[row] = hp;
                    } else {                                          // mask on the INPUT: fp = bbar, y keeps b_{t-1}
                        const float bb = hp * osum[k];
                        z[li] = yy; fp[li] = bb; yk[k] = hp;
                        Zo[row] = yy;
                        p.BBAR[row] = bb;
                    }
                } else {
                    z[li] = 0.0f; fp[li] = 0.0f;
                    if (farnn) { dazv[li] = 0.0f; darv[li] = 0.0f; }
                }
            }
        }
        wg_barrier_lds();
        matvec2_partial<LDSW, NS>(pa, fp, SP, Ma, S, R, tid, nt);            // rr  = fp . (S1 | S2)
        matvec2_partial<LDSW, NS>(pb, z, SP, Mb, S, R, tid, nt);             // u   = z . (S2 | S1)
        if (l_mc) matvec2_partial<true, NS>(pc, z, SP, Mcl, S, S, tid, nt);   // d fp through the wildcard matrix
        else matvec2_partial<LDSW, NS>(pc, z, SP, Mc, S, S, tid, nt);
        wg_barrier_lds();
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            if (rv[k]) {
                float dd = 0.0f;
                if (t <= len[rq[k]]) {
                    const unsigned row = rrow[k] + (unsigned)(t * R);
                    const float rvv = part2_sum<NS>(pa, R, rq[k], rr_[k], nw), uv = part2_sum<NS>(pb, R, rq[k], rr_[k], nw);
                    const float vv = vcur[k];
                    dd = uv * vv;
                    D1o[row] = dd;
                    To[row] = vv * rvv;
                    atomicAdd(p.dVgen + (unsigned)(toks[rq[k] * p.L + t - 1] * R + rr_[k]), uv * rvv);   // d v_t = u * rr
                }
                d1[rq[k] * RP + rr_[k]] = dd;
            }
        }
        wg_barrier_lds();
        matvec2_partial<LDSW, NS>(pa, d1, RP, Md, R, S, tid, nt);            // d fp through the language factors
        wg_barrier_lds();
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            if (sv[k] && t <= len[sq[k]]) {
                const float dfp = part2_sum<NS>(pc, S, sq[k], ss[k], nw) + part2_sum<NS>(pa, S, sq[k], ss[k], nw);
                float dhb;                                             // adjoint of the (unmasked) chain input
                if (dir == 0) dhb = dfp;
                else { dOacc[k] = fmaf(dfp, yk[k], dOacc[k]); dhb = dfp * osum[k]; }       // d Osum += d bbar * (unmasked input)
                if (!farnn) gacc[k] = dhb;
                else {
                    float dar = 0.0f;
                    if (farnn == 2) {                                  // hbar = (1-r) h_init + r h_{t-1}
                        dar = dhb * (hprev[k] - hin[k]) * p.sig_k * rc[k] * (1.0f - rc[k]);
                        dhin[k] = fmaf(dhb, 1.0f - rc[k], dhin[k]);
                        dhk[k] = fmaf(dhb, rc[k], dhk[k]);
                        DAR[srow[k] + (unsigned)(t * S)] = dar;
                        atomicAdd(p.dGV2 + (unsigned)(toks[sq[k] * p.L + t - 1] * S + ss[k]), dar);
                    } else {
                        dhk[k] += dhb;
                    }
                    darv[sq[k] * SP + ss[k]] = dar;
                }
            }
        }
        if (farnn) {
            // the gates read the raw h_{t-1}: d h_{t-1} += daz Wss1^T + dar Wss2^T.  Their input halves were hoisted
            // (GV = Vgen Wrs): the adjoints daz, dar go to the word's rows of dGV, and d Vgen += dGV Wrs^T, d Wrs = Vgen^T dGV
            // are products over the vocabulary after the loop
            wg_barrier_lds();
            if (l_g1) matvec2_partial<true, NS>(pa, dazv, SP, G1l, S, S, tid, nt);
            else matvec2_partial<false, NS>(pa, dazv, SP, p.Wss1T, S, S, tid, nt);
            if (farnn == 2) {
                if (l_g2) matvec2_partial<true, NS>(pc, darv, SP, G2l, S, S, tid, nt);
                else matvec2_partial<false, NS>(pc, darv, SP, p.Wss2T, S, S, tid, nt);
            }
            wg_barrier_lds();
#pragma unroll
            for (int k = 0; k < VPT; k++) {
                if (sv[k] && t <= len[sq[k]]) {
                    float dh = dhk[k] + part2_sum<NS>(pa, S, sq[k], ss[k], nw);
                    if (farnn == 2) dh += part2_sum<NS>(pc, S, sq[k], ss[k], nw);
                    gacc[k] = dh;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            hcur[k] = hprev[k]; hprev[k] = hpp[k]; gs[k] = gsn[k]; pr[k] = prn[k]; vcur[k] = vnext[k];
            zc[k] = zn[k]; rc[k] = rn[k]; cc[k] = cn[k];
        }
        wg_barrier_lds();
    }
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        if (sv[k]) {
            const float g0 = gacc[k] + G_base[srow[k]] + dhin[k];
            if (g0 != 0.0f) atomicAdd((dir == 0 ? p.dh0 : p.dhT) + ss[k], g0);
            if (dOacc[k] != 0.0f) atomicAdd(p.dOsum + ss[k], dOacc[k]);
        }
    }
}

// ---- parameter gradients: out[M][J] += sum_n A[n][M] B[n][J] for several (A, B, out) at once -------------------
// (A, B row-major with the reduction index as the row; rows that do not belong to a valid token are zero.)  Two
// launches for all products of a step: atb_partial_kernel computes 64x64 output tiles over row chunks and writes them
// to a partial buffer without atomics (130 chunks adding into the same 10 k addresses ran at the contended atomic
// rate: 33 us per product); atb_reduce_kernel adds the chunks of every output element.
constexpr int ATB_MAX_JOBS = 20;
struct AtbJob {
    const float *A, *B;
    float *out;
    long long N, part_off;
    int M, J, tiles_m, tiles_j, nsplit, wg0, out0;
};
struct AtbJobs {
    AtbJob j[ATB_MAX_JOBS];
    int n, total_wgs, total_out;
    long long chunk;
    float *partial;
};

// 64x64 output tile per workgroup (four wavefronts, 32x32 each as 2x2 v_mfma_f32_16x16x4_f32 tiles) over one 128-row
// chunk.  Both operands are read straight from global memory in MFMA operand order: lane (l & 15, l >> 4) needs
// A[n + (l >> 4)][m0 + (l & 15)] and B[n + (l >> 4)][j0 + (l & 15)] -- 64 contiguous bytes per 16 lanes, no LDS, no
// transposition.  (The first version staged 32x32 tiles in LDS and did 2x2 outputs per thread on the VALU: 18 TFLOP/s.)
__global__ void __launch_bounds__(256)
atb_partial_kernel(const AtbJobs jobs) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    constexpr int CH = 128;                          // rows per chunk = jobs.chunk
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].wg0) ji++;
    const AtbJob &jb = jobs.j[ji];
    const int local = blockIdx.x - jb.wg0;
    const int tiles = jb.tiles_m * jb.tiles_j, split = local / tiles, tile = local - split * tiles;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m0 = (tile / jb.tiles_j) * 64 + (wv >> 1) * 32, j0 = (tile % jb.tiles_j) * 64 + (wv & 1) * 32;
    const int M = jb.M, J = jb.J, lr = lane & 15, lk = lane >> 4;
    const long long n0 = (long long)split * CH, n1 = n0 + CH < jb.N ? n0 + CH : jb.N;
    f32x4_t acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < 2; y++) acc[x][y] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (m0 < M && j0 < J) {
        const bool ma0 = m0 + lr < M, ma1 = m0 + 16 + lr < M, jb0 = j0 + lr < J, jb1 = j0 + 16 + lr < J;
        const float *Ap = jb.A + m0 + lr, *Bp = jb.B + j0 + lr;
        for (long long nb = n0; nb < n1; nb += 16) {                 // four k-steps of 4 rows, their loads in flight together
            float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const long long n = nb + u * 4 + lk;
                const bool ok = n < n1;
                a0[u] = ok && ma0 ? Ap[n * M] : 0.0f;
                a1[u] = ok && ma1 ? Ap[n * M + 16] : 0.0f;
                b0[u] = ok && jb0 ? Bp[n * J] : 0.0f;
                b1[u] = ok && jb1 ? Bp[n * J + 16] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b0[u], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b1[u], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b0[u], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b1[u], acc[1][1], 0, 0, 0);
            }
        }
    }
    // D: column = lr, rows = lk*4 + {0..3} of each 16x16 tile
    float *po = jobs.partial + jb.part_off + (long long)split * M * J;
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < 2; y++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + x * 16 + lk * 4 + r, j = j0 + y * 16 + lr;
                if (m < M && j < J) po[(long long)m * J + j] = acc[x][y][r];
            }
}

__global__ void __launch_bounds__(256)
atb_reduce_kernel(const AtbJobs jobs) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= jobs.total_out) return;
    int ji = 0;
    while (ji + 1 < jobs.n && idx >= jobs.j[ji + 1].out0) ji++;
    const AtbJob &jb = jobs.j[ji];
    const int e = idx - jb.out0;
    const long long mj = (long long)jb.M * jb.J;
    const float *po = jobs.partial + jb.part_off + e;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;    // eight loads in flight, four chains
    int sp = 0;
    for (; sp + 8 <= jb.nsplit; sp += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = po[(long long)(sp + u) * mj];
        s0 += v[0] + v[4]; s1 += v[1] + v[5]; s2 += v[2] + v[6]; s3 += v[3] + v[7];
    }
    for (; sp < jb.nsplit; sp++) s0 += po[(long long)sp * mj];
    atomicAdd(jb.out + e, (s0 + s1) + (s2 + s3));        // two products may share an output (both chains add into dS1, dS2, dW)
}

// dC[c][s] += dOsum[s] for every label row (Osum = C.sum(0))
__global__ void add_row_to_all_kernel(float *dC, const float *__restrict__ dOsum, int K, int S) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * S) return;
    dC[idx] += dOsum[idx % S];
}

}  // namespace farnn
