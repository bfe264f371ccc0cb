// K1 -- the bidirectional automaton recurrence ("state chain") for dense transition blocks.
//
// Reference: FARNN_S_O_I_S.forward_score time loop (model_onehot.py:372-403), and the same loop
// of FARNN_S_O (:89-102) / FARNN_S_O_I (:255-282).  Per sequence b and direction d:
//
//     fwd:  a[k+1] = nl( (a[k] . M[x_k]) * o )            a[0] = h0
//     bwd:  b[k+1] = nl( M[x'_k] . (b[k] * o) )           b[0] = hT      x' = tokens right-to-left
//
// where M[w] = T[w] + W is the S x S transition block of word w (premixed once at create()).
// Both directions are brought to the same form  out[col] = sum_row in[row] * Blk[row][col]  by
// keeping a transposed copy of every block in HBM (288 GB makes the 2x footprint a non-issue and
// it keeps every load of either direction a fully coalesced 16-byte-per-lane row read).
//
// Mapping (one workgroup per (sequence, direction), NW wavefronts):
//   * the S rows of a block are split into NW*G contiguous row groups of RPG rows; a wavefront
//     owns G groups; lane (g, c) streams the 16-byte column chunk c of the rows of group g;
//   * the block is never staged through LDS: every byte is used once by one lane, so it goes
//     straight to VGPRs through a 3-deep register ring of 4-row chunks.  The addresses depend only
//     on the token ids (known up front), never on the state, so the ring runs ahead across step
//     boundaries and the serial dependence is only the tiny state vector;
//   * per step each lane holds partial column sums; they meet in LDS (double-buffered by step
//     parity), ONE workgroup barrier, then every wavefront reduces only the rows it will need as
//     `in[row]` next step -- no second barrier;
//   * every state a[k], b[k] is written once to the stash in HBM (S floats per token per direction,
//     <1.5% of the block bytes) for the scoring kernel.
//
// Roofline: HBM/MALL-bandwidth bound; algorithmic bytes per token = 2 * S*S*4.
#pragma once
#include "common.hip.h"

namespace farnn {

struct ChainParams {
    const float *Mf;        // [V][S][SP] blocks, row-major, rows padded to SP floats
    const float *Mb;        // [V][S][SP] the transposed blocks
    long long blk;          // floats per block (S*SP)
    const float *o;         // [SP] output-sum vector or nullptr (no scaling)
    const float *h0, *hT;   // [S]
    const int64_t *x;       // [B][L]
    const int64_t *len;     // [B]
    float *A, *Bk;          // stash [B][L+1][SP]: forward / backward states by step count
    int B, L, S, SP, CPR;   // CPR = SP/4 column chunks per row
    int NW, G, LPR, RPG, RPGp, NQ;
    int nl, full;
};

// register-ring depth in 4-row chunks, and the workgroup-size cap that keeps the ring in VGPRs
constexpr int chain_pf(int nch) { return nch <= 2 ? 3 : 2; }
constexpr int chain_max_threads(int nch) { return nch == 1 ? 1024 : 512; }

template <int NCH, bool MAXSR>
__global__ void __launch_bounds__(chain_max_threads(NCH))
chain_kernel(const ChainParams p) {
    constexpr int CHAIN_PF = chain_pf(NCH);
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int nthreads = blockDim.x;
    const int item = blockIdx.x;
    const int b = item >> 1, dir = item & 1;
    const int len = (int)p.len[b];
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, RPG = p.RPG, RPGp = p.RPGp, NQ = p.NQ;
    const int NP = p.NW * p.G;

    // ---- LDS carve (all offsets multiples of 16 bytes) --------------------------------------
    const int Lr = (p.L + 3) & ~3;
    int *tok = reinterpret_cast<int *>(smem);           // [Lr]   tokens in consumption order
    float *hp = smem + Lr;                              // [NP*RPGp] state in padded row space
    float *part = hp + NP * RPGp;                       // [2][NP][SP] partial column sums

    for (int k = tid; k < nsteps; k += nthreads) {
        int idx = (dir == 0) ? k : (k < len ? len - 1 - k : k);
        tok[k] = (int)p.x[(long long)b * p.L + idx];
    }
    float *stash = (dir == 0 ? p.A : p.Bk) + (long long)b * (p.L + 1) * SP;
    const float *hinit = (dir == 0) ? p.h0 : p.hT;
    for (int idx = tid; idx < NP * RPGp; idx += nthreads) {
        int gi = idx / RPGp, ii = idx - gi * RPGp;
        int row = gi * RPG + ii;
        float v = 0.0f;
        if (ii < RPG && row < S) {
            v = hinit[row];
            if (dir == 1 && p.o) v *= p.o[row];        // backward input is pre-scaled (:393)
        }
        hp[idx] = v;
    }
    for (int j = tid; j < SP; j += nthreads) stash[j] = (j < S) ? hinit[j] : 0.0f;
    __syncthreads();
    if (nsteps == 0) return;

    // ---- lane -> (row group, column chunk) ---------------------------------------------------
    int g = lane / p.LPR;
    const int c = lane - g * p.LPR;
    const bool active = g < p.G;
    if (!active) g = 0;
    const int gid = w * p.G + g;
    const int row0 = gid * RPG;
    const float *Mbase = (dir == 0) ? p.Mf : p.Mb;

    float4 ring[CHAIN_PF][4][NCH];
    float4 acc[NCH];
    const float ninf = -INFINITY;
#pragma unroll
    for (int m = 0; m < NCH; m++) acc[m] = MAXSR ? make_float4(ninf, ninf, ninf, ninf)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);

    // Row/column indices are clamped instead of predicated: a clamped load re-reads a line the
    // neighbouring lanes fetch anyway, and its product is multiplied by a zero state entry
    // (padded hp rows stay 0) or lands in a column nobody reads.
    auto issue = [&](int t, int q, float4 (&dst)[4][NCH]) {
        int tt = t < nsteps ? t : nsteps - 1;
        const float *blkp = Mbase + (long long)__builtin_amdgcn_readfirstlane(tok[tt]) * p.blk;
        if (active) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int row = row0 + q * 4 + u;
                row = row < S ? row : S - 1;
#pragma unroll
                for (int m = 0; m < NCH; m++) {
                    int cc = c + 64 * m;
                    cc = cc < p.CPR ? cc : p.CPR - 1;
                    dst[u][m] = ld4(blkp + (long long)row * SP + cc * 4);
                }
            }
        }
    };

    auto consume = [&](int q, const float4 (&src)[4][NCH]) {
        const float4 hv4 = ld4(hp + gid * RPGp + q * 4);
        const float hv[4] = {hv4.x, hv4.y, hv4.z, hv4.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (MAXSR) {
                int i = q * 4 + u;
                bool ok = i < RPG && (row0 + i) < S;
#pragma unroll
                for (int m = 0; m < NCH; m++) {
                    acc[m].x = fmaxf(acc[m].x, ok ? hv[u] * src[u][m].x : ninf);
                    acc[m].y = fmaxf(acc[m].y, ok ? hv[u] * src[u][m].y : ninf);
                    acc[m].z = fmaxf(acc[m].z, ok ? hv[u] * src[u][m].z : ninf);
                    acc[m].w = fmaxf(acc[m].w, ok ? hv[u] * src[u][m].w : ninf);
                }
            } else {
#pragma unroll
                for (int m = 0; m < NCH; m++) {
                    acc[m].x = fmaf(hv[u], src[u][m].x, acc[m].x);
                    acc[m].y = fmaf(hv[u], src[u][m].y, acc[m].y);
                    acc[m].z = fmaf(hv[u], src[u][m].z, acc[m].z);
                    acc[m].w = fmaf(hv[u], src[u][m].w, acc[m].w);
                }
            }
        }
    };

    int pb = 0;
    auto step_end = [&](int t) {
        float *pp = part + (long long)pb * NP * SP;
        if (active) {
#pragma unroll
            for (int m = 0; m < NCH; m++) {
                int cc = c + 64 * m;
                if (cc < p.CPR) st4(pp + gid * SP + cc * 4, acc[m]);
            }
        }
        __syncthreads();
        // each wavefront finishes exactly the rows it consumes next step
        const int rows_w = p.G * RPG;
        float *srow = stash + (long long)(t + 1) * SP;
        for (int li = lane; li < rows_w; li += WAVE) {
            int row = w * rows_w + li;
            if (row < S) {
                float s = pp[row];
                for (int q2 = 1; q2 < NP; q2++) {
                    float v = pp[q2 * SP + row];
                    s = MAXSR ? fmaxf(s, v) : s + v;
                }
                float ov = p.o ? p.o[row] : 1.0f;
                float hn, hnext;
                if (dir == 0) { hn = apply_nl(s * ov, p.nl); hnext = hn; }      // (:377-386)
                else          { hn = apply_nl(s, p.nl);      hnext = hn * ov; } // (:393-402)
                srow[row] = hn;
                int gi = li / RPG, ii = li - gi * RPG;
                hp[(w * p.G + gi) * RPGp + ii] = hnext;
            }
        }
        pb ^= 1;
#pragma unroll
        for (int m = 0; m < NCH; m++) acc[m] = MAXSR ? make_float4(ninf, ninf, ninf, ninf)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    // ---- software pipeline over the flat (step, chunk) stream --------------------------------
    int it = 0, iq = 0;
#pragma unroll
    for (int s = 0; s < CHAIN_PF; s++) {
        issue(it, iq, ring[s]);
        if (++iq == NQ) { iq = 0; it++; }
    }
    int ct = 0, cq = 0;
    const int total = nsteps * NQ;
    for (int n = 0; n < total; n += CHAIN_PF) {
#pragma unroll
        for (int s = 0; s < CHAIN_PF; s++) {
            if (n + s < total) {
                consume(cq, ring[s]);
                issue(it, iq, ring[s]);
                if (++iq == NQ) { iq = 0; it++; }
                if (++cq == NQ) { step_end(ct); cq = 0; ct++; }
            }
        }
    }
}

// Host-side geometry choice for a given S.
struct ChainGeom {
    int NCH, NW, G, LPR, RPG, RPGp, NQ, CPR, SP;
    size_t lds_bytes(int L) const {
        int Lr = (L + 3) & ~3;
        return sizeof(float) * ((size_t)Lr + (size_t)NW * G * RPGp + 2ull * NW * G * SP);
    }
};

inline ChainGeom chain_geometry(int S, int rows_per_group_target) {
    ChainGeom gm;
    gm.SP = round_up(S, 4);
    gm.CPR = gm.SP / 4;
    if (gm.CPR <= 64) { gm.NCH = 1; gm.LPR = gm.CPR; gm.G = 64 / gm.CPR; }
    else              { gm.NCH = (gm.CPR + 63) / 64; gm.LPR = 64; gm.G = 1; }
    int groups = (S + rows_per_group_target - 1) / rows_per_group_target;
    int nw = (groups + gm.G - 1) / gm.G;
    const int nw_max = chain_max_threads(gm.NCH) / 64;
    if (nw < 1) nw = 1;
    if (nw > nw_max) nw = nw_max;
    gm.NW = nw;
    int ng = gm.NW * gm.G;
    gm.RPG = (S + ng - 1) / ng;
    gm.RPGp = round_up(gm.RPG, 4);
    gm.NQ = gm.RPGp / 4;
    return gm;
}

}  // namespace farnn
