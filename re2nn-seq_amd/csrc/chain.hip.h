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
// One workgroup per (sequence, direction), specialised wavefronts:
//   * NW COMPUTE wavefronts.  The S rows of a block are split into NW*G contiguous row groups of
//     RPG rows; a compute wavefront owns G groups and lane (g, c) owns the 16-byte column chunk c
//     of the rows of group g.  Per step it reads its share of the block from the LDS ring
//     (ds_read_b128, each lane exactly the 16 bytes a DMA lane deposited), FMAs it against the
//     state, drops partial column sums in LDS, meets the others at ONE barrier and reduces only
//     the rows it consumes next step.  Compute wavefronts issue no vector-memory instruction at
//     all: the step chain is a short sequence of LDS round trips.
//   * NLD LOADER wavefronts stream the blocks HBM -> LDS with LDS-DMA (global_load_lds_dwordx4,
//     1 KiB per wave-instruction, no VGPR destination) into a ring of KS whole steps.  Block
//     addresses depend only on the token ids (known up front), never on the state, so the loaders
//     run KS-1 steps ahead of the recurrence; they retire DMA with a counted `s_waitcnt vmcnt(N)`
//     (never 0 in the loop) and publish a landed step through the step barrier.
//   * 1 WRITER wavefront copies every finished state from LDS to the HBM stash one step behind
//     (S floats per token per direction, <1.5% of the block bytes).  vmcnt is a per-wave in-order
//     counter shared by loads and stores, so stores live in their own wavefront.
// All barriers are raw `s_barrier` + lgkmcnt(0): a __syncthreads() would also drain vmcnt and
// stall the loaders / writer on a full memory round trip every step.
//
// Measured history (profiles/): a compiler-scheduled register ring drained with vmcnt(0) every few
// chunks; with every wavefront doing load+FMA+reduce itself a step took 1.37 us however deep the
// prefetch, because each wavefront's dependent LDS/SALU/VMEM-issue latencies added up serially.
//
// Roofline: HBM/MALL-bandwidth bound; algorithmic bytes per token = 2 * S*S*4.
#pragma once
#include "common.hip.h"
#include "score_decode.hip.h"
#include "launch_order.hip.h"

namespace farnn {

struct ChainParams {
    const float *Mf;        // [V][SR][SP] blocks, row-major, rows padded to SP floats, SR >= S rows
    const float *Mb;        // [V][SR][SP] the transposed blocks
    long long blk;          // floats per block (SR*SP)
    const float *o;         // [SP] output-sum vector or nullptr (no scaling)
    const float *h0, *hT;   // [S]
    const int64_t *x;       // [B][L]
    const int64_t *len;     // [B]
    const int *order;       // [B] launch order (batch_prep_kernel) or nullptr
    int sort;               // 1: no order array; every workgroup selects its sequence by length rank itself
    float *A, *Bk;          // stash [B][L+1][SP]: forward / backward states by step count
    int B, L, S, SP, CPR;   // CPR = SP/4 column chunks per row
    int V;                  // vocabulary (token ids are clamped to it)
    int NW, NLD, G, LPR, RPG, RPGp, NQ, KS;
    int NQP, PPS;           // chunks per phase per compute wave; phases per step (ring/barrier unit)
    int nl, full;
    int dbg;                // diagnostic ablation mask (FARNN_DBG); 0 in production
};

constexpr int CHAIN_MAX_THREADS = 512;     // NW compute + NLD loader + 1 writer wavefronts <= 8
constexpr int CHAIN_MAX_NP = 20;           // partial vectors per step (NW * G)
constexpr int CHAIN_MAX_G = 4;

// FQ > 0: fast path for one-phase steps of exactly FQ chunks per compute wavefront (small S): the
// next step's block share is read from the ring into registers BEFORE the reduce of the current
// step, so those LDS reads overlap the reduce instead of following it.  FQ == 0: generic loop.
// FQ > 3 (one compute wavefront owning up to 24 rows per lane group) needs the register file of a
// <= 6-wavefront workgroup for the prefetched block share.
//
// One launch per tagging step (scores + decode beside the recurrence) is chain_regs.hip.h's kernel for the geometries it covers
// (S <= 72); this kernel serves the rest, followed by the score / Viterbi kernels of score_decode.hip.h.
template <int NCH, bool MAXSR, int FQ>
__global__ void __launch_bounds__(FQ > 3 ? 384 : CHAIN_MAX_THREADS)
chain_kernel(const ChainParams p) {
    constexpr int PER = 4 * NCH;                 // DMA pieces (1 KiB each) per 4-row chunk
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform
    const int nthreads = blockDim.x;
    // Workgroup -> (launch slot, direction).  Consecutive workgroup ids go to consecutive XCDs (id % 8): ids 2s and 2s + 1 put
    // every forward chain on an even XCD and every backward chain on an odd one, so an XCD's L2 caches one direction's blocks only.
    const int item = blockIdx.x;
    const int dir = item & 1;
    const int slot = item >> 1;
    int b = p.order ? p.order[slot] : slot;
    if (p.sort) b = select_by_length_rank(p.len, p.B, p.L, folded_rank(slot, p.B), reinterpret_cast<int *>(smem), tid, nthreads);
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, RPG = p.RPG, RPGp = p.RPGp, NQ = p.NQ, KS = p.KS;
    const int NW = p.NW, NLD = p.NLD, NP = NW * p.G;

    // ---- LDS carve (all offsets multiples of 16 bytes) --------------------------------------
    const int Lr = (p.L + 3) & ~3;
    int *tok = reinterpret_cast<int *>(smem);           // [Lr]   tokens in consumption order
    float *hp = smem + Lr;                              // [NP*RPGp] state in padded row space
    float *part = hp + NP * RPGp;                       // [2][NP][SP] partial column sums
    float *ol = part + 2 * NP * SP;                     // [SP] output-sum vector (1.0 when unused)
    float *hfull = ol + SP;                             // [2][SP] finished states awaiting the writer
    char *ring = reinterpret_cast<char *>(hfull + 2 * SP);   // [KS][NW*NQ][PER] x 1 KiB
    const int NQP = p.NQP, PPS = p.PPS;
    const unsigned phase_bytes = (unsigned)(NW * NQP * PER) * 1024u;
    const int NF = nsteps * PPS;                        // phases of this item
    // a generic pointer into LDS is {shared aperture : 32-bit LDS byte address}: truncate it
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring);

    for (int k = tid; k < nsteps; k += nthreads) {
        int idx = (dir == 0) ? k : (k < len ? len - 1 - k : k);
        tok[k] = clamp_tok(p.x[(long long)b * p.L + idx], p.V);
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
    for (int j = tid; j < SP; j += nthreads) ol[j] = (p.o && j < S) ? p.o[j] : 1.0f;
    for (int j = tid; j < 2 * SP; j += nthreads) hfull[j] = 0.0f;      // pad columns of every stash row stay zero
    // the recurrence proper; a wavefront leaves it with `return` when its role is done
    {
    if (nsteps == 0) {
        for (int j = tid; j < SP; j += nthreads) stash[j] = (j < S) ? hinit[j] : 0.0f;
        return;
    }
    // no barrier yet: the loaders start their DMA below while the other wavefronts finish the
    // set-up; everybody meets at B_{-1}.

    // ---- lane -> (row group, column chunk); the loaders reproduce the compute lanes' mapping ---
    int g = lane / p.LPR;
    const int c = lane - g * p.LPR;
    const bool active = g < p.G;
    if (!active) g = 0;                  // idle lanes shadow group 0 (same lines, results unused)
    const unsigned rowb = (unsigned)SP * 4u;
    // Blocks are allocated with SR >= (NG-1)*RPG + RPGp zero-filled rows, so every row index the
    // 4-row chunks can form is in bounds: no clamping or predication in the load stream.
    unsigned lane_off[NCH];              // byte offset of (group g of wavefront 0, chunk c + 64m)
#pragma unroll
    for (int m = 0; m < NCH; m++) {
        int cc = c + 64 * m;
        cc = cc < p.CPR ? cc : p.CPR - 1;
        lane_off[m] = ((unsigned)(g * RPG) * (unsigned)SP + (unsigned)cc * 4u) * 4u;
    }
    const unsigned wave_stride = (unsigned)(p.G * RPG) * rowb;     // rows owned by one compute wave

    // =========================================================================================
    // writer wavefront
    // =========================================================================================
    if (w == NW + NLD) {
        for (int j = lane; j < SP; j += WAVE) stash[j] = (j < S) ? hinit[j] : 0.0f;   // state 0
        wg_barrier_lds();                                            // B_{-1}
        for (int f = (p.dbg & 4) ? NF : 0; f <= NF; f++) {
            wg_barrier_lds();                                        // B_f (f == NF: the final one)
            if (f > 0 && f % PPS == 0) {                             // step f/PPS-1 finished one barrier ago
                const int t = f / PPS - 1;
                const float *src = hfull + (t & 1) * SP;
                float *srow = stash + (long long)(t + 1) * SP;
                for (int j = lane; j < SP; j += WAVE) srow[j] = src[j];         // pad columns: hfull's stay zero
            }
        }
        return;
    }
    if (w > NW + NLD) {                                              // helper wavefront (FARNN_CHAIN_HELPER experiment)
        wg_barrier_lds();                                            // B_{-1}
        for (int f = (p.dbg & 4) ? NF : 0; f <= NF; f++) wg_barrier_lds();
        return;
    }

    // =========================================================================================
    // loader wavefronts
    // =========================================================================================
    if (w >= NW) {
        const int l = w - NW;
        const char *Mbase = reinterpret_cast<const char *>((dir == 0) ? p.Mf : p.Mb);
        const long long blk_bytes = p.blk * 4;
        const int nchunks = NW * NQP;                                // chunks of one phase
        int mine = 0;                                                // DMA pieces per phase, this loader
        for (int ch = l; ch < nchunks; ch += NLD) mine += PER;

        auto issue_phase = [&](int f, int tokv) {
            const int t = f / PPS;
            const int ph = f - t * PPS;
            const char *blkp = Mbase + (long long)tokv * blk_bytes;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)blkp);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)blkp >> 32));
            const char *base = reinterpret_cast<const char *>(((size_t)hi << 32) | lo);
            const unsigned dst0 = ring_lds + (unsigned)(f % KS) * phase_bytes;
            for (int ch = l; ch < nchunks; ch += NLD) {
                const int w2 = ch / NQP, qq = ch - w2 * NQP;
                int q = ph * NQP + qq;
                q = q < NQ ? q : NQ - 1;                             // ragged last phase: harmless re-read
                const unsigned goff = (unsigned)w2 * wave_stride + (unsigned)(q * 4) * rowb;
                const unsigned dst = dst0 + (unsigned)ch * (PER * 1024u);
                // the chunk's PER pieces are consecutive in LDS in (u, m) order: 4 per statement
                unsigned vo[PER];
#pragma unroll
                for (int k = 0; k < PER; k++) vo[k] = lane_off[k % NCH] + goff + (unsigned)(k / NCH) * rowb;
#pragma unroll
                for (int k = 0; k < PER; k += 4)
                    lds_dma16x4(vo[k], vo[k + 1], vo[k + 2], vo[k + 3], base, dst + (unsigned)k * 1024u);
            }
        };
        // phases in flight after the loader has issued everything up to phase `upto` (exclusive)
        // and needs phase `need` landed: (upto - 1 - need) phases of `mine` pieces each.
        const int first = NF < KS ? NF : KS;
        if (!(p.dbg & 1)) {
            // The first KS phases start before the workgroup's set-up barrier: their token ids come
            // straight from global memory (all loaded before the first DMA so that the compiler's
            // wait for them cannot drain DMA already in flight).
            int tk[8];
#pragma unroll
            for (int f = 0; f < 8; f++) {
                int t = f / PPS;
                t = t < nsteps ? t : nsteps - 1;
                const int idx = (dir == 0) ? t : (t < len ? len - 1 - t : t);
                tk[f] = (f < first) ? clamp_tok(p.x[(long long)b * p.L + idx], p.V) : 0;
            }
#pragma unroll
            for (int f = 0; f < 8; f++)
                if (f < first) issue_phase(f, __builtin_amdgcn_readfirstlane(tk[f]));
            wait_vmcnt((first - 1) * mine);                          // phase 0 has landed
        }
        wg_barrier_lds();                                            // B_{-1}
        if (!(p.dbg & 4)) {
            for (int f = 0; f < NF; f++) {
                if (!(p.dbg & 1)) {
                    const int issued = (f + KS < NF) ? f + KS : NF;  // phases issued so far
                    const int inflight_ok = issued - (f + 2);        // those newer than phase f+1
                    wait_vmcnt((inflight_ok > 0 ? inflight_ok : 0) * mine);
                }
                wg_barrier_lds();                                    // B_f: slot f%KS is free again
                if (!(p.dbg & 1) && f + KS < NF)
                    issue_phase(f + KS, __builtin_amdgcn_readfirstlane(tok[(f + KS) / PPS]));
            }
        }
        wg_barrier_lds();                                            // final barrier
        return;
    }

    // =========================================================================================
    // compute wavefronts
    // =========================================================================================
    const int gid = w * p.G + g;
    const int row0 = gid * RPG;
    const float ninf = -INFINITY;
    float4 acc[NCH];
    const char *myring = ring + (unsigned)(w * NQP) * (PER * 1024u) + lane * 16;
    const float *myhp = hp + gid * RPGp;
    const int rows_w = p.G * RPG;

    wg_barrier_lds();                                                // B_{-1}: set-up done, phase 0 in LDS
    int pb = 0;
    // the output-sum entry of the row this lane finishes (rows_w <= 64 in the common geometries)
    const float my_o = (lane < rows_w && w * rows_w + lane < S) ? ol[w * rows_w + lane] : 1.0f;

    // finish the rows this wavefront consumes next step (called right after B_t).  Lane li owns
    // row w*rows_w + li; everything that does not depend on the step is hoisted out of the loop.
    const int my_row = w * rows_w + lane;
    const bool my_valid = lane < rows_w && my_row < S;
    const int my_hp = my_valid ? (w * p.G + lane / RPG) * RPGp + lane % RPG : 0;
    const int nl_mode = p.nl;
    auto finish = [&](float s, float ov, float &hn, float &hnext) {
        if (dir == 0) { hn = apply_nl(s * ov, nl_mode); hnext = hn; }      // (:377-386)
        else          { hn = apply_nl(s, nl_mode);      hnext = hn * ov; } // (:393-402)
    };
    auto reduce_rows = [&](int t, const float *pp) {
        float *srow = hfull + (t & 1) * SP;                          // picked up by the writer
        if (my_valid) {
            const float *col = pp + my_row;
            float s = col[0];
#pragma unroll 4
            for (int q2 = 1; q2 < NP; q2++) {
                const float v = col[q2 * SP];
                s = MAXSR ? fmaxf(s, v) : s + v;
            }
            float hn, hnext;
            finish(s, my_o, hn, hnext);
            srow[my_row] = hn;
            hp[my_hp] = hnext;
        }
        for (int li = lane + WAVE; li < rows_w; li += WAVE) {        // only when a wave owns > 64 rows
            const int row = w * rows_w + li;
            if (row < S) {
                float s = pp[row];
                for (int q2 = 1; q2 < NP; q2++) {
                    const float v = pp[q2 * SP + row];
                    s = MAXSR ? fmaxf(s, v) : s + v;
                }
                float hn, hnext;
                finish(s, ol[row], hn, hnext);
                srow[row] = hn;
                const int gi = li / RPG, ii = li - gi * RPG;
                hp[(w * p.G + gi) * RPGp + ii] = hnext;
            }
        }
    };
    auto fma_chunk = [&](int q, const float4 &hv4, const float4 (&v)[4][NCH]) {
        const float hv[4] = {hv4.x, hv4.y, hv4.z, hv4.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            bool ok = true;
            if (MAXSR) {
                const int i = q * 4 + u;
                ok = i < RPG && (row0 + i) < S;
            }
#pragma unroll
            for (int m = 0; m < NCH; m++) {
                if (MAXSR) {
                    acc[m].x = fmaxf(acc[m].x, ok ? hv[u] * v[u][m].x : ninf);
                    acc[m].y = fmaxf(acc[m].y, ok ? hv[u] * v[u][m].y : ninf);
                    acc[m].z = fmaxf(acc[m].z, ok ? hv[u] * v[u][m].z : ninf);
                    acc[m].w = fmaxf(acc[m].w, ok ? hv[u] * v[u][m].w : ninf);
                } else {
                    acc[m].x = fmaf(hv[u], v[u][m].x, acc[m].x);
                    acc[m].y = fmaf(hv[u], v[u][m].y, acc[m].y);
                    acc[m].z = fmaf(hv[u], v[u][m].z, acc[m].z);
                    acc[m].w = fmaf(hv[u], v[u][m].w, acc[m].w);
                }
            }
        }
    };
    auto write_partials = [&](float *pp) {
        if (active) {
#pragma unroll
            for (int m = 0; m < NCH; m++) {
                const int cc = c + 64 * m;
                if (cc < p.CPR) st4(pp + gid * SP + cc * 4, acc[m]);
            }
        }
    };
    auto reset_acc = [&]() {
#pragma unroll
        for (int m = 0; m < NCH; m++) acc[m] = MAXSR ? make_float4(ninf, ninf, ninf, ninf)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    if constexpr (FQ > 0) {
        // ---- fast path: PPS == 1, NQ == NQP == FQ ---------------------------------------------
        float4 v[FQ][4][NCH];
        auto load_step = [&](int t) {
            const char *src = myring + (unsigned)(t % KS) * phase_bytes;
#pragma unroll
            for (int q = 0; q < FQ; q++)
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int m = 0; m < NCH; m++)
                        v[q][u][m] = *reinterpret_cast<const float4 *>(src + (q * PER + u * NCH + m) * 1024);
        };
        load_step(0);
        for (int t = 0; t < nsteps && !(p.dbg & 4); t++) {
            reset_acc();
            float4 hv4[FQ];
#pragma unroll
            for (int q = 0; q < FQ; q++) hv4[q] = ld4(myhp + q * 4);
            if (!(p.dbg & 2)) {
#pragma unroll
                for (int q = 0; q < FQ; q++) fma_chunk(q, hv4[q], v[q]);
            }
            float *pp = part + (long long)pb * NP * SP;
            write_partials(pp);
            wg_barrier_lds();                                        // B_t: step t+1 is in the ring
            pb ^= 1;
            if (t + 1 < nsteps) load_step(t + 1);                    // overlaps the reduce below
            if (!(p.dbg & 8)) reduce_rows(t, pp);
        }
    } else {
        // ---- generic path: PPS phases of up to NQP chunks per step ------------------------------
        int f = 0;
        for (int t = 0; t < nsteps && !(p.dbg & 4); t++) {
            reset_acc();
            float *pp = part + (long long)pb * NP * SP;
            for (int ph = 0; ph < PPS; ph++, f++) {
                const char *src = myring + (unsigned)(f % KS) * phase_bytes;
                if (!(p.dbg & 2)) {
                    for (int qq = 0; qq < NQP; qq++) {
                        const int q = ph * NQP + qq;
                        if (q >= NQ) break;
                        const float4 hv4 = ld4(myhp + q * 4);
                        float4 v[4][NCH];
#pragma unroll
                        for (int u = 0; u < 4; u++)
#pragma unroll
                            for (int m = 0; m < NCH; m++)
                                v[u][m] = *reinterpret_cast<const float4 *>(src + (qq * PER + u * NCH + m) * 1024);
                        fma_chunk(q, hv4, v);
                    }
                }
                if (ph == PPS - 1) write_partials(pp);   // step end: partial sums meet in LDS
                wg_barrier_lds();                                    // B_f
            }
            pb ^= 1;
            if (!(p.dbg & 8)) reduce_rows(t, pp);
        }
    }
    wg_barrier_lds();                                                // final: last state -> writer
    }
}

// Host-side geometry choice for a given S.
struct ChainGeom {
    int NCH, NW, NLD, G, LPR, RPG, RPGp, NQ, CPR, SP, SR;
    size_t shared_bytes(int L) const {      // everything but the DMA ring
        int Lr = (L + 3) & ~3;
        return sizeof(float) * ((size_t)Lr + (size_t)NW * G * RPGp + 2ull * NW * G * SP + 3ull * SP);
    }
    size_t phase_bytes(int nqp) const { return (size_t)NW * nqp * NCH * 4096; }
    size_t lds_bytes(int L, int ks, int nqp) const { return shared_bytes(L) + (size_t)ks * phase_bytes(nqp); }
    // Ring shape: phases of `nqp` chunks per compute wavefront, `ks` phases deep.  Prefer a whole
    // step per phase (one barrier per step) and two workgroups per CU; shrink the phase, then the
    // depth, then give up co-residency, until it fits the CU's 160 KiB.
    bool pick_ring(int L, int want_ks, int &ks, int &nqp) const {
        const size_t cap2 = 80 * 1024, cap1 = 158 * 1024;
        want_ks = want_ks < 2 ? 2 : (want_ks > 8 ? 8 : want_ks);
        for (int pass = 0; pass < 2; pass++) {
            const size_t cap = pass == 0 ? cap2 : cap1;
            for (nqp = NQ; nqp >= 1; nqp = (nqp + 1) / 2 == nqp ? nqp - 1 : (nqp + 1) / 2) {
                for (ks = want_ks; ks >= 2; ks--)
                    if (lds_bytes(L, ks, nqp) <= cap) return true;
                if (nqp == 1) break;
            }
        }
        return false;
    }
};

inline ChainGeom chain_geometry_nw(int S, int nw, int nld) {
    ChainGeom gm;
    gm.SP = round_up(S, 4);
    gm.CPR = gm.SP / 4;
    if (gm.CPR <= 64) {
        gm.NCH = 1; gm.LPR = gm.CPR; gm.G = 64 / gm.CPR;
        if (gm.G > CHAIN_MAX_G) gm.G = CHAIN_MAX_G;
    } else { gm.NCH = (gm.CPR + 63) / 64; gm.LPR = 64; gm.G = 1; }
    gm.NLD = nld;
    gm.NW = nw;
    int ng = gm.NW * gm.G;
    gm.RPG = (S + ng - 1) / ng;
    gm.RPGp = round_up(gm.RPG, 4);
    gm.NQ = gm.RPGp / 4;
    gm.SR = (ng - 1) * gm.RPG + gm.RPGp;      // allocated (zero-padded) rows per block
    if (gm.SR < S) gm.SR = S;
    return gm;
}

inline ChainGeom chain_geometry(int S, int rows_per_group_target, int nld, int L_hint = 128) {
    nld = nld < 1 ? 1 : (nld > 4 ? 4 : nld);
    const int nw_max = CHAIN_MAX_THREADS / 64 - 1 - nld;
    ChainGeom g0 = chain_geometry_nw(S, 1, nld);
    int groups = (S + rows_per_group_target - 1) / rows_per_group_target;
    int nw = (groups + g0.G - 1) / g0.G;
    if (nw < 1) nw = 1;
    if (nw > nw_max) nw = nw_max;
    // fewer compute wavefronts when even the smallest ring would not fit the LDS
    for (; nw >= 1; nw--) {
        ChainGeom gm = chain_geometry_nw(S, nw, nld);
        int ks, nqp;
        if (gm.pick_ring(L_hint, 2, ks, nqp) || nw == 1) return gm;
    }
    return chain_geometry_nw(S, 1, nld);
}

}  // namespace farnn
