// EXPERIMENT (not built): column-owned form of the recurrence for S <= 128.  Parity-green on the whole
// suite; 63 us vs 62 us (rows form) on the all-length-64 batch, 52.6 us vs 43.3 us on the ragged headline
// batch: the LDS round trip it removes is paid back by the DPP reduce chains, the scalar state reads and
// the remainder wavefront, and it profits less from sharing a CU with a short sequence.

// K1c -- the recurrence for small automata (S <= 128), column-owned form.
//
// Same workgroup anatomy, LDS-DMA ring and barrier protocol as chain_kernel (chain.hip.h): NW compute
// wavefronts, NLD loaders running ahead on token ids, one writer.  What changes is who owns what:
// chain_kernel splits the ROWS of a block over the compute wavefronts, so every step ends with partial
// column sums going through LDS and a second pass that reduces them.  Here a compute wavefront owns a
// slice of the COLUMNS for all rows: lane (c, g) holds the 16-byte column chunk c of row group g, the G2
// row groups of a chunk sit on adjacent lanes and meet on the DPP network (row_shr / row_bcast), and the
// lane that ends up with a column's total finishes it (o scaling, non-linearity) and publishes the new
// state entry.  One LDS round trip per step less on the serial chain of the longest sequence.
//
// Lanes are dense: a wavefront owns CW chunks with CW * G2 = 64, both powers of two.  The chunks of a
// block are dealt 8 per wavefront (G2 = 8); the last wavefront takes the remainder with
// CW = pow2ceil(rest) and a correspondingly larger G2 (S = 71: 18 chunks = 8 + 8 + 2, the third wavefront
// runs 2 chunks x 32 row groups of 3 rows).  A DMA piece is one row index of one wavefront: 64 lanes x
// 16 bytes; the DMA lanes walk a row's chunks first (coalesced requests), the compute lane (c, g) reads
// slot g * CW + c of the piece (2-way LDS bank conflicts, against 4x fewer memory requests).
#pragma once
#include "chain.hip.h"

namespace farnn {

constexpr int CC_MAX_NW = 4;
constexpr int CC_THREADS = 576;            // up to 4 compute + 4 loader + 1 writer wavefronts

struct ChainColsGeom {
    int NW, NLD, KS;
    int c0[CC_MAX_NW], CW[CC_MAX_NW], G2[CC_MAX_NW], RPL[CC_MAX_NW], PO[CC_MAX_NW + 1];
    int HP;                                // floats of the state vector incl. zero padding (max G2 * RPL)
    int RMAX;                              // max RPL
    int SR;                                // rows a block must have (zero rows behind S)
    size_t lds_bytes(int L, int SP, int ks) const {
        const int Lr = (L + 3) & ~3;
        return sizeof(float) * ((size_t)Lr + 2ull * HP + 3ull * SP) + (size_t)ks * PO[NW] * 1024;
    }
};

// false: this S is better served by chain_kernel
inline bool chain_cols_geometry(int S, int nld, int L, ChainColsGeom &g) {
    const int SP = round_up(S, 4), CPR = SP / 4;
    if (CPR > 8 * CC_MAX_NW) return false;
    g.NLD = nld;
    if (CPR <= 8) {
        g.NW = 1;
        int cw = 1; while (cw < CPR) cw *= 2;
        g.c0[0] = 0; g.CW[0] = cw;
    } else {
        g.NW = (CPR + 7) / 8;
        for (int w = 0; w < g.NW; w++) { g.c0[w] = 8 * w; g.CW[w] = 8; }
        int rest = CPR - 8 * (g.NW - 1), cw = 1;
        while (cw < rest) cw *= 2;
        g.CW[g.NW - 1] = cw;
    }
    g.HP = 0; g.RMAX = 0; g.PO[0] = 0;
    for (int w = 0; w < g.NW; w++) {
        g.G2[w] = 64 / g.CW[w];
        g.RPL[w] = (S + g.G2[w] - 1) / g.G2[w];
        g.PO[w + 1] = g.PO[w] + g.RPL[w];
        g.HP = g.HP > g.G2[w] * g.RPL[w] ? g.HP : g.G2[w] * g.RPL[w];
        g.RMAX = g.RMAX > g.RPL[w] ? g.RMAX : g.RPL[w];
    }
    g.SR = round_up(g.HP, 4) > S + 1 ? round_up(g.HP, 4) : S + 1;
    g.HP = round_up(g.HP, 4) + 16;          // + RMAX entries the remainder wavefront may read (as zeros)
    if (g.RMAX > 16) return false;
    for (g.KS = 3; g.KS >= 2; g.KS--)
        if (g.lds_bytes(L, SP, g.KS) <= 80 * 1024) return true;
    g.KS = 3;
    return g.lds_bytes(L, SP, 3) <= 158 * 1024;
}

struct ChainColsParams {
    const float *Mf, *Mb;   // [V][SR][SP] blocks / transposed blocks
    long long blk;
    const float *o, *h0, *hT;
    const int64_t *x, *len;
    const int *order;
    int sort;
    float *A, *Bk;
    int B, L, S, SP, CPR, nl, full;
    int dbg;                // diagnostic ablation mask (FARNN_DBG); 0 in production
    ChainColsGeom g;
};

template <bool MAXSR>
__device__ __forceinline__ float cc_dpp(float v, float o) { return MAXSR ? fmaxf(v, o) : v + o; }

// combine over the G2 adjacent lanes of a column chunk; complete in the lane with g == G2 - 1.
// The four values of a lane go through each stage together: the hazard slots of one DPP op are filled
// by the next value's, and the uniform G2 tests cost one scalar branch per stage, not per value.
template <bool MAXSR, int CTRL, int MASK>
__device__ __forceinline__ void cc_stage(float4 &a) {
    const int idb = __float_as_int(MAXSR ? -INFINITY : 0.0f);
    const float x = __int_as_float(__builtin_amdgcn_update_dpp(idb, __float_as_int(a.x), CTRL, MASK, 0xf, false));
    const float y = __int_as_float(__builtin_amdgcn_update_dpp(idb, __float_as_int(a.y), CTRL, MASK, 0xf, false));
    const float z = __int_as_float(__builtin_amdgcn_update_dpp(idb, __float_as_int(a.z), CTRL, MASK, 0xf, false));
    const float w = __int_as_float(__builtin_amdgcn_update_dpp(idb, __float_as_int(a.w), CTRL, MASK, 0xf, false));
    a.x = cc_dpp<MAXSR>(a.x, x); a.y = cc_dpp<MAXSR>(a.y, y); a.z = cc_dpp<MAXSR>(a.z, z); a.w = cc_dpp<MAXSR>(a.w, w);
}

template <bool MAXSR>
__device__ __forceinline__ void cc_group_reduce4(float4 &a, int G2) {
    if (G2 == 8) {                                  // every wavefront but the remainder one
        cc_stage<MAXSR, 0x111, 0xf>(a); cc_stage<MAXSR, 0x112, 0xf>(a); cc_stage<MAXSR, 0x114, 0xf>(a);
        return;
    }
    if (G2 >= 2) cc_stage<MAXSR, 0x111, 0xf>(a);    // row_shr:1
    if (G2 >= 4) cc_stage<MAXSR, 0x112, 0xf>(a);    // row_shr:2
    if (G2 >= 8) cc_stage<MAXSR, 0x114, 0xf>(a);    // row_shr:4
    if (G2 >= 16) cc_stage<MAXSR, 0x118, 0xf>(a);   // row_shr:8
    if (G2 >= 32) cc_stage<MAXSR, 0x142, 0xa>(a);   // row_bcast:15
    if (G2 >= 64) cc_stage<MAXSR, 0x143, 0xc>(a);   // row_bcast:31
}

__device__ __forceinline__ float4 cc_nl4(float4 a, int nl) {
    switch (nl) {
        case FARNN_NL_RELU: return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
        case FARNN_NL_TANH: return make_float4(tanhf(a.x), tanhf(a.y), tanhf(a.z), tanhf(a.w));
        case FARNN_NL_RELUTANH: return make_float4(tanhf(fmaxf(a.x, 0.f)), tanhf(fmaxf(a.y, 0.f)), tanhf(fmaxf(a.z, 0.f)),
                                                   tanhf(fmaxf(a.w, 0.f)));
        default: return a;
    }
}

template <bool MAXSR, int RMAX>
__global__ void __launch_bounds__(CC_THREADS)
chain_cols_kernel(const ChainColsParams p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = blockDim.x;
    const int item = blockIdx.x, dir = item & 1;
    const ChainColsGeom &G = p.g;
    int b = p.order ? p.order[item >> 1] : (item >> 1);
    if (p.sort) b = select_by_length_rank(p.len, p.B, p.L, folded_rank(item >> 1, p.B), reinterpret_cast<int *>(smem), tid, nthreads);
    const int len = (int)p.len[b];
    const int nsteps = p.full ? p.L : len;
    const int S = p.S, SP = p.SP, NW = G.NW, NLD = G.NLD, KS = G.KS, HP = G.HP;
    const int PPS_pieces = G.PO[NW];                    // DMA pieces (1 KiB) of one step

    // ---- LDS carve ---------------------------------------------------------------------------------
    const int Lr = (p.L + 3) & ~3;
    int *tok = reinterpret_cast<int *>(smem);           // [Lr]
    float *hn = smem + Lr;                              // [2][HP] state fed to the next step (ping-pong)
    float *ol = hn + 2 * HP;                            // [SP] output-sum vector (1.0 when unused)
    float *hst = ol + SP;                               // [2][SP] finished states awaiting the writer
    char *ring = reinterpret_cast<char *>(hst + 2 * SP);
    const unsigned step_bytes = (unsigned)PPS_pieces * 1024u;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring);

    for (int k = tid; k < nsteps; k += nthreads) {
        const int idx = (dir == 0) ? k : (k < len ? len - 1 - k : k);
        tok[k] = (int)p.x[(long long)b * p.L + idx];
    }
    float *stash = (dir == 0 ? p.A : p.Bk) + (long long)b * (p.L + 1) * SP;
    const float *hinit = (dir == 0) ? p.h0 : p.hT;
    for (int j = tid; j < 2 * HP; j += nthreads) {
        const int r = j < HP ? j : j - HP;
        float v = 0.0f;
        if (j < HP && r < S) { v = hinit[r]; if (dir == 1 && p.o) v *= p.o[r]; }     // backward input pre-scaled (:393)
        hn[j] = v;
    }
    for (int j = tid; j < SP; j += nthreads) ol[j] = (p.o && j < S) ? p.o[j] : 1.0f;
    if (nsteps == 0) {
        for (int j = tid; j < SP; j += nthreads) stash[j] = (j < S) ? hinit[j] : 0.0f;
        return;
    }
    const unsigned rowb = (unsigned)SP * 4u;

    // =============================================================================================
    // writer wavefront
    // =============================================================================================
    if (w == NW + NLD) {
        for (int j = lane; j < SP; j += WAVE) stash[j] = (j < S) ? hinit[j] : 0.0f;   // state 0
        wg_barrier_lds();                                            // B_{-1}
        for (int t = 0; t < nsteps; t++) {
            wg_barrier_lds();                                        // B_t: the state after step t is in hst[t & 1]
            const float *src = hst + (t & 1) * SP;                   // (rewritten only after B_{t+1})
            float *srow = stash + (long long)(t + 1) * SP;
            for (int j = lane; j < S; j += WAVE) srow[j] = src[j];
        }
        wg_barrier_lds();                                            // final barrier
        return;
    }

    // =============================================================================================
    // loader wavefronts: piece pc of a step = row index (pc - PO[w2]) of compute wavefront w2
    // =============================================================================================
    if (w >= NW) {
        const int l = w - NW;
        const char *Mbase = reinterpret_cast<const char *>((dir == 0) ? p.Mf : p.Mb);
        const long long blk_bytes = p.blk * 4;
        unsigned lane_base[CC_MAX_NW];               // byte offset of (row group g, chunk c) of wavefront w2, row 0
#pragma unroll
        for (int w2 = 0; w2 < CC_MAX_NW; w2++) {
            // DMA lanes walk a row's chunks first (adjacent lanes = adjacent 16 bytes: coalesced requests);
            // the compute lane (c, g) reads LDS slot g * CW + c of the piece
            const int cw = G.CW[w2 < NW ? w2 : 0], g = lane / cw, c = lane & (cw - 1);
            int cc = G.c0[w2 < NW ? w2 : 0] + c;
            cc = cc < p.CPR ? cc : p.CPR - 1;
            lane_base[w2] = ((unsigned)(g * G.RPL[w2 < NW ? w2 : 0]) * (unsigned)SP + (unsigned)cc * 4u) * 4u;
        }
        // this loader's pieces: a contiguous range (consecutive LDS destinations: four per asm statement)
        const int per = (PPS_pieces + NLD - 1) / NLD;
        const int pc0 = l * per < PPS_pieces ? l * per : PPS_pieces;
        const int pc1 = pc0 + per < PPS_pieces ? pc0 + per : PPS_pieces;
        const int mine = pc1 - pc0;
        constexpr int MAXP = 16;                      // pieces per loader per step (RMAX 16 x NW 4 / NLD >= 4)
        unsigned voffs[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; k++) {
            const int pc = pc0 + k < PPS_pieces ? pc0 + k : PPS_pieces - 1;
            int w2 = 0;
#pragma unroll
            for (int q = 1; q < CC_MAX_NW; q++) w2 += (q < NW && pc >= G.PO[q]) ? 1 : 0;
            unsigned lb = lane_base[0];
            int po = G.PO[0];
#pragma unroll
            for (int q = 1; q < CC_MAX_NW; q++) { lb = w2 == q ? lane_base[q] : lb; po = w2 == q ? G.PO[q] : po; }
            voffs[k] = lb + (unsigned)(pc - po) * rowb;
        }
        auto issue_step = [&](int t, int tokv) {
            const char *blkp = Mbase + (long long)tokv * blk_bytes;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)blkp);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)blkp >> 32));
            const char *base = reinterpret_cast<const char *>(((size_t)hi << 32) | lo);
            const unsigned dst0 = ring_lds + (unsigned)(t % KS) * step_bytes + (unsigned)pc0 * 1024u;
            if (p.dbg & 1) return;
#pragma unroll
            for (int k = 0; k < MAXP; k += 4) {
                if (k + 4 <= mine) {
                    lds_dma16x4(voffs[k], voffs[k + 1], voffs[k + 2], voffs[k + 3], base, dst0 + (unsigned)k * 1024u);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (k + u < mine) lds_dma16(voffs[k + u], base, dst0 + (unsigned)(k + u) * 1024u);
                }
            }
        };
        const int first = nsteps < KS ? nsteps : KS;
        {
            int tk[4];
#pragma unroll
            for (int f = 0; f < 4; f++) {
                const int t = f < nsteps ? f : nsteps - 1;
                const int idx = (dir == 0) ? t : (t < len ? len - 1 - t : t);
                tk[f] = (f < first) ? (int)p.x[(long long)b * p.L + idx] : 0;
            }
#pragma unroll
            for (int f = 0; f < 4; f++)
                if (f < first) issue_step(f, __builtin_amdgcn_readfirstlane(tk[f]));
            if (!(p.dbg & 1)) wait_vmcnt((first - 1) * mine);         // step 0 has landed
        }
        wg_barrier_lds();                                            // B_{-1}
        for (int t = 0; t < nsteps; t++) {
            const int issued = (t + KS < nsteps) ? t + KS : nsteps;  // steps issued so far
            const int inflight_ok = issued - (t + 2);                // those newer than step t+1
            if (!(p.dbg & 1)) wait_vmcnt((inflight_ok > 0 ? inflight_ok : 0) * mine);
            wg_barrier_lds();                                        // B_t: slot t%KS is free again
            if (t + KS < nsteps) issue_step(t + KS, __builtin_amdgcn_readfirstlane(tok[t + KS]));
        }
        wg_barrier_lds();                                            // final barrier
        return;
    }

    // =============================================================================================
    // compute wavefronts
    // =============================================================================================
    const int G2 = G.G2[w], RPL = G.RPL[w], CW = G.CW[w];
    const int g = lane & (G2 - 1), c = lane / G2;
    const int chunk = G.c0[w] + c;
    const bool colok = chunk < p.CPR;                                // the remainder wavefront may own padding chunks
    const int row0 = g * RPL;
    const char *myring = ring + (unsigned)G.PO[w] * 1024u + (g * CW + c) * 16;
    const bool fin = g == G2 - 1 && colok;                           // this lane finishes 4 columns
    const int col0 = chunk * 4;
    const float ninf = -INFINITY;

    wg_barrier_lds();                                                // B_{-1}: set-up done, step 0 in LDS
    const float4 ov4 = colok ? ld4(ol + col0) : make_float4(1.f, 1.f, 1.f, 1.f);
    const int4 cm = make_int4(col0 + 0 < S, col0 + 1 < S, col0 + 2 < S, col0 + 3 < S);    // columns behind S stay zero
    const int nl_mode = p.nl;
    // RMAX is the row count of the wide wavefronts exactly; the remainder wavefront (fewer rows per lane)
    // runs the same unrolled code with its surplus rows re-reading its last piece against a zero state
    // entry -- no per-row branches (a uniform guard per unrolled row compiles to a scalar branch each).
    float4 v[RMAX];
    int poff[RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; i++) poff[i] = (i < RPL ? i : RPL - 1) * 1024;
    auto load_step = [&](int t) {
        const char *src = myring + (unsigned)(t % KS) * step_bytes;
        if (p.dbg & 4) return;
#pragma unroll
        for (int i = 0; i < RMAX; i++) v[i] = *reinterpret_cast<const float4 *>(src + poff[i]);
    };
#pragma unroll
    for (int i = 0; i < RMAX; i++) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    load_step(0);
    for (int t = 0; t < nsteps; t++) {
        const float *hc = hn + (t & 1) * HP + row0;
        float *hnx = hn + ((t + 1) & 1) * HP;
        float hv[RMAX];
#pragma unroll
        for (int i = 0; i < RMAX; i++) hv[i] = i < RPL ? hc[i] : 0.0f;          // hn is padded by RMAX entries
        float4 acc = MAXSR ? make_float4(ninf, ninf, ninf, ninf) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RMAX; i++) {
            if (MAXSR) {
                const bool ok = i < RPL && row0 + i < S;
                acc.x = fmaxf(acc.x, ok ? hv[i] * v[i].x : ninf); acc.y = fmaxf(acc.y, ok ? hv[i] * v[i].y : ninf);
                acc.z = fmaxf(acc.z, ok ? hv[i] * v[i].z : ninf); acc.w = fmaxf(acc.w, ok ? hv[i] * v[i].w : ninf);
            } else {
                acc.x = fmaf(hv[i], v[i].x, acc.x); acc.y = fmaf(hv[i], v[i].y, acc.y);
                acc.z = fmaf(hv[i], v[i].z, acc.z); acc.w = fmaf(hv[i], v[i].w, acc.w);
            }
        }
        if (!(p.dbg & 2)) cc_group_reduce4<MAXSR>(acc, G2);
        if (fin && !(p.dbg & 8)) {
            float4 h4, n4;                                          // stash copy / next state
            if (dir == 0) {                                          // (:377-386)
                h4 = cc_nl4(make_float4(acc.x * ov4.x, acc.y * ov4.y, acc.z * ov4.z, acc.w * ov4.w), nl_mode);
                n4 = h4;
            } else {                                                 // (:393-402)
                h4 = cc_nl4(acc, nl_mode);
                n4 = make_float4(h4.x * ov4.x, h4.y * ov4.y, h4.z * ov4.z, h4.w * ov4.w);
            }
            h4.x = cm.x ? h4.x : 0.f; h4.y = cm.y ? h4.y : 0.f; h4.z = cm.z ? h4.z : 0.f; h4.w = cm.w ? h4.w : 0.f;
            n4.x = cm.x ? n4.x : 0.f; n4.y = cm.y ? n4.y : 0.f; n4.z = cm.z ? n4.z : 0.f; n4.w = cm.w ? n4.w : 0.f;
            st4(hst + (t & 1) * SP + col0, h4);
            st4(hnx + col0, n4);
        }
        wg_barrier_lds();                                            // B_t: step t+1 is in the ring
        if (t + 1 < nsteps) load_step(t + 1);
    }
    wg_barrier_lds();                                                // final: last state -> writer
}

}  // namespace farnn
