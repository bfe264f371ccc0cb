// Host-visible half of K1r (chain_regs.hip.h): parameters, geometry, LDS carve and the launcher that chain_regs.hip exports to
// the library's other translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "common.hip.h"
#include "score_params.hip.h"

namespace farnn {

constexpr int RG_NWC = 6;        // compute wavefronts
constexpr int RG_WAVES = 8;      // + writer + scorer
constexpr int RG_D = 4;          // steps of block pieces in flight per lane
constexpr int RG_RQ = 4;         // rows per lane per step (one float4 of the state)
constexpr int RG_MAXG = 4;       // row groups per compute wavefront
constexpr int RG_MAXSP = 72;     // padded state count the geometry reaches (RPG <= 4 rows x 6 * G groups)
constexpr int RG_PART_STRIDE = RG_NWC * RG_MAXG * RG_MAXSP;    // floats between the two partial-sum buffers
// The WIDE form of the kernel (chain_wide.hip.h; 72 < S <= 128: the reference's 104-state automata, RE.py:56-60): two row groups
// per compute wavefront, RPG = 7 .. 11 rows of 16 bytes per lane and step, one workgroup per compute unit (256 VGPRs).
constexpr int RGW_G = 2;                       // row groups per compute wavefront
constexpr int RGW_NP = RG_NWC * RGW_G;         // partial-sum vectors per step
constexpr int RGW_MAXSP = 128;
constexpr int RGW_MAXRQ = 11;                  // rows per lane per step the instantiations reach (12 groups x 11 rows >= 128)
// floats between the two partial-sum buffers of the wide form: 12 vectors of PS floats, six flag slots, the idle lanes' dump slots,
// the identity word.  RQ <= 9 means SP <= 108, PS = 112 (1 344 + 24 + <= 104 floats); RQ = 11: PS <= 144
__host__ __device__ constexpr int rgw_part_stride(int rq) { return rq <= 9 ? 1536 : 2048; }
constexpr int RGW_XCH = 32;                    // floats of a compute wavefront's state-exchange area (2 groups x 12 rows)
// The DESTINATION-split form of the narrow kernel (chain_dest.hip.h; S <= 72, sum semiring): a wavefront owns whole outputs, four
// lanes per output row, five 16-byte chunks of the row per lane and step
constexpr int RD_LPR = 5;                      // lanes per output row (chain_dest.hip.h: 5 lanes x 4 chunks; 4: 4 lanes x 5 chunks)
constexpr int RD_XS = 80;                      // floats of one exchange row: 20 chunks of 16 bytes >= 72 floats, zeros behind S
constexpr int RD_D = 4;                        // steps of block pieces in flight per lane
constexpr int RD_XD_FLOATS = 2 * RD_XS + 128;  // the two exchange rows + two dump slots per lane

struct RegsParams {
    const float *Mf, *Mb;        // [V][SR][SP] blocks and their transposes (layout.hip.h)
    long long blk;               // floats per block
    const float *o, *h0, *hT;
    const int64_t *x, *len;
    const int *order;            // launch order or nullptr
    int sort;                    // 1: every workgroup selects its sequence by length rank (launch_order.hip.h)
    float *A, *Bk;               // stash [B][L+1][SP]
    int B, L, S, SP, CPR, V;
    int G, RPG;                  // row groups per compute wavefront, rows per group
    int RQ, D;                   // rows per lane per step and ring depth (RG_RQ, RG_D: the form for S <= 72; else the wide form)
    int pair;                    // launch order: 1 = folded (slots i and i + B/2 pair a long with a short sequence: two workgroups
                                 // per compute unit), 0 = longest first (one workgroup per compute unit)
    int PS;                      // wide form: floats between two partial-sum vectors (>= SP, = 16 mod 32: the two lanes of a row read different bank halves)
    int nl, full;
    unsigned long long *prog;    // [2][B] {epoch, rows stored} per (direction, sequence)
    unsigned long long *arr;     // [B]    {epoch, 1 << 31 | mask of the tiles it scores} of the workgroup that arrived last
    unsigned long long *done;    // the hand-off's launch counter in device memory (score_params.hip.h, BesideParams::done)
    unsigned epoch_host;         // diagnostic (FARNN_HOST_EPOCH=1, done == nullptr): the epoch as a kernel argument, the round-3 form
    int spin;                    // polls a finished workgroup spends on the tiles of its own half before it leaves them to the other
    int dbg;                     // FARNN_DBG ablation / probe mask: read by the profiling build (-DFARNN_PROBES) only
    int solo_margin;             // the scorer starts a tile alone only if the chain has at least this many steps left after it
    int dest;                    // 1: the destination-split form of the compute wavefronts (chain_dest.hip.h; narrow form, sum semiring)
    ScoreParams sp;
};
constexpr int RG_NG = 5;         // state groups of 16 the scoring stage of this kernel reaches (S <= 72 -> c16 <= 5)
constexpr int RGW_NG = 8;        // ... of the wide form (S <= 128)

struct RegsGeom {
    int G, RPG, NP, CPR, SP, rows;
    bool ok;
    bool wide;                   // the wide form (chain_wide.hip.h): RQ rows per lane and step in a ring D steps deep
    int RQ, D, PS;
};

inline RegsGeom regs_geometry(int S) {
    RegsGeom g;
    g.wide = false; g.RQ = RG_RQ; g.D = RG_D; g.PS = 0;
    g.SP = round_up(S, 4);
    g.CPR = g.SP / 4;
    g.ok = g.CPR <= 64 && S >= 1;
    g.G = g.ok ? 64 / g.CPR : 1;
    if (g.G > RG_MAXG) g.G = RG_MAXG;
    g.NP = RG_NWC * g.G;
    g.RPG = (S + g.NP - 1) / g.NP;
    if (g.RPG > RG_RQ) g.ok = false;
    g.rows = (g.NP - 1) * g.RPG + RG_RQ;      // every row index a lane's four loads can form
    // a partial-sum buffer: NP vectors, a 16-byte slot per idle lane, and its last word = the reduction's identity
    if (g.NP * g.SP + 4 * RG_NWC + 4 * (64 - g.G * g.CPR) >= RG_PART_STRIDE) g.ok = false;      // (+ the six step flags)
    if (!g.ok && S > 4 * RG_RQ && g.SP <= RGW_MAXSP) {
        // the wide form: two groups per wavefront whatever the row length (lanes 2 CPR .. 63 idle), 12 groups of RPG rows
        g.wide = true;
        g.G = RGW_G; g.NP = RGW_NP;
        g.RPG = (S + g.NP - 1) / g.NP;
        g.RQ = g.RPG <= 8 ? 8 : (g.RPG <= 9 ? 9 : RGW_MAXRQ);      // the instantiated ring widths
        g.D = 4;
        g.rows = g.NP * g.RPG;                      // (a lane's row slots past RPG re-read its last row: nothing beyond NP * RPG rows)
        g.PS = ((g.SP + 15) & ~31) + 16;            // >= SP, = 16 mod 32
        g.ok = g.RPG <= RGW_MAXRQ && g.NP * g.PS + 4 * RG_NWC + 4 * (64 - g.G * g.CPR) < rgw_part_stride(g.RQ);
    }
    return g;
}

// LDS carve, in floats (host and device agree through this one function)
struct RegsLds {
    int tok, hp, part, ol, hist, ab, scl, obuf, misc, xch, xd, total;
};
// rq: rows per lane and step of the form (RG_RQ: S <= 72; larger: the wide form, whose partial-sum buffers and exchange area differ)
// lm: the label-map path scores the tiles (label_map.hip.h): no products, no score tiles in LDS
// dest: the destination-split form (chain_dest.hip.h): its exchange rows
__host__ __device__ inline RegsLds regs_lds(int L, int SP, int NP, int c16, int Kc, bool score, int rq = RG_RQ, bool lm = false, bool dest = false) {
    const bool wide = rq != RG_RQ;
    RegsLds l;
    int at = 0;
    l.tok = at;  at += 2 * ((L + 1) & ~1);             // a 64-bit block offset per step
    l.hp = at;
    l.part = at; at += 2 * (wide ? rgw_part_stride(rq) : RG_PART_STRIDE) + 64 * 4;     // two partial-sum buffers + the idle lanes' dump slots
    l.ol = at;   at += SP;
    l.hist = at; at += (L + 1) * SP + 16;          // + the launch-order scratch's tail
    l.ab = at;   at += (score && !lm) ? 2 * RG_TT * (16 * c16 + 4) : 0;    // two tiles' products
    l.scl = at;  at += (score && !lm) ? 2 * RG_TT * Kc : 0;               // two tiles' scores
    l.obuf = at; at += score ? 2 * RG_TT * SP : 0;    // RG_NOB tiles of the other direction's rows
    l.misc = at; at += 32;
    l.xch = at;  at += wide ? RG_NWC * RGW_XCH : 0;     // wide form: where a compute wavefront hands its new state entries to its own lanes
    l.xd = at;   at += dest ? RD_XD_FLOATS : 0;
    l.total = at;
    return l;
}

// launches chain_regs_kernel<maxsr, score> on 2 * p.B workgroups; e0 / e1: optional events on the dispatch packet
int launch_chain_regs(const RegsParams &p, bool maxsr, bool score, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
// chain_wide.hip: the same for the wide form (p.RQ > RG_RQ), one workgroup per compute unit
int launch_chain_wide(const RegsParams &p, bool maxsr, bool score, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
// ... in its PAIRED form: a ring of two steps (128 VGPRs) and the label-map path's smaller LDS let two workgroups share a compute
// unit like the form for S <= 72 -- one's end-of-chain tiles overlap the other's chain (p.D == 2, p.pair == 1, scores on)
int launch_chain_wide_paired(const RegsParams &p, bool maxsr, hipStream_t s, hipEvent_t e0, hipEvent_t e1);

// chain_viterbi.hip: the two chains of a sequence and its scores + CRF decode in ONE workgroup, one launch per tagging step
// (p.prog / p.arr / p.epoch unused).  chain_viterbi_fits: tag sets of 32..159 labels whose decode fits the LDS.
bool chain_viterbi_fits(int L, int SP, int NP, int K, int Kp, bool label_map, int rq = RG_RQ);
int launch_chain_viterbi(const RegsParams &p, const ScoreParams &sp, bool maxsr, hipStream_t s, hipEvent_t e0, hipEvent_t e1);

}  // namespace farnn
