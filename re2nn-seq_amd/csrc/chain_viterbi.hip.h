// K1v's kernel and LDS plan (chain_viterbi.hip has the account): a header because two translation units instantiate it --
// chain_viterbi.hip (S <= 72) and chain_viterbi_wide.hip (72 < S <= 108).
#pragma once
#include "common.hip.h"
#include "host_util.hip.h"
#include "chain_regs.hip.h"
#include "viterbi_hist.hip.h"

namespace farnn {

// LDS plan of a launch (floats).  `lds_rows`: the chains' halves are placed clear of the decode's product area and of the area the
// output matrix's image is staged in, so that (i) the image is fetched while the chains run, (ii) the products are formed from
// the chains' own state rows in LDS -- no stash, no writer traffic, no reload.  Else the halves lie at 0 / half and the rows go
// through the stash (longer sequences, larger tag sets).
struct ChainViterbiPlan {
    int off0, off1, half, image_off, image_pieces, lds_rows;
    size_t bytes;
};

// label_map: the scores come from the label map (label_map.hip.h) -- no products area, no image; a wavefront writes a token's
// emission row while others still read state rows, so the halves lie BEHIND the emission rows, in the area the transition table
// takes over once the scores are done.
inline ChainViterbiPlan chain_viterbi_plan(int L, int SP, int NP, int K, int Kp, int c16, bool label_map = false, int rq = RG_RQ) {
    ChainViterbiPlan pl;
    const RegsLds rl = regs_lds(L, SP, NP, 0, 0, false, rq);
    pl.half = (rl.total + 3) & ~3;
    const size_t v = viterbi_hist_lds_bytes(K, Kp, SP, L, true);
    const int abT = (int)viterbi_products_floats(SP, L);
    const int sc_pieces = (L * Kp * 4 + 1023) / 1024, tr_pieces = (K * Kp * 4 + 1023) / 1024;
    pl.image_off = (int)viterbi_hist_floats(Kp, SP, L, true) + sc_pieces * 256;       // = the transition table's area (viterbi_hist_body)
    pl.image_pieces = ((K + 15) >> 4) * c16;
    pl.off0 = (abT + 3) & ~3;
    pl.off1 = pl.image_off + pl.image_pieces * 256;
    pl.lds_rows = pl.image_pieces <= tr_pieces && pl.off0 + pl.half <= pl.image_off &&
                  (size_t)(pl.off1 + pl.half) * sizeof(float) <= 158 * 1024;
    if (label_map) {
        pl.image_pieces = 0;
        pl.off0 = pl.image_off;
        pl.off1 = pl.off0 + pl.half;
        pl.lds_rows = (size_t)(pl.off1 + pl.half) * sizeof(float) <= 158 * 1024;
    }
    if (!pl.lds_rows) { pl.off0 = 0; pl.off1 = pl.half; pl.image_pieces = 0; }
    pl.bytes = (size_t)(pl.off1 + pl.half) * sizeof(float);
    if (v > pl.bytes) pl.bytes = v;
    return pl;
}

// RQ / D: the form of the chains (chain_regs_body): (RG_RQ, RG_D) for S <= 72; the wide form with a ring of D = 2 steps for
// 72 < S <= 108 (RQ = 8 / 9: 72 ring registers at most, inside the 128 a sixteen-wavefront workgroup leaves a lane)
template <bool MAXSR, bool NLX, int IB4, int RQ = RG_RQ, int D = RG_D>
__global__ void __launch_bounds__(2 * RG_WAVES * 64)
chain_viterbi_kernel(const RegsParams p, const ScoreParams sp, const ChainViterbiPlan pl, const int vthreads) {
    extern __shared__ __align__(16) float smem[];
    const int tid = (int)threadIdx.x;
    const int half = __builtin_amdgcn_readfirstlane(tid >> 9);               // 0: forward chain, 1: backward chain
    // the decode's wavefronts fetch their packed label-map words now: two VGPRs for the length of the chains instead of an L2
    // round trip in front of the first token (label_map.hip.h)
    unsigned lm_pk[2] = {0u, 0u};
    if (sp.lm.on && tid < vthreads) lm_load_packed(sp.lm, tid & 63, lm_pk[0], lm_pk[1]);
    if (pl.image_pieces) {                           // the output matrix's matrix-core image: in flight while the chains run
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(smem + pl.image_off));
        for (int k = wv; k < pl.image_pieces; k += 2 * RG_WAVES)
            lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, reinterpret_cast<const char *>(sp.OTm), lds0 + (unsigned)k * 1024u);
    }
    int b = 0;
    float *mine = smem + (half ? pl.off1 : pl.off0);
    // wavefront w of a workgroup sits on SIMD w % 4: the forward chain's six compute wavefronts (roles 0-5) are wavefronts 0-5,
    // on SIMDs 0 1 2 3 0 1; the backward chain's roles are rotated by two -- wavefronts 10-15, on SIMDs 2 3 0 1 2 3 -- so that
    // every SIMD carries three compute wavefronts (unrotated: four on SIMDs 0 and 1, two on 2 and 3)
    const int role_tid = half ? ((tid + 6 * 64) & (RG_WAVES * 64 - 1)) : tid;
    chain_regs_body<MAXSR, false, NLX, RQ, D>(p, mine, role_tid, 2 * (int)blockIdx.x + half, &b);
    __syncthreads();                                 // (s_waitcnt vmcnt(0) in front of it: the image / every stash row has landed)
    if (tid >= vthreads) return;
    const RegsLds rl = regs_lds(p.L, p.SP, RG_NWC * p.G, 0, 0, false, RQ);
    const int hist_off = rl.hist;
    // {length, flat offset} of the sequence: left in the forward half's misc words by its idle scorer wavefront (chain_regs_body).
    // Only where those words survive the decode's first writes: the rows-in-LDS plans (the halves lie behind the emission rows).
    const int *pre = pl.lds_rows ? reinterpret_cast<const int *>(smem + pl.off0 + rl.misc) + RGM_FOFF : nullptr;
    int pre_regs[2] = {0, 0};
    if (pre) { pre_regs[0] = pre[1]; pre_regs[1] = pre[0]; }
    if (pl.lds_rows)
        viterbi_hist_body<IB4, true>(sp, smem, tid, vthreads, b, smem + pl.off0 + hist_off, smem + pl.off1 + hist_off, true,
                                     sp.lm.on ? lm_pk : nullptr, pre_regs);
    else
        viterbi_hist_body<IB4, true>(sp, smem, tid, vthreads, b, nullptr, nullptr, false, sp.lm.on ? lm_pk : nullptr);
}


// launches chain_viterbi_kernel<MX, NX, IB4, RQ, D> for the tag count's IB4 (shared by the two translation units)
template <int RQ, int D>
int launch_chain_viterbi_form(const RegsParams &p, const ScoreParams &sp, const ChainViterbiPlan &plk, size_t lds, int vthreads,
                              bool maxsr, bool nlx, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const dim3 grid(p.B), block(2 * RG_WAVES * 64);
    int rc;
#define FARNN_LAUNCH_CV3(MX, NX, IB)                                                           \
    do {                                                                                       \
        if ((rc = raise_lds_limit(chain_viterbi_kernel<MX, NX, IB, RQ, D>, lds))) return rc;   \
        if (e0 && e1)                                                                          \
            hipExtLaunchKernelGGL((chain_viterbi_kernel<MX, NX, IB, RQ, D>), grid, block, (uint32_t)lds, s, e0, e1, 0, p, sp, plk, vthreads); \
        else                                                                                   \
            chain_viterbi_kernel<MX, NX, IB, RQ, D><<<grid, block, lds, s>>>(p, sp, plk, vthreads);   \
    } while (0)
#define FARNN_LAUNCH_CV2(MX, NX)                                                               \
    switch (viterbi_hist_ib4(sp.K)) {                                                          \
        case 1: FARNN_LAUNCH_CV3(MX, NX, 1); break;                                            \
        case 2: FARNN_LAUNCH_CV3(MX, NX, 2); break;                                            \
        case 3: FARNN_LAUNCH_CV3(MX, NX, 3); break;                                            \
        case 4: FARNN_LAUNCH_CV3(MX, NX, 4); break;                                            \
        default: return FARNN_ERANGE;                                                          \
    }
    if (maxsr) { if (nlx) { FARNN_LAUNCH_CV2(true, true) } else { FARNN_LAUNCH_CV2(true, false) } }
    else       { if (nlx) { FARNN_LAUNCH_CV2(false, true) } else { FARNN_LAUNCH_CV2(false, false) } }
#undef FARNN_LAUNCH_CV2
#undef FARNN_LAUNCH_CV3
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

// chain_viterbi_wide.hip
int launch_chain_viterbi_wide(const RegsParams &p, const ScoreParams &sp, const ChainViterbiPlan &plk, size_t lds, int vthreads,
                              bool maxsr, bool nlx, hipStream_t s, hipEvent_t e0, hipEvent_t e1);

}  // namespace farnn
