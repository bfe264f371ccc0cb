// libfarnn_hip.so -- config 4's tagging step as ONE launch: the two chains of a sequence and, behind them, its scores and CRF
// decode in the same workgroup (model_onehot.py:372-426 -> model_decompose.py:351-356 -> crf.py:102-195).
// build-flags: -fno-slp-vectorize
//
// A workgroup of SIXTEEN wavefronts owns a sequence: wavefronts 0-7 run chain_regs_body for the forward chain, 8-15 for the
// backward chain -- the same code, LDS carve and register budget (128 VGPRs) as two chain_regs workgroups sharing a compute
// unit, which is what a compute unit holds in the two-launch form too.  The chains of one sequence are equally long, so they
// end together; their state rows are in the stash (plain stores by the two writer wavefronts), drained and made visible to the
// workgroup by the barrier that follows.  Then the first viterbi_hist_threads(K) threads run viterbi_hist_body<IB4, fused>
// (products -> scores on the matrix cores -> forward pass -> back-trace) over the LDS the chains no longer need; the rest
// leave.  No inter-workgroup hand-off, no epoch, no progress words: the step is graph-capturable.
// What the two-launch form has and this one has not: a short sequence's compute unit idles while the long ones finish (one
// sequence per compute unit per round either way for the decode, but the stand-alone recurrence pairs long with short).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

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
static ChainViterbiPlan chain_viterbi_plan(int L, int SP, int NP, int K, int Kp, int c16, bool label_map = false) {
    ChainViterbiPlan pl;
    const RegsLds rl = regs_lds(L, SP, NP, 0, 0, false);
    pl.half = (rl.total + 3) & ~3;
    const size_t v = viterbi_hist_lds_bytes(K, Kp, SP, L, true);
    const int abT = SP * (((L + 3) & ~3) + 16);
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

template <bool MAXSR, bool NLX, int IB4>
__global__ void __launch_bounds__(2 * RG_WAVES * 64)
chain_viterbi_kernel(const RegsParams p, const ScoreParams sp, const ChainViterbiPlan pl, const int vthreads) {
    extern __shared__ __align__(16) float smem[];
    const int tid = (int)threadIdx.x;
    const int half = __builtin_amdgcn_readfirstlane(tid >> 9);               // 0: forward chain, 1: backward chain
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
    chain_regs_body<MAXSR, false, NLX>(p, mine, role_tid, 2 * (int)blockIdx.x + half, &b);
    __syncthreads();                                 // (s_waitcnt vmcnt(0) in front of it: the image / every stash row has landed)
    if (tid >= vthreads) return;
    const int hist_off = regs_lds(p.L, p.SP, RG_NWC * p.G, 0, 0, false).hist;
    if (pl.lds_rows)
        viterbi_hist_body<IB4, true>(sp, smem, tid, vthreads, b, smem + pl.off0 + hist_off, smem + pl.off1 + hist_off, true);
    else
        viterbi_hist_body<IB4, true>(sp, smem, tid, vthreads, b);
}

bool chain_viterbi_fits(int L, int SP, int NP, int K, int Kp, bool label_map) {
    const int ib4 = viterbi_hist_ib4(K);
    if (ib4 < 1 || ib4 > 4) return false;
    const ChainViterbiPlan pl = chain_viterbi_plan(L, SP, NP, K, Kp, (SP + 15) / 16, label_map);
    return pl.bytes <= 158 * 1024 && viterbi_hist_threads(K) <= 2 * RG_WAVES * 64;
}

int launch_chain_viterbi(const RegsParams &p_in, const ScoreParams &sp, bool maxsr, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int NP = RG_NWC * p_in.G;
    const ChainViterbiPlan pl = chain_viterbi_plan(p_in.L, p_in.SP, NP, sp.K, sp.Kp, sp.c16, sp.lm.on != 0);
    RegsParams p = p_in;
    if (pl.lds_rows && !env_int("FARNN_CV_STASH", 0)) { p.A = nullptr; p.Bk = nullptr; }     // no stash: the rows stay in LDS
    // (Launch order: longest sequence first, as selected in the kernel.  A workgroup owns a whole sequence and a compute unit holds
    //  one workgroup, so for a lone launch of at most a sequence per compute unit the order changes nothing and the selection costs
    //  0.8 us of 84 -- FARNN_CV_NOSORT=1 --, but with a second batch in flight on another stream its workgroups fill the compute
    //  units as this launch's short sequences leave them, and longest-first packs that 22 % tighter: 1.71e8 against 1.40e8 tokens/s.)
    if (env_int("FARNN_CV_NOSORT", 0)) { p.sort = 0; p.order = nullptr; }
    ChainViterbiPlan plk = pl;
    if (p.A) plk.lds_rows = 0;                       // (diagnostic: the same layout, rows through the stash)
    const size_t lds = pl.bytes;
    const int vthreads = viterbi_hist_threads(sp.K);
    const dim3 grid(p.B), block(2 * RG_WAVES * 64);
    const bool nlx = p.nl != FARNN_NL_NONE && p.nl != FARNN_NL_RELU;
    int rc;
#define FARNN_LAUNCH_CV3(MX, NX, IB)                                                           \
    do {                                                                                       \
        if ((rc = raise_lds_limit(chain_viterbi_kernel<MX, NX, IB>, lds))) return rc;          \
        if (e0 && e1)                                                                          \
            hipExtLaunchKernelGGL((chain_viterbi_kernel<MX, NX, IB>), grid, block, (uint32_t)lds, s, e0, e1, 0, p, sp, plk, vthreads); \
        else                                                                                   \
            chain_viterbi_kernel<MX, NX, IB><<<grid, block, lds, s>>>(p, sp, plk, vthreads);   \
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

}  // namespace farnn
