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
// Compiled into the A/B build only (build.py --probes: -DFARNN_AB): two launches beat this form at every shape measured,
// so the production library does not carry its 48 kernels; FARNN_CV_ONE=1 there fails with a message that says where the form lives.
#if defined(FARNN_AB)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "chain_viterbi.hip.h"

namespace farnn {

bool chain_viterbi_fits(int L, int SP, int NP, int K, int Kp, bool label_map, int rq) {
    const int ib4 = viterbi_hist_ib4(K);
    if (ib4 < 1 || ib4 > 4) return false;
    if (rq != RG_RQ && rq > 9) return false;             // the wide form at 128 VGPRs: a ring of two steps x at most nine rows
    const ChainViterbiPlan pl = chain_viterbi_plan(L, SP, NP, K, Kp, (SP + 15) / 16, label_map, rq);
    return pl.bytes <= 158 * 1024 && viterbi_hist_threads(K) <= 2 * RG_WAVES * 64;
}

int launch_chain_viterbi(const RegsParams &p_in, const ScoreParams &sp, bool maxsr, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int NP = RG_NWC * p_in.G;
    const ChainViterbiPlan pl = chain_viterbi_plan(p_in.L, p_in.SP, NP, sp.K, sp.Kp, sp.c16, sp.lm.on != 0, p_in.RQ);
    RegsParams p = p_in;
    p.sp = sp;                                       // (the chains' idle wavefront works out the flat-output offset: it needs len / flat / offs)
    if (pl.lds_rows && !tun(TUN_CV_STASH)) { p.A = nullptr; p.Bk = nullptr; }     // no stash: the rows stay in LDS
    // (Launch order: longest sequence first, as selected in the kernel.  A workgroup owns a whole sequence and a compute unit holds
    //  one workgroup, so for a lone launch of at most a sequence per compute unit the order changes nothing and the selection costs
    //  0.8 us of 84 -- FARNN_CV_NOSORT=1 --, but with a second batch in flight on another stream its workgroups fill the compute
    //  units as this launch's short sequences leave them, and longest-first packs that 22 % tighter: 1.71e8 against 1.40e8 tokens/s.)
    if (tun(TUN_CV_NOSORT)) { p.sort = 0; p.order = nullptr; }
    ChainViterbiPlan plk = pl;
    if (p.A) plk.lds_rows = 0;                       // (diagnostic: the same layout, rows through the stash)
    const size_t lds = pl.bytes;
    const int vthreads = viterbi_hist_threads(sp.K);
    const bool nlx = p.nl != FARNN_NL_NONE && p.nl != FARNN_NL_RELU;
    if (p.RQ != RG_RQ) {                             // the wide form: its ring is two steps deep here (128 VGPRs)
        p.D = 2;
        return launch_chain_viterbi_wide(p, sp, plk, lds, vthreads, maxsr, nlx, s, e0, e1);
    }
    return launch_chain_viterbi_form<RG_RQ, RG_D>(p, sp, plk, lds, vthreads, maxsr, nlx, s, e0, e1);
}

}  // namespace farnn
#endif  // FARNN_AB
