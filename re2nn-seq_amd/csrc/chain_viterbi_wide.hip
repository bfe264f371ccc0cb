// libfarnn_hip.so -- K1v (chain_viterbi.hip.h: both chains of a sequence, its scores and its CRF decode in ONE launch) for automata
// of 73..108 states: the chains in the wide form (chain_wide.hip.h) with a ring of two steps (the reference's 104-state automata:
// RE.py:56-60; model_onehot.py:372-426 -> model_decompose.py:351-356 -> crf.py:102-195).
// build-flags: -fno-slp-vectorize
// Compiled into the A/B build only (build.py --probes: -DFARNN_AB): two launches beat this form at every shape measured,
// so the production library does not carry its 48 kernels; FARNN_CV_ONE=1 there fails with a message that says where the form lives.
#if defined(FARNN_AB)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "chain_viterbi.hip.h"

namespace farnn {

int launch_chain_viterbi_wide(const RegsParams &p, const ScoreParams &sp, const ChainViterbiPlan &plk, size_t lds, int vthreads,
                              bool maxsr, bool nlx, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if (p.G != RGW_G || p.D != 2) return fail(FARNN_EINVAL, "chain_viterbi_wide: geometry%s%s");
    if (p.RQ == 8) return launch_chain_viterbi_form<8, 2>(p, sp, plk, lds, vthreads, maxsr, nlx, s, e0, e1);
    if (p.RQ == 9) return launch_chain_viterbi_form<9, 2>(p, sp, plk, lds, vthreads, maxsr, nlx, s, e0, e1);
    return fail(FARNN_EINVAL, "chain_viterbi_wide: no instantiation for this ring width%s%s");
}

}  // namespace farnn
#endif  // FARNN_AB
