// libfarnn_hip.so -- K1r, the register-fed recurrence with the score + decode stage beside it (chain_regs.hip.h), and its launcher.
// build-flags: -fno-slp-vectorize
// (the step's sixteen FMAs and its add tree stay scalar: hipcc's SLP pass packs them into v_pk_fma_f32 / v_pk_add_f32, which
//  one wavefront alone issues far slower than the scalar pairs -- MI355X_MICROARCH.md, packed f32 VALU: "an anti-lever")
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "common.hip.h"
#include "host_util.hip.h"
#include "chain_regs.hip.h"

namespace farnn {

int launch_chain_regs(const RegsParams &p, bool maxsr, bool score, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int NP = RG_NWC * p.G;
    const size_t lds = (size_t)regs_lds(p.L, p.SP, NP, p.sp.c16, p.sp.Kc, score, RG_RQ, score && bs_label_map_path(p.sp)).total * sizeof(float);
    const dim3 grid(2 * p.B), block(RG_WAVES * 64);
    int rc;
    // NLX: tanh / relu-tanh / sigmoid between the steps (the kernel's none / relu form has no branch in the step)
    const bool nlx = p.nl != FARNN_NL_NONE && p.nl != FARNN_NL_RELU;
    const bool lmo = score && bs_label_map_path(p.sp);                     // the label-map instantiation (no matrix-core tile code)
#define FARNN_LAUNCH_REGS4(MX, SC, NX, LM)                                                     \
    do {                                                                                       \
        if ((rc = raise_lds_limit(chain_regs_kernel<MX, SC, NX, LM>, lds))) return rc;         \
        if (e0 && e1)                                                                          \
            hipExtLaunchKernelGGL((chain_regs_kernel<MX, SC, NX, LM>), grid, block, (uint32_t)lds, s, e0, e1, 0, p); \
        else                                                                                   \
            chain_regs_kernel<MX, SC, NX, LM><<<grid, block, lds, s>>>(p);                    \
    } while (0)
#define FARNN_LAUNCH_REGS3(MX, SC, NX) do { if (SC && lmo) FARNN_LAUNCH_REGS4(MX, SC, NX, SC); else FARNN_LAUNCH_REGS4(MX, SC, NX, false); } while (0)
#define FARNN_LAUNCH_REGS(MX, SC) do { if (nlx) FARNN_LAUNCH_REGS3(MX, SC, true); else FARNN_LAUNCH_REGS3(MX, SC, false); } while (0)
    if (maxsr) { if (score) FARNN_LAUNCH_REGS(true, true); else FARNN_LAUNCH_REGS(true, false); }
    else       { if (score) FARNN_LAUNCH_REGS(false, true); else FARNN_LAUNCH_REGS(false, false); }
#undef FARNN_LAUNCH_REGS
#undef FARNN_LAUNCH_REGS3
#undef FARNN_LAUNCH_REGS4
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

}  // namespace farnn
