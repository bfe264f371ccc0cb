// libfarnn_hip.so -- K1r's wide form (chain_wide.hip.h): the register-fed recurrence with the score + decode stage beside it for
// automata of 73..128 states (the reference's 104-state SNIPS-BIO / ATIS-ZH-BIO automata, RE.py:56-60), and its launcher.
// build-flags: -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "common.hip.h"
#include "host_util.hip.h"
#include "chain_regs.hip.h"

namespace farnn {

int launch_chain_wide(const RegsParams &p, bool maxsr, bool score, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const size_t lds = (size_t)regs_lds(p.L, p.SP, RGW_NP, p.sp.c16, p.sp.Kc, score, p.RQ, score && bs_label_map_path(p.sp)).total * sizeof(float);
    const dim3 grid(2 * p.B), block(RG_WAVES * 64);
    int rc;
    const bool nlx = p.nl != FARNN_NL_NONE && p.nl != FARNN_NL_RELU;
    if (p.G != RGW_G || p.D != 4 || p.RPG > p.RQ) return fail(FARNN_EINVAL, "chain_wide: geometry%s%s");
#define FARNN_LAUNCH_WIDE5(MX, SC, NX, RQ_, D_)                                                \
    do {                                                                                       \
        if ((rc = raise_lds_limit(chain_wide_kernel<MX, SC, NX, RQ_, D_>, lds))) return rc;    \
        if (e0 && e1)                                                                          \
            hipExtLaunchKernelGGL((chain_wide_kernel<MX, SC, NX, RQ_, D_>), grid, block, (uint32_t)lds, s, e0, e1, 0, p); \
        else                                                                                   \
            chain_wide_kernel<MX, SC, NX, RQ_, D_><<<grid, block, lds, s>>>(p);                \
    } while (0)
#define FARNN_LAUNCH_WIDE3(MX, SC, NX)                                                         \
    do {                                                                                       \
        if (p.RQ == 8) FARNN_LAUNCH_WIDE5(MX, SC, NX, 8, 4);                                   \
        else if (p.RQ == 9) FARNN_LAUNCH_WIDE5(MX, SC, NX, 9, 4);                              \
        else if (p.RQ == 11) FARNN_LAUNCH_WIDE5(MX, SC, NX, 11, 4);                            \
        else return fail(FARNN_EINVAL, "chain_wide: no instantiation for this ring width%s%s"); \
    } while (0)
#define FARNN_LAUNCH_WIDE(MX, SC) do { if (nlx) FARNN_LAUNCH_WIDE3(MX, SC, true); else FARNN_LAUNCH_WIDE3(MX, SC, false); } while (0)
    if (maxsr) { if (score) FARNN_LAUNCH_WIDE(true, true); else FARNN_LAUNCH_WIDE(true, false); }
    else       { if (score) FARNN_LAUNCH_WIDE(false, true); else FARNN_LAUNCH_WIDE(false, false); }
#undef FARNN_LAUNCH_WIDE
#undef FARNN_LAUNCH_WIDE3
#undef FARNN_LAUNCH_WIDE5
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

int launch_chain_wide_paired(const RegsParams &p, bool maxsr, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const size_t lds = (size_t)regs_lds(p.L, p.SP, RGW_NP, p.sp.c16, p.sp.Kc, true, p.RQ, true).total * sizeof(float);
    const dim3 grid(2 * p.B), block(RG_WAVES * 64);
    int rc;
    const bool nlx = p.nl != FARNN_NL_NONE && p.nl != FARNN_NL_RELU;
    if (p.G != RGW_G || p.D != 2 || !p.pair || p.RPG > p.RQ || !bs_label_map_path(p.sp)) return fail(FARNN_EINVAL, "chain_wide_paired: geometry%s%s");
#define FARNN_LAUNCH_WP3(MX, NX, RQ_)                                                          \
    do {                                                                                       \
        if ((rc = raise_lds_limit(chain_wide_paired_kernel<MX, NX, RQ_>, lds))) return rc;     \
        if (e0 && e1)                                                                          \
            hipExtLaunchKernelGGL((chain_wide_paired_kernel<MX, NX, RQ_>), grid, block, (uint32_t)lds, s, e0, e1, 0, p); \
        else                                                                                   \
            chain_wide_paired_kernel<MX, NX, RQ_><<<grid, block, lds, s>>>(p);                 \
    } while (0)
#define FARNN_LAUNCH_WP2(MX, NX)                                                               \
    do {                                                                                       \
        if (p.RQ == 8) FARNN_LAUNCH_WP3(MX, NX, 8);                                            \
        else if (p.RQ == 9) FARNN_LAUNCH_WP3(MX, NX, 9);                                       \
        else return fail(FARNN_EINVAL, "chain_wide_paired: no instantiation for this ring width%s%s"); \
    } while (0)
    if (maxsr) { if (nlx) FARNN_LAUNCH_WP2(true, true); else FARNN_LAUNCH_WP2(true, false); }
    else       { if (nlx) FARNN_LAUNCH_WP2(false, true); else FARNN_LAUNCH_WP2(false, false); }
#undef FARNN_LAUNCH_WP2
#undef FARNN_LAUNCH_WP3
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

}  // namespace farnn
