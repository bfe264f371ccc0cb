// libfarnn_hip.so -- K1r, the register-fed recurrence with the score + decode stage beside it (chain_regs.hip.h), and its launcher.
// build-flags: -fno-slp-vectorize
// build-check: ring-registers ELb1EEEvNS_10RegsParamsE
// (the destination-split kernels -- last template argument true -- keep a ring of in-flight global loads in ordinary asm outputs
//  (chain_dest.hip.h): csrc/build.py fails the build if the compiler ever copies or spills one of them while its load is in flight)
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
    const bool dest = p.dest && !maxsr;                                   // the destination-split compute wavefronts (chain_dest.hip.h)
    const size_t lds = (size_t)regs_lds(p.L, p.SP, NP, p.sp.c16, p.sp.Kc, score, RG_RQ, score && bs_label_map_path(p.sp), dest).total * sizeof(float);
    const dim3 grid(2 * p.B), block(RG_WAVES * 64);
    int rc;
    // NLX: tanh / relu-tanh / sigmoid between the steps (the kernel's none / relu form has no branch in the step)
    const bool nlx = p.nl != FARNN_NL_NONE && p.nl != FARNN_NL_RELU;
    const bool lmo = score && bs_label_map_path(p.sp);                     // the label-map instantiation (no matrix-core tile code)
#define FARNN_LAUNCH_REGS5(MX, SC, NX, LM, DS)                                                 \
    do {                                                                                       \
        if ((rc = raise_lds_limit(chain_regs_kernel<MX, SC, NX, LM, DS>, lds))) return rc;     \
        if (e0 && e1)                                                                          \
            hipExtLaunchKernelGGL((chain_regs_kernel<MX, SC, NX, LM, DS>), grid, block, (uint32_t)lds, s, e0, e1, 0, p); \
        else                                                                                   \
            chain_regs_kernel<MX, SC, NX, LM, DS><<<grid, block, lds, s>>>(p);                \
    } while (0)
    // (the sum semiring's source-split compute wavefronts -- round 3's, FARNN_NODEST=1 -- live in the A/B build: six kernels the
    //  production library's dispatch cannot reach; the max semiring keeps the source split, its only form)
#if defined(FARNN_AB)
#define FARNN_LAUNCH_REGS4(MX, SC, NX, LM) do { if (!MX && dest) FARNN_LAUNCH_REGS5(MX, SC, NX, LM, !MX); else FARNN_LAUNCH_REGS5(MX, SC, NX, LM, false); } while (0)
#else
#define FARNN_LAUNCH_REGS4(MX, SC, NX, LM)                                                     \
    do {                                                                                       \
        if (!MX && dest) FARNN_LAUNCH_REGS5(MX, SC, NX, LM, !MX);                              \
        else if constexpr (MX) FARNN_LAUNCH_REGS5(MX, SC, NX, LM, false);                      \
        else return fail(FARNN_EINVAL, "FARNN_NODEST=1: the source-split compute wavefronts of the sum semiring are compiled into the A/B " \
                                       "build only (FARNN_LIB=.../libfarnn_hip_probes.so, csrc/build.py --probes)%s%s"); \
    } while (0)
#endif
#define FARNN_LAUNCH_REGS3(MX, SC, NX) do { if (SC && lmo) FARNN_LAUNCH_REGS4(MX, SC, NX, SC); else FARNN_LAUNCH_REGS4(MX, SC, NX, false); } while (0)
#define FARNN_LAUNCH_REGS(MX, SC) do { if (nlx) FARNN_LAUNCH_REGS3(MX, SC, true); else FARNN_LAUNCH_REGS3(MX, SC, false); } while (0)
    if (maxsr) { if (score) FARNN_LAUNCH_REGS(true, true); else FARNN_LAUNCH_REGS(true, false); }
    else       { if (score) FARNN_LAUNCH_REGS(false, true); else FARNN_LAUNCH_REGS(false, false); }
#undef FARNN_LAUNCH_REGS
#undef FARNN_LAUNCH_REGS3
#undef FARNN_LAUNCH_REGS4
#undef FARNN_LAUNCH_REGS5
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

}  // namespace farnn

#if defined(FARNN_PROBES)
// profiling build only (not part of include/farnn.h): the per-workgroup stamps of the last chain_regs_kernel launch under FARNN_DBG=2048
extern "C" int farnn_debug_wg_stamps(long long *out, int n_workgroups) {
    if (!out || n_workgroups < 0 || n_workgroups > farnn::WG_STAMP_MAX) return FARNN_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return FARNN_EIO;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(farnn::g_wg_stamps), sizeof(long long) * 16 * (size_t)n_workgroups) == hipSuccess ? FARNN_OK : FARNN_EIO;
}
#endif
