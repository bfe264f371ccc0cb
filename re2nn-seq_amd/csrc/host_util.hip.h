// Host-side helpers shared by the translation units of libfarnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>
#include "common.hip.h"

namespace farnn {

inline int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Switches a create call refuses instead of ignoring (include/farnn.h): forms that live in the A/B build only, asked of the
// production library, and switches of earlier rounds that no longer exist.  Checked where the switches are read: once per
// handle, in farnn_*_create / farnn_train_create -- never on the tagging path.
inline int check_env_switches() {
#if !defined(FARNN_AB)
    for (const char *name : {"FARNN_CV_ONE", "FARNN_CV_STASH", "FARNN_NODEST"})
        if (env_int(name, 0))
            return fail(FARNN_EINVAL, "%s is set: that form is compiled into the A/B build only (csrc/build.py --probes; load it with "
                                      "FARNN_LIB=.../libfarnn_hip_probes.so)%s", name);
#endif
    for (const char *name : {"FARNN_CV_WIDE", "FARNN_DECOMP_OLD"})
        if (env_int(name, 0))
            return fail(FARNN_EINVAL, "%s is set: that switch was removed (round 5; include/farnn.h lists the supported ones)%s", name);
    return FARNN_OK;
}

// first step of every farnn_*_create / farnn_train_create: refuse unsupported switches, then make `device` current
inline int select_device(int device) {
    if (int rc = check_env_switches()) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(FARNN_ENODEV, "no HIP device visible%s%s");
    if (device < 0 || device >= n) return fail(FARNN_EINVAL, "device index out of range%s%s");
    FARNN_HIP_TRY(hipSetDevice(device));
    return FARNN_OK;
}

// ---- the FARNN_* environment switches -------------------------------------------------------------------------------
// Every switch the library knows, read ONCE per handle when it is created (farnn_*_create / farnn_train_create): no call on
// the tagging path touches the environment.  SUPPORTED switches (include/farnn.h lists them) select between code paths the
// test suite holds to the same results; DIAGNOSTIC ones (ablations, geometry overrides of the tuning scripts) exist in the
// profiling build only (build.py --probes, -DFARNN_PROBES) -- the production library ignores them.
#define FARNN_TUNABLES(X)                                                                                      \
    /* supported */                                                                                            \
    X(NOFUSE, 0, true) X(FUSE, 0, true) X(NOREGS, 0, true) X(NOLABELMAP, 0, true) X(PREP, 0, true) X(NOSORT, 0, true)           \
    X(VITERBI_BP, 0, true) X(VITERBI_UNFUSED, 0, true) X(DECOMP_NOREGS, 0, true)                               \
    X(ROWS_NOREGS, 0, true) X(TRAIN_NOLDS, 0, true) X(TRAIN_NSEQ, 0, true)              \
    X(WIDE_UNPAIRED, 0, true) X(ROWS_LPR4, 0, true) X(ROWS_NOROUNDS, 0, true)                                  \
    /* A/B build only (the production library refuses them at create: check_env_switches) */                   \
    X(CV_ONE, 0, false) X(CV_STASH, 0, false) X(NODEST, 0, false)                                              \
    /* diagnostic: profiling build only */                                                                     \
    X(DBG, 0, false) X(KS, 3, false) X(RPG, 12, false) X(NLD, 4, false) X(NOFAST, 0, false)                    \
    X(CHAIN_HELPER, 0, false) X(HOST_EPOCH, 0, false) X(CV_NOSORT, 0, false) X(NOKZ, 0, false)                 \
    X(FUSE_SPIN, 4, false) X(SOLO_MARGIN, 24, false) X(DECOMP_FOUR, 0, false) X(ROWS_NSEQ, 0, false)
enum TunId {
#define FARNN_TUN_ENUM(name, dflt, prod) TUN_##name,
    FARNN_TUNABLES(FARNN_TUN_ENUM)
#undef FARNN_TUN_ENUM
    TUN_COUNT
};
struct Tunables {
    int v[TUN_COUNT];
    Tunables() {
#if defined(FARNN_PROBES)
        constexpr bool diag = true;
#else
        constexpr bool diag = false;
#endif
#define FARNN_TUN_READ(name, dflt, prod) v[TUN_##name] = ((prod) || diag) ? env_int("FARNN_" #name, dflt) : (dflt);
        FARNN_TUNABLES(FARNN_TUN_READ)
#undef FARNN_TUN_READ
    }
};
// the handle the current C-ABI call works for: tun(TUN_X) reads ITS snapshot (one caller thread per handle, include/farnn.h)
extern thread_local const Tunables *g_tun;
struct TunScope {
    const Tunables *prev;
    explicit TunScope(const Tunables *t) : prev(g_tun) { g_tun = t; }
    ~TunScope() { g_tun = prev; }
};
inline int tun(TunId id) {
    static const Tunables defaults_at_load;              // (outside any handle's call: the environment as the library was loaded)
    return (g_tun ? g_tun : &defaults_at_load)->v[id];
}

// Raise a kernel's dynamic-LDS limit (needed above 48 KiB).  The attribute is sticky, so it is set
// only when a kernel needs more than it was last given (a host API call per launch otherwise).
template <typename KernelT>
inline int raise_lds_limit(KernelT kern, size_t bytes) {
    static std::map<std::pair<int, const void *>, size_t> granted;    // (device, kernel) -> bytes
    static std::mutex granted_mu;                                     // (handles on different threads share the map)
    if (bytes > 160 * 1024) return fail(FARNN_ERANGE, "kernel needs more than 160 KiB of LDS%s%s");
    if (bytes <= 48 * 1024) return FARNN_OK;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void *fn = reinterpret_cast<const void *>(kern);
    std::lock_guard<std::mutex> lock(granted_mu);
    size_t &g = granted[{dev, fn}];
    if (bytes > g) {
        FARNN_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        g = bytes;
    }
    return FARNN_OK;
}

}  // namespace farnn
