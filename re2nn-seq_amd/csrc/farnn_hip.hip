// libfarnn_hip.so -- C-ABI entry points (include/farnn.h) of the MI355X-native FA-RNN tagging path.
// gfx950 only; no CPU fallback lives here (the CPU oracle is test infrastructure under oracle/).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <new>
#include <utility>
#include <vector>

#include "common.hip.h"
#include "host_util.hip.h"
#include "chain.hip.h"
#include "score_decode.hip.h"
#include "layout.hip.h"
#include "fst4_score.hip.h"
#include "decomp_chain.hip.h"
#include "decomp1_score.hip.h"
#include "decomp_rows.hip.h"
#include "compact_tag.hip.h"
#include "decomp_regs.hip.h"
#include "compact.hip.h"
#include "host_util.hip.h"
#include "chain_regs_params.hip.h"

namespace farnn {
thread_local char g_err[512] = "";
thread_local const Tunables *g_tun = nullptr;
}
using namespace farnn;

enum { KIND_IFST = 2, KIND_IND1 = 1, KIND_FST4 = 0, KIND_DECOMP = 12, KIND_DECOMP1 = 11, KIND_DECOMP0 = 10 };
enum { KERN_CHAIN = 0, KERN_SCORE = 1, KERN_PREP = 2, KERN_COUNT = 3 };

struct Prof {
    std::vector<hipEvent_t> ev[KERN_COUNT];   // (start, stop) pairs
    std::vector<hipEvent_t> pool;             // events created ahead of the timed region (farnn_set_profiling)
    double ms[KERN_COUNT] = {0, 0, 0};
    long long n[KERN_COUNT] = {0, 0, 0};
    hipEvent_t get() {
        hipEvent_t e = nullptr;
        if (!pool.empty()) { e = pool.back(); pool.pop_back(); return e; }
        return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    }
};

struct farnn_model {
    Tunables tun;                           // the FARNN_* switches as they stood when the handle was created (host_util.hip.h)
    int kind = 0, device = 0;
    int V = 0, S = 0, SP = 0, C = 0, K = 0, Kp = 0, Kc = 0, R = 0, Rp = 0;
    int nl = 0, semiring = 0, o_idx = 0, use_crf = 0, farnn_gate = 0, mask_by_output = 0;
    float threshold = 0.5f, sig_k = 1.0f;
    // device-resident, library-owned weights
    float *Mf = nullptr, *Mb = nullptr;     // chain blocks [V][S][SP] (+ transposed)
    u64 *bmF = nullptr, *bmB = nullptr, *bmWF = nullptr, *bmWB = nullptr;   // compact form: bit-packed blocks (compact.hip.h)
    int bmNS = 0;                           // 64-bit words per bitmap row; 0: no compact form
    u64 *bmMF = nullptr, *bmMB = nullptr, *bmXF = nullptr, *bmXB = nullptr;   // K1t's planes: T | W, T & W (merge_planes_kernel; S <= 128)
    unsigned *bmTok = nullptr;              // [V] block offset | second-plane flag
    bool compact_on = false;                // farnn_set_compact: the recurrence walks the bitmaps instead of the dense blocks
    float *Ms = nullptr;                    // ind1: unmasked blocks for scoring
    float *A4 = nullptr;                    // fst4: [V][C][S][SP] premixed T4+W4
    float *Oten = nullptr;                  // ind1: [C][S][SP]
    float *o = nullptr, *h0 = nullptr, *hT = nullptr;
    float *OT = nullptr, *P = nullptr, *tr = nullptr;
    float *OTm = nullptr; int c16 = 0;       // matrix-core image of OT for score_tiles (ot_to_mfma_kernel)
    LabelMap lm = {nullptr, 0, 0, -1, 0.0f, 0, 0, 0.0f}; // the output matrix as a label map, when it is one (label_map.hip.h)
    DecompWeights dw;                       // decomposed model weights
    DecompRowsPack rows;                    // packed rows of the K12 rows kernel (sum semiring)
    int RO = 0, ROp = 0;                    // decomposed independent=1: output factors
    float *d1_S1o = nullptr, *d1_S2o = nullptr, *d1_CoutT = nullptr;
    float *d1_BSSp = nullptr;               // [V][MT][KQ4][64][4] per-word bss = sum_r S1 S2 v + W in MFMA operand order
    float *d1_S1oP = nullptr;               // [MT][NT][64][4] S1o in MFMA accumulator order
    float *d1_S2oP = nullptr;               // [KQ4][NT][64][4] S2o in MFMA operand order
    int n_cu = 0;                           // compute units of the device (persistent launches)
    int RW = 0, RWp = 0;                    // decomposed independent=0: wildcard factors + label factor
    float *d0_Vgen = nullptr, *d0_CT = nullptr, *d0_S1w = nullptr, *d0_S2w = nullptr, *d0_CwT = nullptr;
    // workspace
    float *A = nullptr, *Bk = nullptr, *crf_scores = nullptr;
    float *d1_br = nullptr;                 // [B*L][MT][NT*16] per-row-tile partial output-rank vectors (decomposed independent=1)
    int64_t *offs = nullptr;
    int *order = nullptr;
    int wsB = 0, wsL = 0;                   // workspace CAPACITY: sequences, positions
    int curL = 0;                           // the current call's L: every stride of the workspace arrays
    ChainGeom geom;
    RegsGeom rgeom;                         // geometry of the register-fed recurrence kernel (chain_regs.hip.h); rgeom.ok: usable
    unsigned long long *hs = nullptr;       // hand-off words of that kernel: progress [2][B], arrival [B] (64-bit each)
    size_t hs_bytes = 0;
    unsigned epoch_u = 0;                   // diagnostic FARNN_HOST_EPOCH=1: the round-3 host-side epoch
    bool last_regs = false;                 // the last recurrence ran on chain_regs_kernel
    bool last_lm_score = false;             // the last stand-alone score launch was label_map_score_kernel (K2l)
    int chain_ks = 3;
    bool prep_in_kernel = false, sort_in_kernel = false;
    bool dense_decomp = false;              // decomposed model served by dense per-word blocks + chain_kernel
    bool order_valid = false;
    bool last_wave = false;                 // the last decomposed recurrence ran on decomp_regs_kernel
    bool last_fused = false;                // the last farnn_tag ran the single-launch form (chain + score/decode epilogue)
    int profiling = 0;          // 0 off, N>0: time every N-th farnn_tag call
    long long calls = 0;
    int prof_this_call = 0;
    Prof prof;
    std::vector<void *> owned;              // everything to hipFree at destroy
    // host-buffer path (farnn_tag_host_*): pinned staging + device twins per in-flight batch, three streams
    struct HostSlot {
        int64_t *x_pin = nullptr, *flat_pin = nullptr;      // [x | lengths] staged together; flat predictions
        int64_t *x_dev = nullptr, *flat_dev = nullptr, *x_map = nullptr;   // device copy of x; device views of the pinned buffers
        size_t capN = 0, capB = 0;
        long long total = 0;
        hipEvent_t ev_out = nullptr;
        bool busy = false;
        unsigned gen = 0;                   // submits this slot has seen: a ticket = slot | gen << 8, so a stale ticket cannot consume a newer batch
    } hslot[FARNN_HOST_SLOTS];
    hipStream_t hs_run = nullptr;
    int hnext = 0;
    // stream ordering of the handle's ONE workspace (stash, hand-off words, launch order): a call on another stream than the
    // previous call's waits for that call's work first
    hipStream_t last_stream = nullptr;
    bool have_last = false, multi_stream = false;
    hipEvent_t ev_order = nullptr;
};

// ---- small helpers ---------------------------------------------------------------------------
static int dev_alloc(farnn_model *m, void **p, size_t bytes) {
    // +1 KiB slack: LDS-DMA moves whole 1 KiB pieces, the last piece of a table may run past its end
    FARNN_HIP_TRY(hipMalloc(p, bytes + 1024));
    m->owned.push_back(*p);
    return FARNN_OK;
}

// copy (host or device) floats into a fresh device buffer of `n_alloc` floats (zero padded)
static int dev_upload(farnn_model *m, float **dst, const float *src, size_t n, size_t n_alloc,
                      int on_device) {
    int rc = dev_alloc(m, (void **)dst, n_alloc * sizeof(float));
    if (rc) return rc;
    FARNN_HIP_TRY(hipMemset(*dst, 0, n_alloc * sizeof(float)));
    if (src && n)
        FARNN_HIP_TRY(hipMemcpy(*dst, src, n * sizeof(float),
                                on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    return FARNN_OK;
}

// a temporary device view of a (host|device) array
struct TmpDev {
    const float *p = nullptr;
    float *owned = nullptr;
    ~TmpDev() { if (owned) (void)hipFree(owned); }
    int init(const float *src, size_t n, int on_device) {
        if (on_device || !src) { p = src; return FARNN_OK; }
        FARNN_HIP_TRY(hipMalloc((void **)&owned, n * sizeof(float)));
        FARNN_HIP_TRY(hipMemcpy(owned, src, n * sizeof(float), hipMemcpyHostToDevice));
        p = owned;
        return FARNN_OK;
    }
};

// rows x cols (row-major, host|device) -> device [rows_alloc][cols_p], zero padded
static int upload_padded(farnn_model *m, float **dst, const float *src, int rows, int cols,
                         int rows_alloc, int cols_p, int on_device) {
    int rc = dev_alloc(m, (void **)dst, (size_t)rows_alloc * cols_p * sizeof(float));
    if (rc) return rc;
    FARNN_HIP_TRY(hipMemset(*dst, 0, (size_t)rows_alloc * cols_p * sizeof(float)));
    if (src)
        FARNN_HIP_TRY(hipMemcpy2D(*dst, (size_t)cols_p * 4, src, (size_t)cols * 4, (size_t)cols * 4,
                                  rows, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    return FARNN_OK;
}

// transposed upload: src [rows][cols] -> dst [cols][rows_p]
static int upload_transposed(farnn_model *m, float **dst, const float *src, int rows, int cols,
                             int rows_p, int on_device) {
    TmpDev t;
    int rc = t.init(src, (size_t)rows * cols, on_device);
    if (rc) return rc;
    rc = dev_alloc(m, (void **)dst, (size_t)cols * rows_p * sizeof(float));
    if (rc) return rc;
    FARNN_HIP_TRY(hipMemset(*dst, 0, (size_t)cols * rows_p * sizeof(float)));
    int n = rows * cols;
    transpose_pad_kernel<<<(n + 255) / 256, 256>>>(t.p, *dst, rows, cols, rows_p);
    FARNN_HIP_TRY(hipGetLastError());
    FARNN_HIP_TRY(hipDeviceSynchronize());
    return FARNN_OK;
}


// the dense-block recurrence's geometries: the ring kernel's (chain.hip.h) and the register-fed kernel's (chain_regs.hip.h);
// the blocks get enough zero rows for either
static void pick_chain_geometry(farnn_model *m) {
    m->geom = chain_geometry(m->S, tun(TUN_RPG), tun(TUN_NLD));
    m->rgeom = regs_geometry(m->S);
    if (m->rgeom.SP != m->geom.SP) m->rgeom.ok = false;
    if (m->rgeom.ok && m->rgeom.rows > m->geom.SR) m->geom.SR = m->rgeom.rows;
}

static int default_crf_transitions(std::vector<float> &tr, int K) {
    // CRF.__init__ (crf.py:39-46): zeros, [:,START]=-1e4, [STOP,:]=-1e4
    tr.assign((size_t)K * K, 0.0f);
    for (int i = 0; i < K; i++) tr[(size_t)i * K + (K - 2)] = -10000.0f;
    for (int j = 0; j < K; j++) tr[(size_t)(K - 1) * K + j] = -10000.0f;
    return FARNN_OK;
}

static int setup_priority(farnn_model *m, const float *P, int on_device) {
    // P is [K][K] (already expanded, priority.py:6-18); stored [K][Kc]
    if (!P) return FARNN_OK;
    return upload_padded(m, &m->P, P, m->K, m->K, m->K, m->Kc, on_device);
}

static int setup_crf(farnn_model *m, const float *crf_trans, int on_device) {
    if (!m->use_crf) return FARNN_OK;
    std::vector<float> dflt;
    if (!crf_trans) { default_crf_transitions(dflt, m->K); crf_trans = dflt.data(); on_device = 0; }
    // stored transposed (trT[j][i] = tr[i][j]) so the Viterbi inner loop walks contiguous memory
    return upload_transposed(m, &m->tr, crf_trans, m->K, m->K, m->Kp, on_device);
}

// ---- compact form of a 0/1 automaton (compact.hip.h): bit-packed blocks beside (or instead of) the dense ones ----------
// the matrix-core image of the transposed output matrix (call once m->OT is final)
static int build_ot_image(farnn_model *m) {
    int rc;
    m->c16 = (m->S + 15) / 16;
    const long long n = (long long)(m->Kc / 16) * m->c16 * 256;
    if ((rc = dev_alloc(m, (void **)&m->OTm, (size_t)n * 4))) return rc;
    ot_to_mfma_kernel<<<(unsigned)((n + 255) / 256), 256>>>(m->OT, m->OTm, m->S, m->Kc, m->c16);
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

// The output matrix as a label map (label_map.hip.h): every state at most one label, weight exactly 1, at most 128 labelled
// states.  Read back from the final OT[S][Kc] (a few KB), sorted by (label, state) on the host, uploaded as one table.
static int build_label_map(farnn_model *m) {
    m->lm.on = 0;
    if (!m->OT || m->S > 1024 || tun(TUN_NOLABELMAP)) return FARNN_OK;
    std::vector<float> ot((size_t)m->S * m->Kc);
    FARNN_HIP_TRY(hipMemcpy(ot.data(), m->OT, ot.size() * 4, hipMemcpyDeviceToHost));
    std::vector<std::pair<int, int>> pos;                // (label, state)
    for (int s = 0; s < m->S; s++) {
        int lab = -1;
        for (int c = 0; c < m->K; c++) {
            const float v = ot[(size_t)s * m->Kc + c];
            if (v == 0.0f) continue;
            if (v != 1.0f || lab >= 0) return FARNN_OK;  // a weight, or a second label: the matrix form
            lab = c;
        }
        if (lab >= 0) pos.push_back({lab, s});
    }
    const int n = (int)pos.size();
    if (n > LM_MAXS || m->S > 256 || m->K > 510) return FARNN_OK;      // (8 bits of state, 9 of label per packed word)
    std::sort(pos.begin(), pos.end());
    const int clampcol = m->use_crf ? m->K - 3 : m->K - 1;     // model_decompose.py:353 / model_onehot.py:166
    std::vector<unsigned> tab(128, 0u);
    int lb[128];
    for (int j = 0; j < 128; j++) lb[j] = j < n ? pos[j].first : m->K + j;      // pads: distinct, above every label
    for (int j = 0; j < 128; j++) {
        unsigned wd = j < n ? ((unsigned)pos[j].second | ((unsigned)lb[j] << LM_LB_SHIFT)) : (0x1ffu << LM_LB_SHIFT);
        const int base = j & ~63, r = (j & 63) >> 4;
        const int dd[4] = {1, 2, 4, 8};
        for (int d = 0; d < 4; d++)
            if ((j & 15) >= dd[d] && lb[j - dd[d]] == lb[j]) wd |= 1u << (LM_CF_SHIFT + d);
        if ((r == 1 || r == 3) && lb[base + 16 * r - 1] == lb[j]) wd |= 1u << (LM_CF_SHIFT + 4);
        if ((r == 2 || r == 3) && lb[base + 31] == lb[j]) wd |= 1u << (LM_CF_SHIFT + 5);
        if (j >= 64 && lb[j] == lb[63]) wd |= 1u << LM_CC_BIT;
        if (j < n && (j == n - 1 || lb[j + 1] != lb[j])) wd |= 1u << LM_TL_BIT;
        tab[j] = wd;
    }
    std::vector<char> has((size_t)m->K, 0);
    for (int j = 0; j < n; j++) has[pos[j].first] = 1;
    m->lm.e0 = -1;
    for (int c = 0; c < m->K; c++)
        if (!has[c]) { m->lm.e0 = c; break; }
    m->lm.z0 = (m->lm.e0 == clampcol) ? std::min(0.0f, m->threshold) : 0.0f;
    m->lm.nq = n > 64 ? 2 : 1;
    m->lm.clampcol = clampcol; m->lm.threshold = m->threshold;
    m->lm.clamp_empty = (clampcol >= 0 && clampcol < m->K && !has[clampcol]) ? 1 : 0;
    unsigned *dv = nullptr;
    int rc = dev_alloc(m, (void **)&dv, tab.size() * 4);
    if (rc) return rc;
    FARNN_HIP_TRY(hipMemcpy(dv, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    m->lm.tab = dv;
    m->lm.on = 1;
    return FARNN_OK;
}

static int alloc_bitmaps(farnn_model *m) {
    m->bmNS = (m->semiring == FARNN_SEMIRING_SUM) ? compact_ns(m->S) : 0;
    if (!m->bmNS) return FARNN_OK;
    const size_t nb = (size_t)m->V * m->S * m->bmNS * sizeof(u64), nw = (size_t)m->S * m->bmNS * sizeof(u64);
    int rc;
    if ((rc = dev_alloc(m, (void **)&m->bmF, nb)) || (rc = dev_alloc(m, (void **)&m->bmB, nb)) ||
        (rc = dev_alloc(m, (void **)&m->bmWF, nw)) || (rc = dev_alloc(m, (void **)&m->bmWB, nw))) return rc;
    FARNN_HIP_TRY(hipMemset(m->bmF, 0, nb)); FARNN_HIP_TRY(hipMemset(m->bmB, 0, nb));
    FARNN_HIP_TRY(hipMemset(m->bmWF, 0, nw)); FARNN_HIP_TRY(hipMemset(m->bmWB, 0, nw));
    return FARNN_OK;
}

static int finish_bitmaps(farnn_model *m, int *bad_dev) {
    int bad = 0;
    FARNN_HIP_TRY(hipGetLastError());
    FARNN_HIP_TRY(hipMemcpy(&bad, bad_dev, sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(bad_dev);
    if (bad) m->bmNS = 0;            // a weight other than 0 / 1: the dense blocks are the only form (the bitmaps stay unused)
    if (m->bmNS >= 1 && m->bmNS <= 2 && (unsigned long long)m->V * m->S * m->bmNS * 8ull < (1ull << 32) - 4096) {     // (= compact_tag_fits' bound)
        // compact_tag_kernel's planes (compact_tag.hip.h).  Its lanes without a state read rows past their block: 4 KiB of slack
        const size_t nb = (size_t)m->V * m->S * m->bmNS * sizeof(u64);
        int rc;
        if ((rc = dev_alloc(m, (void **)&m->bmMF, nb + 4096)) || (rc = dev_alloc(m, (void **)&m->bmMB, nb + 4096)) ||
            (rc = dev_alloc(m, (void **)&m->bmXF, nb + 4096)) || (rc = dev_alloc(m, (void **)&m->bmXB, nb + 4096)) ||
            (rc = dev_alloc(m, (void **)&m->bmTok, (size_t)m->V * sizeof(unsigned)))) return rc;
        FARNN_HIP_TRY(hipMemset(m->bmMF + nb / 8, 0, 4096)); FARNN_HIP_TRY(hipMemset(m->bmMB + nb / 8, 0, 4096));
        FARNN_HIP_TRY(hipMemset(m->bmXF + nb / 8, 0, 4096)); FARNN_HIP_TRY(hipMemset(m->bmXB + nb / 8, 0, 4096));
        std::vector<unsigned> off((size_t)m->V);
        for (int v = 0; v < m->V; v++) off[(size_t)v] = (unsigned)((size_t)v * m->S * m->bmNS * 8);
        FARNN_HIP_TRY(hipMemcpy(m->bmTok, off.data(), off.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        const long long n = (long long)m->V * m->S * m->bmNS;
        merge_planes_kernel<<<(unsigned)((n + 255) / 256), 256>>>(m->bmF, m->bmB, m->bmWF, m->bmWB, m->bmMF, m->bmMB, m->bmXF, m->bmXB,
                                                                  m->bmTok, m->V, m->S, m->bmNS);
        FARNN_HIP_TRY(hipGetLastError());
        FARNN_HIP_TRY(hipDeviceSynchronize());
    }
    return FARNN_OK;
}

static int build_bitmaps_from_dense(farnn_model *m, const float *T, const float *W) {
    int rc = alloc_bitmaps(m);
    if (rc || !m->bmNS) return rc;
    int *bad = nullptr;
    FARNN_HIP_TRY(hipMalloc((void **)&bad, sizeof(int)));
    FARNN_HIP_TRY(hipMemset(bad, 0, sizeof(int)));
    dense_to_bits_kernel<<<dim3(m->V + 1, (m->S * m->S + 255) / 256), 256>>>(T, W, m->bmF, m->bmB, m->bmWF, m->bmWB, m->V, m->S,
                                                                          m->bmNS, bad);
    return finish_bitmaps(m, bad);
}

// ---- create: onehot i-FST --------------------------------------------------------------------
struct DevEdges { const int32_t *word, *from, *to; const float *val; long long n; };     // device copies of an edge list

// compact_edges != nullptr: build ONLY the compact form, from the edge list (no dense blocks; d->T / d->W unused)
static int ifst_create_impl(const farnn_onehot_ifst_desc *d, int device, farnn_model **out, const DevEdges *compact_edges) {
    if (!d || !out) return fail(FARNN_EINVAL, "null argument%s%s");
    *out = nullptr;
    if (d->V <= 0 || d->S <= 0 || d->C <= 0 || ((!d->T || !d->W) && !compact_edges) || !d->O || !d->h0 || !d->hT)
        return fail(FARNN_EINVAL, "onehot_ifst: sizes must be positive and T/W/O/h0/hT non-null%s%s");
    if (d->nl < 0 || d->nl > FARNN_NL_RELUTANH) return fail(FARNN_EINVAL, "onehot_ifst: bad nl%s%s");
    if (d->semiring != FARNN_SEMIRING_SUM && d->semiring != FARNN_SEMIRING_MAX)
        return fail(FARNN_EINVAL, "onehot_ifst: bad semiring%s%s");
    int rc = select_device(device);
    if (rc) return rc;
    farnn_model *m = new (std::nothrow) farnn_model();
    if (!m) return fail(FARNN_ENOMEM, "host allocation failed%s%s");
    TunScope tun_scope(&m->tun);
    m->kind = KIND_IFST; m->device = device;
    m->V = d->V; m->S = d->S; m->C = d->C;
    m->use_crf = d->use_crf ? 1 : 0;
    m->K = d->C + (m->use_crf ? 2 : 0);
    m->Kp = round_up(m->K, 4); m->Kc = round_up(m->K, 64);
    m->nl = d->nl; m->semiring = d->semiring; m->threshold = d->threshold; m->o_idx = d->o_idx;
    pick_chain_geometry(m);
    m->chain_ks = tun(TUN_KS);
    m->SP = m->geom.SP;
    const int od = d->weights_on_device;
    auto bail = [&](int code) { farnn_destroy(m); return code; };
    if (m->K > 64 * SCORE_KCH) return bail(fail(FARNN_ERANGE, "more than 256 label columns%s%s"));
    if (m->geom.NCH > 4) return bail(fail(FARNN_ERANGE, "more than 1024 states%s%s"));

    if (compact_edges) {
        if ((rc = alloc_bitmaps(m))) return bail(rc);
        if (!m->bmNS) return bail(fail(FARNN_ERANGE, "onehot_ifst compact form: needs the sum semiring and at most 512 states%s%s"));
        int *bad = nullptr;
        FARNN_HIP_TRY(hipMalloc((void **)&bad, sizeof(int)));
        FARNN_HIP_TRY(hipMemset(bad, 0, sizeof(int)));
        if (compact_edges->n > 0)
            edges_to_bits_kernel<<<(unsigned)((compact_edges->n + 255) / 256), 256>>>(
                compact_edges->word, compact_edges->from, compact_edges->to, compact_edges->val, compact_edges->n, m->bmF, m->bmB,
                m->bmWF, m->bmWB, m->V, m->S, m->bmNS, bad);
        if ((rc = finish_bitmaps(m, bad))) return bail(rc);
        if (!m->bmNS) return bail(fail(FARNN_EINVAL, "onehot_ifst compact form: an edge is out of range or has a weight other than 1%s%s"));
        m->compact_on = true;
    } else {   // premix T+W once (the reference re-adds it on every call, model_onehot.py:366)
        const size_t nT = (size_t)m->V * m->S * m->S;
        TmpDev T, W;
        if ((rc = T.init(d->T, nT, od))) return bail(rc);
        if ((rc = W.init(d->W, (size_t)m->S * m->S, od))) return bail(rc);
        const size_t nM = (size_t)m->V * m->geom.SR * m->SP;
        if ((rc = dev_alloc(m, (void **)&m->Mf, nM * 4))) return bail(rc);
        if ((rc = dev_alloc(m, (void **)&m->Mb, nM * 4))) return bail(rc);
        if ((rc = launch_premix(T.p, W.p, nullptr, m->Mf, m->Mb, m->V, m->S, m->SP, m->geom.SR))) return bail(rc);
        if ((rc = build_bitmaps_from_dense(m, T.p, W.p))) return bail(rc);
    }
    // o = sum_c O[c,:]  (CE1, model_onehot.py:368); OT = O^T padded, with zero rows for START/STOP
    {
        TmpDev O;
        if ((rc = O.init(d->O, (size_t)m->C * m->S, od))) return bail(rc);
        if ((rc = dev_alloc(m, (void **)&m->o, (size_t)m->SP * 4))) return bail(rc);
        if ((rc = dev_alloc(m, (void **)&m->OT, round_up_sz((size_t)m->S * m->Kc * 4, 1024)))) return bail(rc);
        FARNN_HIP_TRY(hipMemset(m->o, 0, (size_t)m->SP * 4));
        FARNN_HIP_TRY(hipMemset(m->OT, 0, round_up_sz((size_t)m->S * m->Kc * 4, 1024)));
        colsum_kernel<<<(m->S + 255) / 256, 256>>>(O.p, m->o, m->C, m->S);
        int n = m->C * m->S;
        transpose_pad_kernel<<<(n + 255) / 256, 256>>>(O.p, m->OT, m->C, m->S, m->Kc);
        FARNN_HIP_TRY(hipGetLastError());
        if ((rc = build_ot_image(m))) return bail(rc);
        FARNN_HIP_TRY(hipDeviceSynchronize());
        if ((rc = build_label_map(m))) return bail(rc);
    }
    if ((rc = dev_upload(m, &m->h0, d->h0, m->S, m->SP, od))) return bail(rc);
    if ((rc = dev_upload(m, &m->hT, d->hT, m->S, m->SP, od))) return bail(rc);
    // the priority matrix of the onehot models is [C][C]; with CRF the two extra tags pass through
    if (d->P && m->use_crf) {
        std::vector<float> Pc((size_t)m->C * m->C), Pk((size_t)m->K * m->K, 0.0f);
        FARNN_HIP_TRY(hipMemcpy(Pc.data(), d->P, Pc.size() * 4,
                                od ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
        for (int i = 0; i < m->C; i++)
            for (int j = 0; j < m->C; j++) Pk[(size_t)i * m->K + j] = Pc[(size_t)i * m->C + j];
        Pk[(size_t)(m->K - 2) * m->K + m->K - 2] = 1.0f;
        Pk[(size_t)(m->K - 1) * m->K + m->K - 1] = 1.0f;
        if ((rc = setup_priority(m, Pk.data(), 0))) return bail(rc);
    } else if ((rc = setup_priority(m, d->P, od))) return bail(rc);
    if ((rc = setup_crf(m, d->crf_trans, od))) return bail(rc);
    *out = m;
    return FARNN_OK;
}

extern "C" int farnn_onehot_ifst_create(const farnn_onehot_ifst_desc *d, int device, farnn_model **out) {
    return ifst_create_impl(d, device, out, nullptr);
}

extern "C" int farnn_has_compact(const farnn_model *m) { return (m && m->bmNS > 0 && m->bmF) ? 1 : 0; }

extern "C" int farnn_set_compact(farnn_model *m, int32_t enable) {
    if (!m) return fail(FARNN_EINVAL, "set_compact: null model%s%s");
    if (enable && !farnn_has_compact(m))
        return fail(FARNN_EINVAL, "set_compact: this model has no compact form (i-FST with 0/1 weights, sum semiring, at most 512 states)%s%s");
    if (!enable && !m->Mf) return fail(FARNN_EINVAL, "set_compact: this handle was created compact-only (no dense blocks)%s%s");
    m->compact_on = enable != 0;
    return FARNN_OK;
}

// ---- workspace -------------------------------------------------------------------------------
extern "C" int farnn_reserve(farnn_model *m, int32_t B, int32_t L) {
    if (!m || B <= 0 || L <= 0) return fail(FARNN_EINVAL, "reserve: bad arguments%s%s");
    if (B <= m->wsB && L <= m->wsL) return FARNN_OK;
    TunScope tun_scope(&m->tun);
    FARNN_HIP_TRY(hipSetDevice(m->device));
    int nB = B > m->wsB ? B : m->wsB, nL = L > m->wsL ? L : m->wsL;
    if (m->A) {
        FARNN_HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(m->A); (void)hipFree(m->Bk); (void)hipFree(m->offs); (void)hipFree(m->order);
        if (m->hs) (void)hipFree(m->hs);
        if (m->crf_scores) (void)hipFree(m->crf_scores);
        if (m->d1_br) (void)hipFree(m->d1_br);
    }
    m->d1_br = nullptr;
    m->A = m->Bk = m->crf_scores = nullptr; m->offs = nullptr; m->order = nullptr; m->wsB = m->wsL = 0;
    m->hs = nullptr;
    size_t stash = (size_t)nB * (nL + 1) * m->SP * sizeof(float);
    FARNN_HIP_TRY(hipMalloc((void **)&m->A, stash));
    FARNN_HIP_TRY(hipMalloc((void **)&m->Bk, stash));
    FARNN_HIP_TRY(hipMalloc((void **)&m->offs, (size_t)(nB + 1) * sizeof(int64_t)));
    FARNN_HIP_TRY(hipMalloc((void **)&m->order, (size_t)nB * sizeof(int)));
    m->hs_bytes = round_up_sz((size_t)(3 * nB + 48) * sizeof(unsigned long long), 16);     // progress [2][nB], arrival [nB], the launch counter
    FARNN_HIP_TRY(hipMalloc((void **)&m->hs, m->hs_bytes));
    FARNN_HIP_TRY(hipMemset(m->hs, 0, m->hs_bytes));
    if (m->use_crf)
        FARNN_HIP_TRY(hipMalloc((void **)&m->crf_scores, (size_t)nB * nL * m->Kp * sizeof(float) + 1024));   // +1 KiB: LDS-DMA pieces
    if (m->d1_BSSp)
        FARNN_HIP_TRY(hipMalloc((void **)&m->d1_br, (size_t)nB * nL * ((m->S + 15) / 16) * ((m->RO + 15) / 16 * 16) * sizeof(float)));
    FARNN_HIP_TRY(hipMemset(m->A, 0, stash));
    FARNN_HIP_TRY(hipMemset(m->Bk, 0, stash));
    FARNN_HIP_TRY(hipDeviceSynchronize());
    m->wsB = nB; m->wsL = nL;
    return FARNN_OK;
}

// ---- profiling -------------------------------------------------------------------------------
struct KernelTimer {
    farnn_model *m; int which; hipStream_t s; hipEvent_t e0 = nullptr, e1 = nullptr;
    // ext = true: the events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL start/stop events): no
    // extra packets on the stream, so a timed step costs the same as an untimed one; the caller passes e0/e1 to the launch
    bool ext;
    KernelTimer(farnn_model *m_, int w, hipStream_t s_, bool ext_ = false) : m(m_), which(w), s(s_), ext(ext_) {
        if (m->prof_this_call) {
            e0 = m->prof.get(); e1 = m->prof.get();
            if (e0 && e1 && !ext) (void)hipEventRecord(e0, s);
        }
    }
    ~KernelTimer() {
        if (e0 && e1) {
            if (!ext) (void)hipEventRecord(e1, s);
            m->prof.ev[which].push_back(e0);
            m->prof.ev[which].push_back(e1);
        }
    }
};

static void prof_fold(farnn_model *m) {
    for (int k = 0; k < KERN_COUNT; k++) {
        auto &v = m->prof.ev[k];
        for (size_t i = 0; i + 1 < v.size(); i += 2) {
            float ms = 0.f;
            if (hipEventSynchronize(v[i + 1]) == hipSuccess &&
                hipEventElapsedTime(&ms, v[i], v[i + 1]) == hipSuccess) {
                m->prof.ms[k] += ms; m->prof.n[k] += 1;
            }
            m->prof.pool.push_back(v[i]); m->prof.pool.push_back(v[i + 1]);
        }
        v.clear();
    }
}

extern "C" int farnn_set_profiling(farnn_model *m, int32_t enable) {
    if (!m) return fail(FARNN_EINVAL, "null model%s%s");
    prof_fold(m);
    if (enable) {
        for (int k = 0; k < KERN_COUNT; k++) { m->prof.ms[k] = 0; m->prof.n[k] = 0; }
        while (m->prof.pool.size() < 256) {       // events exist before the timed region starts
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) break;
            m->prof.pool.push_back(e);
        }
    }
    m->profiling = enable > 0 ? enable : 0;
    m->calls = 0;
    return FARNN_OK;
}

extern "C" int farnn_kernel_time(farnn_model *m, int32_t which, double *total_ms, int64_t *launches) {
    if (!m || which < 0 || which >= KERN_COUNT) return fail(FARNN_EINVAL, "kernel_time: bad arguments%s%s");
    prof_fold(m);
    if (total_ms) *total_ms = m->prof.ms[which];
    if (launches) *launches = m->prof.n[which];
    return FARNN_OK;
}

extern "C" const char *farnn_kernel_name(const farnn_model *m, int32_t which) {
    if (!m) return "";
    TunScope tun_scope(&m->tun);
    switch (which) {
        case KERN_CHAIN:
            if (m->compact_on) return m->last_fused ? "compact_tag_kernel<fused: both chains + label-map scores + decode>" : "compact_chain_kernel";
            if (m->last_regs && m->last_fused && m->use_crf) return "chain_viterbi_kernel<fused: recurrence + scores + CRF decode>";
            if (m->last_regs && m->rgeom.wide) return m->last_fused ? "chain_wide_kernel<fused: scores + decode beside the recurrence>" : "chain_wide_kernel";
            if (m->last_regs) return m->last_fused ? "chain_regs_kernel<fused: scores + decode beside the recurrence>" : "chain_regs_kernel";
            if (m->dense_decomp) return "chain_kernel";
            if (m->kind == KIND_DECOMP && m->last_fused && m->last_wave) return "decomp_regs_kernel<fused: scores + decode beside the recurrence>";
            if (m->kind == KIND_DECOMP || m->kind == KIND_DECOMP1 || m->kind == KIND_DECOMP0)
                return m->rows.ok ? ((m->last_wave || (m->calls == 0 && m->dw.farnn == 0 && m->dw.R <= DG_ROWS && !tun(TUN_DECOMP_NOREGS))) ? "decomp_regs_kernel" : "decomp_rows_kernel") : "decomp_chain_kernel";
            return "chain_kernel";
        case KERN_SCORE: return m->kind == KIND_FST4 ? "fst4_score_kernel"
                              : (m->kind == KIND_IND1 ? "ind1_score_kernel"
                              : (m->kind == KIND_DECOMP1 ? (m->d1_BSSp ? "decomp1_br_mfma_kernel+decomp1_label_kernel" : "decomp1_score_kernel")
                              : (m->kind == KIND_DECOMP0 ? "decomp0_score_kernel"
                              : (m->use_crf ? "score_tile_kernel+viterbi_kernel" : (m->last_lm_score ? "label_map_score_kernel" : "score_tile_kernel")))));
        case KERN_PREP:  return "batch_prep_kernel";
        default: return "";
    }
}

// ---- the hot path ----------------------------------------------------------------------------

// the register-fed recurrence's view of a call (chain_regs_params.hip.h); the hand-off words are filled in by the callers that use them
static RegsParams make_regs_params(farnn_model *m, const int64_t *x, const int64_t *len, int B, int full) {
    const RegsGeom &rg = m->rgeom;
    RegsParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.Mf = m->Mf; rp.Mb = m->Mb; rp.blk = (long long)m->geom.SR * m->SP;
    rp.o = m->o; rp.h0 = m->h0; rp.hT = m->hT; rp.x = x; rp.len = len;
    rp.order = m->order_valid ? m->order : nullptr; rp.sort = m->sort_in_kernel ? 1 : 0;
    rp.A = m->A; rp.Bk = m->Bk; rp.B = B; rp.L = m->curL; rp.S = m->S; rp.SP = m->SP; rp.CPR = rg.CPR; rp.V = m->V;
    rp.G = rg.G; rp.RPG = rg.RPG; rp.RQ = rg.RQ; rp.D = rg.D; rp.PS = rg.PS; rp.pair = rg.wide ? 0 : 1;
    rp.nl = m->nl; rp.full = full; rp.dbg = tun(TUN_DBG);
    rp.dest = (!rg.wide && m->semiring != FARNN_SEMIRING_MAX && !tun(TUN_NODEST)) ? 1 : 0;     // chain_dest.hip.h
    return rp;
}

// The hand-off words of the one-launch forms (chain_regs.hip.h / decomp_regs.hip.h + beside.hip.h) and their launch counter.
// A launch's epoch is (the counter >> BS_EPOCH_SHIFT) + 1, read from device memory by the kernel; every launch adds exactly
// BS_EPOCH_SPAN to the counter whatever its batch size (beside.hip.h, bs_launch_epoch).  Nothing per launch comes from the host and
// nothing is ever reset: the step replays from a HIP graph as the very same launch, and graphs captured at different batch sizes
// and eager calls may interleave on a handle (stream-ordered).  The words are zeroed once, when the workspace is allocated.
static int handoff_words(farnn_model *m, unsigned long long **prog, unsigned long long **arr, unsigned long long **done) {
    *prog = m->hs; *arr = m->hs + (size_t)2 * m->wsB; *done = m->hs + (size_t)3 * m->wsB + 16;      // (the counter on a 128-byte line of its own)
    return FARNN_OK;
}

// fuse_sp != nullptr: ask for the fused launch (scores + argmax decode as the chain kernel's epilogue); *fused tells
// whether the geometry allowed it (else the caller launches the score kernel itself)
static int launch_chain(farnn_model *m, const int64_t *x, const int64_t *len, int B, int L, int full,
                        hipStream_t s, const ScoreParams *fuse_sp = nullptr, bool *fused = nullptr) {
    const ChainGeom &g = m->geom;
    if (fused) *fused = false;
    m->last_regs = false;
    // ---- the register-fed kernel (chain_regs.hip.h) where its geometry applies: S <= 72, two workgroups per compute unit, or
    // its wide form (chain_wide.hip.h): 72 < S <= 128, one workgroup per compute unit.
    // With fuse_sp (threshold/argmax decode, K <= 256) the scores and the decode run beside the recurrence: ONE launch.
    if (m->rgeom.ok && !tun(TUN_NOREGS)) {
        const RegsGeom &rg = m->rgeom;
        const size_t lds_cap = rg.wide ? 158 * 1024 : 80 * 1024;
        bool score = fuse_sp && m->hs && m->OTm && m->c16 >= 1 && m->c16 <= (rg.wide ? RGW_NG : RG_NG) && m->Kc <= 256 && m->curL <= 31 * RG_TT &&
                     (B <= 1024 || !fuse_sp->flat || fuse_sp->offs) && !tun(TUN_NOFUSE);
        // Which form is the faster one was measured, and the faster one is the default.  S <= 72 with a label-map output matrix (tags
        // only): the recurrence-only kernel followed by the label-map score launch (K2l).  Round 5 kept ONE launch (the scores and
        // the decode beside the recurrence: north_star's form) while 2 B <= compute units, on measurements at B = 64 / 256 / 1 024 and
        // L = 64 only; round 6's grid (scripts/gpu_r06_dispatch_grid.py -> profiles/r06_dispatch_grid.txt: B = 16..128 x L = 16..100)
        // has two launches ahead at 32 of its 35 points -- by 5-20 % at the reference's default --seq_max_len 30 and below, within
        // 1 % either way at L = 64 -- so the rule no longer looks at the batch.  FARNN_FUSE=1 keeps the one launch (what a HIP graph
        // replays as one node); tests/test_gpu_dispatch_ab.py holds the default to "not the slower form" at nine (B, L) points.
        if (score && !rg.wide && bs_label_map_path(*fuse_sp) && !tun(TUN_FUSE)) score = false;
        const bool lm_path = score && bs_label_map_path(*fuse_sp);
        const bool dest = !rg.wide && m->semiring != FARNN_SEMIRING_MAX && !tun(TUN_NODEST);
        size_t lds = (size_t)regs_lds(m->curL, m->SP, rg.NP, score ? m->c16 : 0, score ? m->Kc : 0, score, rg.RQ, lm_path, dest).total * sizeof(float);
        // the wide form PAIRED (two workgroups per compute unit, like S <= 72): a ring of two steps and the label-map path's LDS
        const bool paired = rg.wide && lm_path && rg.RQ <= 9 && lds <= 80 * 1024 && !tun(TUN_WIDE_UNPAIRED);
        if (score && lds > lds_cap) {               // the score tiles do not fit (beside a second workgroup): recurrence only
            score = false;
            lds = (size_t)regs_lds(m->curL, m->SP, rg.NP, 0, 0, false, rg.RQ, false, dest).total * sizeof(float);
        }
        if (lds <= lds_cap) {
            RegsParams rp = make_regs_params(m, x, len, B, full);
            if (score) {
                int hrc = handoff_words(m, &rp.prog, &rp.arr, &rp.done);
                if (hrc) return hrc;
                if (tun(TUN_HOST_EPOCH)) {       // diagnostic A/B: the epoch as a kernel argument (not graph-capturable)
                    if (++m->epoch_u == 0) { FARNN_HIP_TRY(hipMemsetAsync(m->hs, 0, m->hs_bytes, s)); m->epoch_u = 1; }
                    rp.done = nullptr; rp.epoch_host = m->epoch_u + 0x40000000u;
                }
                rp.spin = tun(TUN_FUSE_SPIN);
                rp.solo_margin = tun(TUN_SOLO_MARGIN);
                rp.sp = *fuse_sp;
                if (fused) *fused = true;
            }
            KernelTimer kt(m, KERN_CHAIN, s, /*ext=*/true);
            if (paired && score) { rp.D = 2; rp.pair = 1; }
            const int rc = (paired && score) ? launch_chain_wide_paired(rp, m->semiring == FARNN_SEMIRING_MAX, s, kt.e0, kt.e1)
                           : rg.wide ? launch_chain_wide(rp, m->semiring == FARNN_SEMIRING_MAX, score, s, kt.e0, kt.e1)
                                     : launch_chain_regs(rp, m->semiring == FARNN_SEMIRING_MAX, score, s, kt.e0, kt.e1);
            if (rc) return rc;
            m->last_regs = true;
            return FARNN_OK;
        }
    }
    (void)fuse_sp;
    ChainParams p;
    p.Mf = m->Mf; p.Mb = m->Mb; p.blk = (long long)m->geom.SR * m->SP;
    p.o = m->o; p.h0 = m->h0; p.hT = m->hT; p.x = x; p.len = len; p.A = m->A; p.Bk = m->Bk;
    p.order = m->order_valid ? m->order : nullptr;
    p.sort = m->sort_in_kernel ? 1 : 0;
    p.B = B; p.L = m->curL; p.S = m->S; p.SP = m->SP; p.CPR = g.CPR; p.V = m->V;
    p.NW = g.NW; p.NLD = g.NLD; p.G = g.G; p.LPR = g.LPR; p.RPG = g.RPG; p.RPGp = g.RPGp; p.NQ = g.NQ;
    p.nl = m->nl; p.full = full; p.dbg = tun(TUN_DBG);
    // ring shape: a whole step per phase when it fits, KS phases deep
    int ks = 2, nqp = g.NQ;
    if (!g.pick_ring(m->curL, m->chain_ks, ks, nqp))
        return fail(FARNN_ERANGE, "chain kernel: LDS ring does not fit (sequence too long for this S)%s%s");
    p.KS = ks; p.NQP = nqp; p.PPS = (g.NQ + nqp - 1) / nqp;
    const size_t lds = g.lds_bytes(m->curL, ks, nqp);
    dim3 grid(2 * B), block((g.NW + g.NLD + 1) * 64);         // compute + loader + writer wavefronts
    const bool mx = m->semiring == FARNN_SEMIRING_MAX;
    int rc = FARNN_OK;
    if (tun(TUN_CHAIN_HELPER) && block.x < 512) block = dim3(block.x + 64);     // experiment: an idle eighth wavefront
#define FARNN_LAUNCH_CHAIN(NCH, MX, FQ)                                                       \
    do {                                                                                      \
        if ((rc = raise_lds_limit(chain_kernel<NCH, MX, FQ>, lds))) return rc;                \
        if (kt.e0 && kt.e1)                                                                   \
            hipExtLaunchKernelGGL((chain_kernel<NCH, MX, FQ>), grid, block, (uint32_t)lds, s, kt.e0, kt.e1, 0, p); \
        else                                                                                  \
            chain_kernel<NCH, MX, FQ><<<grid, block, lds, s>>>(p);                            \
    } while (0)
#define FARNN_LAUNCH_CHAIN_MX(NCH, FQ)                                                        \
    do { if (mx) FARNN_LAUNCH_CHAIN(NCH, true, FQ); else FARNN_LAUNCH_CHAIN(NCH, false, FQ); } while (0)
    KernelTimer kt(m, KERN_CHAIN, s, /*ext=*/true);
    const int fq_max = block.x <= 384 ? 6 : 3;
    const int fq = (g.NCH == 1 && p.PPS == 1 && g.NQ <= fq_max && !tun(TUN_NOFAST)) ? g.NQ : 0;
    if (g.NCH == 1) {
        if (fq == 1) FARNN_LAUNCH_CHAIN_MX(1, 1);
        else if (fq == 2) FARNN_LAUNCH_CHAIN_MX(1, 2);
        else if (fq == 3) FARNN_LAUNCH_CHAIN_MX(1, 3);
        else if (fq == 4) FARNN_LAUNCH_CHAIN_MX(1, 4);
        else if (fq == 5) FARNN_LAUNCH_CHAIN_MX(1, 5);
        else if (fq == 6) FARNN_LAUNCH_CHAIN_MX(1, 6);
        else FARNN_LAUNCH_CHAIN_MX(1, 0);
    } else if (g.NCH == 2) FARNN_LAUNCH_CHAIN_MX(2, 0);
    else if (g.NCH <= 4) FARNN_LAUNCH_CHAIN_MX(4, 0);
    else return fail(FARNN_ERANGE, "unsupported state count%s%s");
#undef FARNN_LAUNCH_CHAIN_MX
#undef FARNN_LAUNCH_CHAIN
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

// ---- decomposed modes whose step matrix is materialised anyway: dense per-word blocks + the chain kernel ----
static int build_dense_blocks(farnn_model *m) {
    const DecompWeights &w = m->dw;
    if (w.farnn != 0 || !(w.semiring == FARNN_SEMIRING_MAX || w.mask)) return FARNN_OK;
    pick_chain_geometry(m);
    m->chain_ks = tun(TUN_KS);
    if (m->geom.NCH > 4 || m->geom.SP != m->SP) return FARNN_OK;
    const size_t nM = (size_t)m->V * m->geom.SR * m->SP;
    if (nM * 8 > (size_t)64 << 30) return FARNN_OK;               // keep it under 64 GB; else the generic kernel
    int rc;
    if ((rc = dev_alloc(m, (void **)&m->Mf, nM * 4))) return rc;
    if ((rc = dev_alloc(m, (void **)&m->Mb, nM * 4))) return rc;
    FARNN_HIP_TRY(hipMemset(m->Mb, 0, nM * 4));
    dim3 grid((m->geom.SR * m->SP + 255) / 256, m->V);
    materialise_blocks_kernel<<<grid, 256>>>(w.Vgen, w.S1, w.S2, w.W, w.mask, m->Mf, m->Mb, m->S, m->SP, m->geom.SR,
                                             m->R, m->Rp);
    FARNN_HIP_TRY(hipGetLastError());
    FARNN_HIP_TRY(hipDeviceSynchronize());
    m->dense_decomp = true;
    return FARNN_OK;
}

// ---- packed rows + gate tables for decomp_rows_kernel (create time) ------------------------------
static int build_rows_pack(farnn_model *m) {
    DecompWeights &w = m->dw;
    DecompRowsPack &k = m->rows;
    k.ok = false;
    if (w.semiring != FARNN_SEMIRING_SUM || w.mask) return FARNN_OK;
    const int tvl = m->Rp + (w.farnn >= 1 ? m->SP : 0) + (w.farnn == 2 ? m->SP : 0);
    if (tvl > DR_MAX_PF * DR_THREADS) return FARNN_OK;
    k.nch2 = (m->S + DR_CHUNK - 1) / DR_CHUNK; k.nch3 = (m->Rp + m->S + DR_CHUNK - 1) / DR_CHUNK;
    k.ld2 = rows_ld(m->S); k.ld3 = rows_ld(m->Rp + m->S);
    k.n1 = w.farnn == 2 ? 2 * m->S : 0;
    k.n2 = m->R + (w.farnn == 1 ? m->S : 0);
    k.n3 = m->S;
    PackSrc q;
    q.S1 = w.S1; q.S2 = w.S2; q.W = w.W; q.Wss1 = w.Wss1; q.Wss2 = w.Wss2; q.o = w.o;
    q.S = m->S; q.SP = m->SP; q.R = m->R; q.Rp = m->Rp; q.farnn = w.farnn;
    int rc;
    auto blocks = [](long long n) { return (unsigned)((n + 255) / 256); };
    for (int dir = 0; dir < 2; dir++) {
        if ((rc = dev_alloc(m, (void **)&k.P2[dir], (size_t)k.n2 * k.ld2 * 4))) return rc;
        if ((rc = dev_alloc(m, (void **)&k.P3[dir], (size_t)k.n3 * k.ld3 * 4))) return rc;
        pack_p2_kernel<<<blocks((long long)k.n2 * k.ld2), 256>>>(q, k.P2[dir], k.n2, k.ld2, dir);
        pack_p3_kernel<<<blocks((long long)k.n3 * k.ld3), 256>>>(q, k.P3[dir], k.ld3, dir);
    }
    if (k.n1) {
        if ((rc = dev_alloc(m, (void **)&k.P1, (size_t)k.n1 * k.ld2 * 4))) return rc;
        pack_p1_kernel<<<blocks((long long)k.n1 * k.ld2), 256>>>(q, k.P1, k.ld2);
    }
    // the per-word rows a step reads, side by side: [Vgen row | update-gate row | reset-gate row] (one base, one load per prefetch slot)
    if (w.farnn == 0) {
        k.TVt = w.Vgen;
    } else {
        float *T = nullptr;
        if ((rc = dev_alloc(m, (void **)&T, (size_t)m->V * tvl * 4))) return rc;
        word_rows_kernel<<<blocks((long long)m->V * m->Rp), 256>>>(w.Vgen, T, tvl, m->V, m->Rp);
        gate_table_kernel<<<blocks((long long)m->V * m->SP), 256>>>(w.Vgen, w.Wrs1, w.bs1, T, tvl, m->Rp, m->V, m->R, m->Rp, m->S, m->SP);
        if (w.farnn == 2)
            gate_table_kernel<<<blocks((long long)m->V * m->SP), 256>>>(w.Vgen, w.Wrs2, w.bs2, T, tvl, m->Rp + m->SP, m->V, m->R, m->Rp, m->S, m->SP);
        k.TVt = T;
    }
    FARNN_HIP_TRY(hipGetLastError());
    FARNN_HIP_TRY(hipDeviceSynchronize());
    k.ok = true;
    return FARNN_OK;
}

// the recurrence of the three decomposed kinds: rows kernel when packed, else the older kernels
// fuse_sp != nullptr: ask for the one-launch form (scores + argmax decode beside the recurrence); *fused tells whether the
// model's kernel and geometry allowed it (else the caller launches the score kernel itself)
static int launch_decomp_recurrence(farnn_model *m, const int64_t *x, const int64_t *lengths, int B, int full,
                                    hipStream_t s, const ScoreParams *fuse_sp = nullptr, bool *fused = nullptr) {
    if (fused) *fused = false;
    const int *order = m->order_valid ? m->order : nullptr;
    if (m->n_cu <= 0) {
        int dev = 0, ncu = 0;
        FARNN_HIP_TRY(hipGetDevice(&dev));
        FARNN_HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        m->n_cu = ncu > 0 ? ncu : 256;
    }
    RegsPlan rp;
    m->last_wave = m->rows.ok && regs_plan(m->rows, m->dw, m->curL, rp);
    if (m->last_wave) {      // farnn = 0, rank <= 64: four wavefronts per chain, the packed rows in registers
        BesideParams bs;
        const BesideParams *use = nullptr;
        bool score = fuse_sp && m->hs && m->OTm && m->c16 >= 1 && m->c16 <= DG_NG && m->Kc <= 256 && m->curL <= 31 * RG_TT &&
                     (B <= 1024 || !fuse_sp->flat || fuse_sp->offs) && !tun(TUN_NOFUSE);
        if (score) {
            rp.lds_score = regs_score_lds(rp, m->curL, m->SP, m->c16, m->Kc);
            if (rp.lds_score > 80 * 1024) score = false;
        }
        if (score) {
            memset(&bs, 0, sizeof(bs));
            bs.A = m->A; bs.Bk = m->Bk; bs.B = B; bs.L = m->curL; bs.SP = m->SP; bs.CPR = m->SP / 4;
            int hrc = handoff_words(m, &bs.prog, &bs.arr, &bs.done);
            if (hrc) return hrc;
            bs.spin = tun(TUN_FUSE_SPIN); bs.dbg = tun(TUN_DBG); bs.sp = *fuse_sp;
            use = &bs;
            if (fused) *fused = true;
        }
        return launch_decomp_regs(m->rows, m->dw, rp, x, lengths, order, m->sort_in_kernel ? 1 : 0, m->A, m->Bk, B, m->curL,
                                  full, s, use);
    }
    RowsPlan pl;
    if (m->rows.ok && rows_plan(m->rows, m->dw, B, m->curL, pl))
        return launch_decomp_rows(m->rows, m->dw, pl, x, lengths, order, m->sort_in_kernel ? 1 : 0, m->A, m->Bk, B,
                                  m->curL, full, m->n_cu, s);
    return launch_decomp_chain(m->dw, x, lengths, order, m->A, m->Bk, B, m->curL, full, s);
}

// fused: the Viterbi kernel computes the scores itself (no score_tile launch went before it)
static bool viterbi_can_fuse(const farnn_model *m, const ScoreParams &p) {
    return m->use_crf && !p.scores && !p.P && p.A && p.OT && m->K <= 256 &&
           viterbi_hist_lds_bytes(m->K, m->Kp, p.SP, p.L, true) <= 158 * 1024 && viterbi_hist_ib4(m->K) <= 6 &&
           !tun(TUN_VITERBI_BP) && !tun(TUN_VITERBI_UNFUSED);
}

static int launch_viterbi(farnn_model *m, const ScoreParams &p, int B, hipStream_t s, bool fused = false) {
    int rc;
    if (m->K > 256) return fail(FARNN_ERANGE, "Viterbi: more than 256 tags%s%s");
    const size_t hlds = viterbi_hist_lds_bytes(m->K, m->Kp, p.SP, p.L, fused);
    // (K >= 224 never takes this form: the transposed transition table alone is 196 KiB -- no IB4 = 7, 8 instantiations)
    if (hlds <= 158 * 1024 && viterbi_hist_ib4(m->K) <= 6 && !tun(TUN_VITERBI_BP)) {
        // partition history in LDS, back-pointers recomputed along the path
        const int threads = viterbi_hist_score_threads(m->K, fused);
#define FARNN_LAUNCH_VITH(N)                                                                  \
    case N:                                                                                   \
        if (fused) {                                                                          \
            if ((rc = raise_lds_limit(viterbi_hist_kernel<N, true>, hlds))) return rc;        \
            viterbi_hist_kernel<N, true><<<dim3(B), dim3(threads), hlds, s>>>(p);             \
        } else {                                                                              \
            if ((rc = raise_lds_limit(viterbi_hist_kernel<N, false>, hlds))) return rc;       \
            viterbi_hist_kernel<N, false><<<dim3(B), dim3(threads), hlds, s>>>(p);            \
        }                                                                                     \
        break;
        switch (viterbi_hist_ib4(m->K)) {
            FARNN_LAUNCH_VITH(0) FARNN_LAUNCH_VITH(1) FARNN_LAUNCH_VITH(2) FARNN_LAUNCH_VITH(3) FARNN_LAUNCH_VITH(4)
            FARNN_LAUNCH_VITH(5) FARNN_LAUNCH_VITH(6)
        }
#undef FARNN_LAUNCH_VITH
        FARNN_HIP_TRY(hipGetLastError());
        return FARNN_OK;
    }
    if (fused) return fail(FARNN_EINVAL, "Viterbi: the fused form needs the history in LDS%s%s");
    // long sequences: two partition rows + stored back-pointers
    const size_t vlds = viterbi_lds_bytes(m->K, m->Kp, p.L);
    const int threads = round_up(4 * m->K, 64);
#define FARNN_LAUNCH_VIT(N)                                                                   \
    do {                                                                                      \
        if ((rc = raise_lds_limit(viterbi_kernel<N>, vlds))) return rc;                       \
        viterbi_kernel<N><<<dim3(B), dim3(threads), vlds, s>>>(p);                            \
    } while (0)
    const int ib4 = viterbi_ib4(m->K);
    if (ib4 == 2) FARNN_LAUNCH_VIT(2);            // K <= 32
    else if (ib4 == 4) FARNN_LAUNCH_VIT(4);       // K <= 64
    else if (ib4 == 9) FARNN_LAUNCH_VIT(9);       // K <= 144
    else if (ib4 == 13) FARNN_LAUNCH_VIT(13);     // K <= 208
    else FARNN_LAUNCH_VIT(16);                    // K <= 256
#undef FARNN_LAUNCH_VIT
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

static int launch_decomp1_score(farnn_model *m, const int64_t *x, const int64_t *len, int B, int full,
                                int32_t *tags, int64_t *flat, float *scores, hipStream_t s) {
    Decomp1ScoreParams p;
    p.A = m->A; p.Bk = m->Bk; p.Vgen = m->dw.Vgen; p.S1 = m->dw.S1; p.S2 = m->dw.S2; p.W = m->dw.W;
    p.S1o = m->d1_S1o; p.S2o = m->d1_S2o; p.CoutT = m->d1_CoutT; p.P = m->P;
    p.x = x; p.len = len; p.offs = flat ? m->offs : nullptr;
    p.tags = tags; p.flat = flat; p.scores = scores; p.crf_scores = m->crf_scores;
    p.B = B; p.L = m->curL; p.S = m->S; p.SP = m->SP; p.R = m->R; p.Rp = m->Rp; p.RO = m->RO; p.ROp = m->ROp;
    p.V = m->V;
    p.K = m->K; p.Kp = m->Kp; p.Kc = m->Kc;
    p.full = full; p.use_crf = m->use_crf; p.o_idx = m->o_idx; p.threshold = m->threshold;
    int rc;
    KernelTimer kt(m, KERN_SCORE, s);
    Decomp1MfmaParams qm;
    qm.base = p; qm.BSSp = m->d1_BSSp; qm.S1oP = m->d1_S1oP; qm.S2oP = m->d1_S2oP; qm.br = m->d1_br;
    if (!full) qm.base.offs = m->offs;       // farnn_tag computes the flat offsets for this path even without flat output
    qm.MT = (m->S + 15) / 16; qm.NT = (m->RO + 15) / 16; qm.KQ4 = (m->S + 15) / 16;
    const size_t mlds = decomp1_mfma_lds_bytes(qm.MT, qm.NT);
    if (m->d1_BSSp && m->d1_br && mlds <= 80 * 1024) {
        // persistent: two workgroups per CU, every wavefront owns a contiguous slice of the live tokens
        if (m->n_cu <= 0) {
            int dev = 0, ncu = 0;
            FARNN_HIP_TRY(hipGetDevice(&dev));
            FARNN_HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
            m->n_cu = ncu > 0 ? ncu : 256;
        }
        const int nwg = std::min(2 * m->n_cu, (B * p.L + 3) / 4);
#define FARNN_LAUNCH_D1M(N)                                                                    \
        case N:                                                                               \
            if ((rc = raise_lds_limit(decomp1_br_mfma_kernel<N>, mlds))) return rc;           \
            decomp1_br_mfma_kernel<N><<<dim3(nwg), dim3(256), mlds, s>>>(qm);                 \
            break;
        switch (qm.NT) {
            FARNN_LAUNCH_D1M(1) FARNN_LAUNCH_D1M(2) FARNN_LAUNCH_D1M(3) FARNN_LAUNCH_D1M(4) FARNN_LAUNCH_D1M(5)
            default: return fail(FARNN_ERANGE, "decomp_ind1: output rank above 80 on the MFMA path%s%s");
        }
#undef FARNN_LAUNCH_D1M
        FARNN_HIP_TRY(hipGetLastError());
        const int NC = qm.NT * 16;
        // one 16-wavefront workgroup per CU when the weights fit in LDS beside the per-wavefront rows
        const bool staged = decomp1_label_lds_bytes(NC, m->RO, m->K, m->Kc, true, 16) <= 150 * 1024;
        const int lthreads = staged ? 1024 : 256;
        const size_t llds = decomp1_label_lds_bytes(NC, m->RO, m->K, m->Kc, staged, lthreads / 64);
        const int lgrid = std::min((staged ? 1 : 4) * m->n_cu, (B * p.L + lthreads / 64 - 1) / (lthreads / 64));
        if (staged) {
            if ((rc = raise_lds_limit(decomp1_label_kernel<true>, llds))) return rc;
            decomp1_label_kernel<true><<<dim3(lgrid), dim3(lthreads), llds, s>>>(p, m->d1_br, NC, qm.MT, full ? nullptr : m->offs);
        } else {
            decomp1_label_kernel<false><<<dim3(lgrid), dim3(lthreads), llds, s>>>(p, m->d1_br, NC, qm.MT, full ? nullptr : m->offs);
        }
    } else {
        const size_t lds = decomp1_score_lds_bytes(m->S, m->SP, m->Rp, m->ROp, m->Kc);
        if ((rc = raise_lds_limit(decomp1_score_kernel, lds))) return rc;
        decomp1_score_kernel<<<dim3(p.L, B), dim3(256), lds, s>>>(p);
    }
    FARNN_HIP_TRY(hipGetLastError());
    if (m->use_crf) {
        ScoreParams v;
        memset(&v, 0, sizeof(v));
        v.trT = m->tr; v.len = len; v.offs = flat ? m->offs : nullptr; v.tags = tags; v.flat = flat;
        v.crf_scores = m->crf_scores; v.B = B; v.L = m->curL; v.K = m->K; v.Kp = m->Kp;
        v.full = full; v.use_crf = 1; v.o_idx = m->o_idx; v.threshold = m->threshold;
        if ((rc = launch_viterbi(m, v, B, s))) return rc;
    }
    return FARNN_OK;
}

static int launch_decomp0_score(farnn_model *m, const int64_t *x, const int64_t *len, int B, int full,
                                int32_t *tags, int64_t *flat, float *scores, hipStream_t s) {
    Decomp0ScoreParams p;
    p.A = m->A; p.Bk = m->Bk; p.Vgen = m->d0_Vgen; p.S1 = m->dw.S1; p.S2 = m->dw.S2; p.CT = m->d0_CT;
    p.S1w = m->d0_S1w; p.S2w = m->d0_S2w; p.CwT = m->d0_CwT; p.P = m->P;
    p.x = x; p.len = len; p.offs = flat ? m->offs : nullptr;
    p.tags = tags; p.flat = flat; p.scores = scores; p.crf_scores = m->crf_scores;
    p.B = B; p.L = m->curL; p.S = m->S; p.SP = m->SP; p.R = m->R; p.Rp = m->Rp; p.RW = m->RW; p.RWp = m->RWp;
    p.V = m->V;
    p.K = m->K; p.Kp = m->Kp; p.Kc = m->Kc;
    p.full = full; p.use_crf = m->use_crf; p.o_idx = m->o_idx; p.threshold = m->threshold;
    const size_t lds = decomp0_score_lds_bytes(m->SP, m->Rp, m->RWp, m->Kc);
    int rc;
    if ((rc = raise_lds_limit(decomp0_score_kernel, lds))) return rc;
    KernelTimer kt(m, KERN_SCORE, s);
    decomp0_score_kernel<<<dim3((p.L + D0_TOK - 1) / D0_TOK, B), dim3(256), lds, s>>>(p);
    FARNN_HIP_TRY(hipGetLastError());
    if (m->use_crf) {
        ScoreParams v;
        memset(&v, 0, sizeof(v));
        v.trT = m->tr; v.len = len; v.offs = flat ? m->offs : nullptr; v.tags = tags; v.flat = flat;
        v.crf_scores = m->crf_scores; v.B = B; v.L = m->curL; v.K = m->K; v.Kp = m->Kp;
        v.full = full; v.use_crf = 1; v.o_idx = m->o_idx; v.threshold = m->threshold;
        if ((rc = launch_viterbi(m, v, B, s))) return rc;
    }
    return FARNN_OK;
}

static ScoreParams make_score_params(farnn_model *m, const int64_t *len, int B, int full, int32_t *tags,
                                     int64_t *flat, float *scores) {
    ScoreParams p;
    memset(&p, 0, sizeof(p));
    p.A = m->A; p.Bk = m->Bk; p.OT = m->OT; p.OTm = m->OTm; p.c16 = m->c16; p.P = m->P; p.trT = m->tr; p.len = len;
    p.offs = (flat && !m->prep_in_kernel) ? m->offs : nullptr; p.tags = tags; p.flat = flat; p.scores = scores;
    p.crf_scores = m->crf_scores;
    p.B = B; p.L = m->curL; p.S = m->S; p.SP = m->SP; p.K = m->K; p.Kp = m->Kp; p.Kc = m->Kc;
    p.kch = m->Kc / 64;
    p.full = full; p.use_crf = m->use_crf; p.o_idx = m->o_idx; p.threshold = m->threshold;
    p.dbg = tun(TUN_DBG);
    p.kz = (m->kind == KIND_IFST && m->use_crf && !tun(TUN_NOKZ)) ? m->C : 0;      // the library appended the two zero rows itself
    p.lm = m->lm;
    return p;
}

// the dense-block recurrence followed by scores + decode: ONE launch (the decode is the chain kernel's epilogue) when
// the decode is the threshold/argmax one and the geometry allows it, else the chain kernel + the score / Viterbi kernels
static int launch_chain_and_decode(farnn_model *m, const int64_t *x, const int64_t *len, int B, int L, int full,
                                   int32_t *tags, int64_t *flat, float *scores, hipStream_t s);

static int launch_score_decode(farnn_model *m, const int64_t *len, int B, int full, int32_t *tags,
                               int64_t *flat, float *scores, hipStream_t s) {
    ScoreParams p = make_score_params(m, len, B, full, tags, flat, scores);
    if (viterbi_can_fuse(m, p)) {          // stash -> scores -> Viterbi -> tags in one kernel
        KernelTimer kt(m, KERN_SCORE, s);
        return launch_viterbi(m, p, B, s, true);
    }
    int rc;
    if (p.lm.on && !p.P && !scores && !m->use_crf && m->S <= LM_MAXS * 2 && (B <= 1024 || !flat || p.offs)) {
        // K2l: the output matrix is a label map and only tags are asked for -- S multiply-adds and a scan per token, one workgroup
        // per sequence (score_decode.hip.h).  FARNN_NOLABELMAP=1 (no label map is built then) keeps the matrix form.
        KernelTimer kt(m, KERN_SCORE, s);
        label_map_score_kernel<<<B, LMS_WAVES * 64, 0, s>>>(p);
        FARNN_HIP_TRY(hipGetLastError());
        m->last_lm_score = true;
        return FARNN_OK;
    }
    m->last_lm_score = false;
    const size_t lds = score_lds_bytes(m->S, m->Kc);
    const dim3 grid((p.L + SCORE_TT - 1) / SCORE_TT, B), block(SCORE_WAVES * 64);
    KernelTimer kt(m, KERN_SCORE, s);
#define FARNN_LAUNCH_SCORE(KCH_)                                                              \
    do {                                                                                      \
        if ((rc = raise_lds_limit(score_tile_kernel<KCH_>, lds))) return rc;                  \
        score_tile_kernel<KCH_><<<grid, block, lds, s>>>(p);                                  \
    } while (0)
    switch (p.kch) {
        case 1: FARNN_LAUNCH_SCORE(1); break;
        case 2: FARNN_LAUNCH_SCORE(2); break;
        case 3: FARNN_LAUNCH_SCORE(3); break;
        default: FARNN_LAUNCH_SCORE(4); break;
    }
#undef FARNN_LAUNCH_SCORE
    FARNN_HIP_TRY(hipGetLastError());
    if (m->use_crf && (rc = launch_viterbi(m, p, B, s))) return rc;
    return FARNN_OK;
}

static int launch_chain_and_decode(farnn_model *m, const int64_t *x, const int64_t *len, int B, int L, int full,
                                   int32_t *tags, int64_t *flat, float *scores, hipStream_t s) {
    int rc;
    if (!m->use_crf) {
        const ScoreParams sp = make_score_params(m, len, B, full, tags, flat, scores);
        bool fused = false;
        if ((rc = launch_chain(m, x, len, B, L, full, s, &sp, &fused))) return rc;
        m->last_fused = fused;
        if (fused) return FARNN_OK;
    } else {
        // CRF decode: recurrence + scores + Viterbi in ONE launch (chain_viterbi.hip) where the register-fed recurrence applies and
        // the decode's LDS fits; else the recurrence kernel followed by the (fused score +) Viterbi kernel
        m->last_fused = false;
        const ScoreParams sp = make_score_params(m, len, B, full, tags, flat, scores);
        // The one launch (north_star: "decode fused into the same kernel") exists for S <= 108 and is parity-tested, but is NOT the
        // default (round 5): a compute unit then holds BOTH chains of a sequence, which run 1 390 cycles per step there against
        // 780-960 when a long chain shares its unit with a short one, and the decode waits behind them.  Measured at K = 130,
        // 256 x 64: S = 71 77.7 us in one launch against 72.9 in two (round 4; round 5's recurrence kernel: 69), S = 104 113.8 against
        // 95.7 (profiles/r04_*, r05_*).  FARNN_CV_ONE=1 selects it.
        // The production library does not carry the form's 48 kernels: it lives in the A/B build (build.py --probes, -DFARNN_AB).
#if defined(FARNN_AB)
        if (tun(TUN_CV_ONE) && m->rgeom.ok && !tun(TUN_NOREGS) && !tun(TUN_NOFUSE) && viterbi_can_fuse(m, sp) &&
            (B <= 1024 || !flat || sp.offs) && chain_viterbi_fits(m->curL, m->SP, m->rgeom.NP, m->K, m->Kp, m->lm.on != 0, m->rgeom.RQ)) {
            const RegsParams rp = make_regs_params(m, x, len, B, full);
            KernelTimer kt(m, KERN_CHAIN, s, /*ext=*/true);
            if ((rc = launch_chain_viterbi(rp, sp, m->semiring == FARNN_SEMIRING_MAX, s, kt.e0, kt.e1))) return rc;
            m->last_regs = true; m->last_fused = true;
            return FARNN_OK;
        }
#endif
        if ((rc = launch_chain(m, x, len, B, L, full, s))) return rc;
    }
    return launch_score_decode(m, len, B, full, tags, flat, scores, s);
}

// forward_RE's view of the scores (model_onehot.py:153-154): the `oo` column (the last one) capped at the threshold
__global__ void clamp_oo_column_kernel(float *scores, long long rows, int K, int col, float threshold) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) {                                  // torch.min: a NaN score stays a NaN (fminf would turn it into the threshold)
        const float x = scores[r * K + col];
        scores[r * K + col] = (x < threshold || x != x) ? x : threshold;
    }
}

static int tag_impl(farnn_model *m, const int64_t *x, const int64_t *lengths, int32_t B, int32_t L,
                    int32_t mode, int32_t *tags, int64_t *flat_tags, float *scores, void *stream);

extern "C" int farnn_tag(farnn_model *m, const int64_t *x, const int64_t *lengths, int32_t B, int32_t L,
                         int32_t mode, int32_t *tags, int64_t *flat_tags, float *scores, void *stream) {
    if (!m || !x || !lengths) return fail(FARNN_EINVAL, "tag: null model / x / lengths%s%s");
    if (B <= 0 || L <= 0) return fail(FARNN_EINVAL, "tag: B and L must be positive%s%s");
    if (mode != FARNN_MODE_LOCAL && mode != FARNN_MODE_FULL && mode != FARNN_MODE_RE)
        return fail(FARNN_EINVAL, "tag: bad mode%s%s");
    if (mode != FARNN_MODE_RE) return tag_impl(m, x, lengths, B, L, mode, tags, flat_tags, scores, stream);
    if (m->kind != KIND_IFST && m->kind != KIND_FST4 && m->kind != KIND_IND1)
        return fail(FARNN_EINVAL, "tag: FARNN_MODE_RE exists on the onehot models only (model_onehot.py:148)%s%s");
    int rc = tag_impl(m, x, lengths, B, L, FARNN_MODE_FULL, tags, flat_tags, scores, stream);
    if (rc || !scores) return rc;
    const long long rows = (long long)B * L;
    clamp_oo_column_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(
        scores, rows, m->K, m->C - 1, m->threshold);
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

static int tag_impl(farnn_model *m, const int64_t *x, const int64_t *lengths, int32_t B, int32_t L,
                    int32_t mode, int32_t *tags, int64_t *flat_tags, float *scores, void *stream) {
    TunScope tun_scope(&m->tun);
    FARNN_HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc;
    // One workspace per handle: two calls on different streams (a forward_score on torch's stream while batches submitted
    // through farnn_tag_host_submit are in flight on the handle's own stream, ...) would race on the stash and on the
    // hand-off words.  A stream switch costs one event: recorded NOW on the previous call's stream (i.e. behind all its work),
    // awaited by this call's stream.  Calls that stay on one stream pay nothing.
    // The FIRST switch records the event on the previous call's stream (which must still exist: include/farnn.h); from then on
    // the caller is known to alternate streams and every call leaves the handle's event behind itself on its OWN stream, so the
    // previous stream is never touched again.  A call that is being captured into a graph is not ordered against other streams.
    hipStreamCaptureStatus cap_ = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap_) != hipSuccess || cap_ != hipStreamCaptureStatusNone;
    if (m->have_last && m->last_stream != s && !capturing) {
        if (!m->ev_order) FARNN_HIP_TRY(hipEventCreateWithFlags(&m->ev_order, hipEventDisableTiming));
        if (!m->multi_stream) FARNN_HIP_TRY(hipEventRecord(m->ev_order, m->last_stream));
        FARNN_HIP_TRY(hipStreamWaitEvent(s, m->ev_order, 0));
        m->multi_stream = true;
    }
    m->last_stream = s; m->have_last = true;
    struct LeaveEvent {            // (multi-stream callers only) the handle's event behind this call's work, on this call's stream
        farnn_model *m; hipStream_t s; bool on;
        ~LeaveEvent() { if (on && m->ev_order) (void)hipEventRecord(m->ev_order, s); }
    } leave_event{m, s, m->multi_stream && !capturing};
    // the workspace arrays are strided with the CALL's L (every kernel writes whatever it later reads, pad columns
    // included), so (B, L) only have to fit the capacity: a loop whose batches vary in size or length allocates once
    if (B > m->wsB || L > m->wsL)
        if ((rc = farnn_reserve(m, B, L))) return rc;
    m->curL = L;
    const int full = mode == FARNN_MODE_FULL;
    m->prof_this_call = m->profiling > 0 && (m->calls++ % m->profiling) == 0;
    // batch preparation: flat-output offsets and the length-sorted launch order (full mode runs
    // every sequence for L steps, so there is nothing to balance)
    const bool want_order = !full && B > 2 && !tun(TUN_NOSORT);
    // the plain i-FST path needs no prep launch up to B = 1024: the chain workgroups select their sequence
    // by length rank themselves and the score workgroups sum the lengths in front of theirs
    m->prep_in_kernel = (m->kind == KIND_IFST || (m->kind == KIND_DECOMP && m->rows.ok)) && B <= 1024 && L <= 1023 &&
                        !tun(TUN_PREP);
    m->order_valid = want_order && !m->prep_in_kernel;
    m->sort_in_kernel = want_order && m->prep_in_kernel;
    // the decomposed independent=1 scoring kernel slices the batch by flat offsets even when no flat output is asked for
    const bool want_offs = flat_tags || (m->kind == KIND_DECOMP1 && m->d1_BSSp && !full);
    if ((want_offs || want_order) && !m->prep_in_kernel) {
        KernelTimer kt(m, KERN_PREP, s);
        if (B <= 1024) {
            int G = 1;
            while (G < 16 && B * G * 2 <= 1024) G *= 2;               // lanes per sequence
            batch_prep_small_kernel<<<1, round_up(B * G, 64), 0, s>>>(
                lengths, want_offs ? m->offs : nullptr, want_order ? m->order : nullptr, B, L, G);
        }
        else
            batch_prep_kernel<<<1, 1024, (size_t)(L + 2) * sizeof(int), s>>>(
                lengths, want_offs ? m->offs : nullptr, want_order ? m->order : nullptr, B, L);
        FARNN_HIP_TRY(hipGetLastError());
    }
    switch (m->kind) {
        case KIND_IFST:
            if (m->compact_on) {
                if (L > 1024) return fail(FARNN_ERANGE, "compact recurrence: more than 1024 positions%s%s");
                CompactParams cp;
                cp.bitsF = m->bmF; cp.bitsB = m->bmB; cp.wF = m->bmWF; cp.wB = m->bmWB; cp.o = m->o; cp.h0 = m->h0; cp.hT = m->hT;
                cp.mF = m->bmMF; cp.mB = m->bmMB; cp.xF = m->bmXF; cp.xB = m->bmXB; cp.tokoff = m->bmTok;
                cp.x = x; cp.len = lengths; cp.order = m->order_valid ? m->order : nullptr; cp.A = m->A; cp.Bk = m->Bk;
                cp.B = B; cp.L = L; cp.S = m->S; cp.SP = m->SP; cp.V = m->V; cp.nl = m->nl; cp.full = full; cp.dbg = tun(TUN_DBG);
                m->last_fused = false;
                {
                    // ONE launch (compact_tag.hip.h: both chains of a sequence in LDS, label-map scores, argmax decode) where it
                    // applies; FARNN_NOFUSE=1: round 2's two launches
                    const ScoreParams sp = make_score_params(m, lengths, B, full, tags, flat_tags, scores);
                    if (m->bmTok && sp.lm.on && !sp.P && !scores && !m->use_crf && !tun(TUN_NOFUSE) &&
                        compact_tag_fits(m->V, m->S, L) && (B <= 1024 || !flat_tags || sp.offs)) {
                        const size_t lds = (size_t)compact_tag_lds(L, m->bmNS).total * 4;
                        const int nlk = m->nl == FARNN_NL_NONE ? 0 : m->nl == FARNN_NL_RELU ? 1 : 2;
                        const int nw = (m->S + 31) / 32;                       // 32-bit words of a bitmap row in use
                        KernelTimer kt(m, KERN_CHAIN, s);
#define FARNN_LAUNCH_CT(NW_, NL_)                                                              \
                        do {                                                                   \
                            if ((rc = raise_lds_limit(compact_tag_kernel<NW_, NL_>, lds))) return rc; \
                            compact_tag_kernel<NW_, NL_><<<B, CT_WAVES * 64, lds, s>>>(cp, sp); \
                        } while (0)
#define FARNN_LAUNCH_CT_NW(NW_)                                                                \
                        do {                                                                   \
                            if (nlk == 0) FARNN_LAUNCH_CT(NW_, 0); else if (nlk == 1) FARNN_LAUNCH_CT(NW_, 1); else FARNN_LAUNCH_CT(NW_, 2); \
                        } while (0)
                        if (nw <= 1) FARNN_LAUNCH_CT_NW(1); else if (nw == 2) FARNN_LAUNCH_CT_NW(2);
                        else if (nw == 3) FARNN_LAUNCH_CT_NW(3); else FARNN_LAUNCH_CT_NW(4);
#undef FARNN_LAUNCH_CT_NW
#undef FARNN_LAUNCH_CT
                        FARNN_HIP_TRY(hipGetLastError());
                        m->last_fused = true;
                        return FARNN_OK;
                    }
                }
                {
                    KernelTimer kt(m, KERN_CHAIN, s);
                    if ((rc = launch_compact_chain(cp, m->bmNS, s))) return rc;
                }
                return launch_score_decode(m, lengths, B, full, tags, flat_tags, scores, s);
            }
            return launch_chain_and_decode(m, x, lengths, B, L, full, tags, flat_tags, scores, s);
        case KIND_FST4:
            if ((rc = launch_chain(m, x, lengths, B, L, full, s))) return rc;
            {
                KernelTimer kt(m, KERN_SCORE, s);
                return launch_fst4_score(m->A4, m->A, m->Bk, m->P, x, lengths, flat_tags ? m->offs : nullptr,
                                         tags, flat_tags, scores, B, m->curL, m->S, m->SP, m->C, m->Kc, full,
                                         m->o_idx, m->threshold, /*Oten*/ nullptr, m->V, s);
            }
        case KIND_IND1:
            if ((rc = launch_chain(m, x, lengths, B, L, full, s))) return rc;
            {
                KernelTimer kt(m, KERN_SCORE, s);
                return launch_fst4_score(m->Ms, m->A, m->Bk, m->P, x, lengths, flat_tags ? m->offs : nullptr,
                                         tags, flat_tags, scores, B, m->curL, m->S, m->SP, m->C, m->Kc, full,
                                         m->o_idx, m->threshold, m->Oten, m->V, s);
            }
        case KIND_DECOMP: {
            if (m->dense_decomp) return launch_chain_and_decode(m, x, lengths, B, L, full, tags, flat_tags, scores, s);
            bool fused = false;
            {
                KernelTimer kt(m, KERN_CHAIN, s);
                if (!m->use_crf) {
                    const ScoreParams sp = make_score_params(m, lengths, B, full, tags, flat_tags, scores);
                    if ((rc = launch_decomp_recurrence(m, x, lengths, B, full, s, &sp, &fused))) return rc;
                } else if ((rc = launch_decomp_recurrence(m, x, lengths, B, full, s))) return rc;
            }
            m->last_fused = fused;
            if (fused) return FARNN_OK;
            return launch_score_decode(m, lengths, B, full, tags, flat_tags, scores, s);
        }
        case KIND_DECOMP0: {
            {
                KernelTimer kt(m, KERN_CHAIN, s);
                if ((rc = launch_decomp_recurrence(m, x, lengths, B, full, s))) return rc;
            }
            return launch_decomp0_score(m, x, lengths, B, full, tags, flat_tags, scores, s);
        }
        case KIND_DECOMP1: {
            if (m->dense_decomp) {
                if ((rc = launch_chain(m, x, lengths, B, L, full, s))) return rc;
            } else {
                KernelTimer kt(m, KERN_CHAIN, s);
                if ((rc = launch_decomp_chain(m->dw, x, lengths, nullptr, m->A, m->Bk, B, m->curL, full, s))) return rc;
            }
            return launch_decomp1_score(m, x, lengths, B, full, tags, flat_tags, scores, s);
        }
        default:
            return fail(FARNN_EINVAL, "tag: unknown model kind%s%s");
    }
}

// ---- create from the automaton's edge list: dense tensors are scattered on the device ------------
struct EdgeTmp {            // device temporaries, freed on every path
    void *p[16] = {nullptr}; int n = 0;
    ~EdgeTmp() { for (int i = 0; i < n; i++) (void)hipFree(p[i]); }
    int get(void **q, size_t bytes) {
        FARNN_HIP_TRY(hipMalloc(q, bytes ? bytes : 4));
        p[n++] = *q;
        return FARNN_OK;
    }
    int zeros(float **q, size_t floats) {
        int rc = get((void **)q, floats * 4);
        if (rc) return rc;
        FARNN_HIP_TRY(hipMemset(*q, 0, floats * 4));
        return FARNN_OK;
    }
    int stage(const float *&ptr, size_t n_) {      // host array -> device copy (NULL stays NULL)
        if (!ptr) return FARNN_OK;
        float *dv = nullptr;
        int rc = get((void **)&dv, n_ * 4);
        if (rc) return rc;
        FARNN_HIP_TRY(hipMemcpy(dv, ptr, n_ * 4, hipMemcpyHostToDevice));
        ptr = dv;
        return FARNN_OK;
    }
};

static int scatter_edges(EdgeTmp &tmp, const farnn_edge_list *e, float *T, float *W, float *O, int V, int S, int C,
                         int mode) {
    if (!e || e->n_edges < 0 || (e->n_edges > 0 && (!e->word || !e->from || !e->to)))
        return fail(FARNN_EINVAL, "from_edges: edge arrays missing%s%s");
    const size_t ne = (size_t)e->n_edges;
    if (!ne) return FARNN_OK;
    int32_t *word = nullptr;
    float *val = nullptr;
    int *bad = nullptr;
    int rc;
    if ((rc = tmp.get((void **)&word, ne * 4 * 4))) return rc;           // word | from | to | label
    int32_t *from = word + ne, *to = from + ne, *label = to + ne;
    if ((rc = tmp.get((void **)&val, ne * 4))) return rc;
    if ((rc = tmp.get((void **)&bad, 4))) return rc;
    FARNN_HIP_TRY(hipMemset(bad, 0, 4));
    FARNN_HIP_TRY(hipMemcpy(word, e->word, ne * 4, hipMemcpyHostToDevice));
    FARNN_HIP_TRY(hipMemcpy(from, e->from, ne * 4, hipMemcpyHostToDevice));
    FARNN_HIP_TRY(hipMemcpy(to, e->to, ne * 4, hipMemcpyHostToDevice));
    if (e->label) FARNN_HIP_TRY(hipMemcpy(label, e->label, ne * 4, hipMemcpyHostToDevice));
    if (e->val) FARNN_HIP_TRY(hipMemcpy(val, e->val, ne * 4, hipMemcpyHostToDevice));
    scatter_edges_kernel<<<(unsigned)((ne + 255) / 256), 256>>>(word, from, to, e->label ? label : nullptr,
                                                                 e->val ? val : nullptr, (long long)ne, T, W, O,
                                                                 V, S, C, mode, bad);
    FARNN_HIP_TRY(hipGetLastError());
    int hbad = 0;
    FARNN_HIP_TRY(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    if (hbad) return fail(FARNN_EINVAL, "from_edges: an edge has a word, state or label index out of range%s%s");
    return FARNN_OK;
}

extern "C" int farnn_onehot_ifst_create_from_edges(const farnn_onehot_ifst_desc *b, const farnn_edge_list *e,
                                                   int device, farnn_model **out) {
    if (!b || !out) return fail(FARNN_EINVAL, "null argument%s%s");
    *out = nullptr;
    if (b->V <= 0 || b->S <= 0 || b->C <= 0) return fail(FARNN_EINVAL, "ifst_from_edges: V, S, C must be positive%s%s");
    int rc = select_device(device);
    if (rc) return rc;
    EdgeTmp tmp;
    float *T = nullptr, *W = nullptr, *O = nullptr;
    if ((rc = tmp.zeros(&T, (size_t)b->V * b->S * b->S)) || (rc = tmp.zeros(&W, (size_t)b->S * b->S)) ||
        (rc = tmp.zeros(&O, (size_t)b->C * b->S))) return rc;
    if ((rc = scatter_edges(tmp, e, T, W, O, b->V, b->S, b->C, 0))) return rc;
    farnn_onehot_ifst_desc full = *b;
    if (!b->weights_on_device) {
        const size_t K = (size_t)b->C + (b->use_crf ? 2 : 0);
        if ((rc = tmp.stage(full.h0, b->S)) || (rc = tmp.stage(full.hT, b->S)) ||
            (rc = tmp.stage(full.P, (size_t)b->C * b->C)) || (rc = tmp.stage(full.crf_trans, K * K))) return rc;
    }
    full.T = T; full.W = W; full.O = O; full.weights_on_device = 1;
    return farnn_onehot_ifst_create(&full, device, out);
}

extern "C" int farnn_onehot_ifst_create_compact(const farnn_onehot_ifst_desc *b, const farnn_edge_list *e, int device,
                                                farnn_model **out) {
    if (!b || !out || !e) return fail(FARNN_EINVAL, "null argument%s%s");
    *out = nullptr;
    if (b->V <= 0 || b->S <= 0 || b->C <= 0) return fail(FARNN_EINVAL, "ifst_create_compact: V, S, C must be positive%s%s");
    if (e->n_edges < 0 || (e->n_edges > 0 && (!e->word || !e->from || !e->to)))
        return fail(FARNN_EINVAL, "ifst_create_compact: edge arrays missing%s%s");
    int rc = select_device(device);
    if (rc) return rc;
    EdgeTmp tmp;
    float *O = nullptr;
    if ((rc = tmp.zeros(&O, (size_t)b->C * b->S))) return rc;
    if ((rc = scatter_edges(tmp, e, nullptr, nullptr, O, b->V, b->S, b->C, 0))) return rc;       // labels -> O only
    // the edge arrays once more on the device, for the bitmap scatter
    const size_t ne = (size_t)e->n_edges;
    int32_t *dw = nullptr;
    float *dv = nullptr;
    if ((rc = tmp.get((void **)&dw, (ne ? ne : 1) * 3 * 4))) return rc;
    if (ne) {
        FARNN_HIP_TRY(hipMemcpy(dw, e->word, ne * 4, hipMemcpyHostToDevice));
        FARNN_HIP_TRY(hipMemcpy(dw + ne, e->from, ne * 4, hipMemcpyHostToDevice));
        FARNN_HIP_TRY(hipMemcpy(dw + 2 * ne, e->to, ne * 4, hipMemcpyHostToDevice));
        if (e->val) {
            if ((rc = tmp.get((void **)&dv, ne * 4))) return rc;
            FARNN_HIP_TRY(hipMemcpy(dv, e->val, ne * 4, hipMemcpyHostToDevice));
        }
    }
    DevEdges de{dw, dw + ne, dw + 2 * ne, dv, (long long)ne};
    farnn_onehot_ifst_desc full = *b;
    if (!b->weights_on_device) {
        const size_t K = (size_t)b->C + (b->use_crf ? 2 : 0);
        if ((rc = tmp.stage(full.h0, b->S)) || (rc = tmp.stage(full.hT, b->S)) ||
            (rc = tmp.stage(full.P, (size_t)b->C * b->C)) || (rc = tmp.stage(full.crf_trans, K * K))) return rc;
    }
    full.T = nullptr; full.W = nullptr; full.O = O; full.weights_on_device = 1;
    return ifst_create_impl(&full, device, out, &de);
}

extern "C" int farnn_onehot_fst4_create_from_edges(const farnn_onehot_fst4_desc *b, const farnn_edge_list *e,
                                                   int device, farnn_model **out) {
    if (!b || !out) return fail(FARNN_EINVAL, "null argument%s%s");
    *out = nullptr;
    if (b->V <= 0 || b->S <= 0 || b->C <= 0) return fail(FARNN_EINVAL, "fst4_from_edges: V, S, C must be positive%s%s");
    int rc = select_device(device);
    if (rc) return rc;
    EdgeTmp tmp;
    float *T4 = nullptr, *W4 = nullptr;
    if ((rc = tmp.zeros(&T4, (size_t)b->V * b->C * b->S * b->S)) || (rc = tmp.zeros(&W4, (size_t)b->C * b->S * b->S)))
        return rc;
    if ((rc = scatter_edges(tmp, e, T4, W4, nullptr, b->V, b->S, b->C, 1))) return rc;
    farnn_onehot_fst4_desc full = *b;
    if (!b->weights_on_device)
        if ((rc = tmp.stage(full.h0, b->S)) || (rc = tmp.stage(full.hT, b->S)) ||
            (rc = tmp.stage(full.P, (size_t)b->C * b->C))) return rc;
    full.T4 = T4; full.W4 = W4; full.weights_on_device = 1;
    return farnn_onehot_fst4_create(&full, device, out);
}

extern "C" int farnn_onehot_ind1_create_from_edges(const farnn_onehot_ind1_desc *b, const farnn_edge_list *e,
                                                   int device, farnn_model **out) {
    if (!b || !out) return fail(FARNN_EINVAL, "null argument%s%s");
    *out = nullptr;
    if (b->V <= 0 || b->S <= 0 || b->C <= 0) return fail(FARNN_EINVAL, "ind1_from_edges: V, S, C must be positive%s%s");
    int rc = select_device(device);
    if (rc) return rc;
    EdgeTmp tmp;
    float *T = nullptr, *W = nullptr, *Oten = nullptr;
    if ((rc = tmp.zeros(&T, (size_t)b->V * b->S * b->S)) || (rc = tmp.zeros(&W, (size_t)b->S * b->S)) ||
        (rc = tmp.zeros(&Oten, (size_t)b->C * b->S * b->S))) return rc;
    if ((rc = scatter_edges(tmp, e, T, W, Oten, b->V, b->S, b->C, 2))) return rc;
    farnn_onehot_ind1_desc full = *b;
    if (!b->weights_on_device)
        if ((rc = tmp.stage(full.h0, b->S)) || (rc = tmp.stage(full.hT, b->S)) ||
            (rc = tmp.stage(full.P, (size_t)b->C * b->C))) return rc;
    full.T = T; full.W = W; full.Oten = Oten; full.weights_on_device = 1;
    return farnn_onehot_ind1_create(&full, device, out);
}

// ---- create: onehot FST 4-D -------------------------------------------------------------------
extern "C" int farnn_onehot_fst4_create(const farnn_onehot_fst4_desc *d, int device, farnn_model **out) {
    if (!d || !out) return fail(FARNN_EINVAL, "null argument%s%s");
    *out = nullptr;
    if (d->V <= 0 || d->S <= 0 || d->C <= 0 || !d->T4 || !d->W4 || !d->h0 || !d->hT)
        return fail(FARNN_EINVAL, "onehot_fst4: sizes must be positive and T4/W4/h0/hT non-null%s%s");
    int rc = select_device(device);
    if (rc) return rc;
    farnn_model *m = new (std::nothrow) farnn_model();
    if (!m) return fail(FARNN_ENOMEM, "host allocation failed%s%s");
    TunScope tun_scope(&m->tun);
    m->kind = KIND_FST4; m->device = device;
    m->V = d->V; m->S = d->S; m->C = d->C; m->K = d->C; m->Kp = round_up(m->K, 4); m->Kc = round_up(m->K, 64);
    m->nl = FARNN_NL_RELU;                       // relu is unconditional (model_onehot.py:93-94)
    m->semiring = d->semiring; m->threshold = d->threshold; m->o_idx = d->o_idx;
    pick_chain_geometry(m);
    m->chain_ks = tun(TUN_KS);
    m->SP = m->geom.SP;
    const int od = d->weights_on_device;
    auto bail = [&](int code) { farnn_destroy(m); return code; };
    if (m->K > 1024) return bail(fail(FARNN_ERANGE, "more than 1024 label columns%s%s"));
    if (m->geom.NCH > 4) return bail(fail(FARNN_ERANGE, "more than 1024 states%s%s"));
    {
        const size_t nT = (size_t)m->V * m->C * m->S * m->S;
        TmpDev T4, W4;
        if ((rc = T4.init(d->T4, nT, od))) return bail(rc);
        if ((rc = W4.init(d->W4, (size_t)m->C * m->S * m->S, od))) return bail(rc);
        const size_t nM = (size_t)m->V * m->geom.SR * m->SP;
        if ((rc = dev_alloc(m, (void **)&m->Mf, nM * 4))) return bail(rc);
        if ((rc = dev_alloc(m, (void **)&m->Mb, nM * 4))) return bail(rc);
        if ((rc = dev_alloc(m, (void **)&m->A4, (size_t)m->V * m->C * m->S * m->SP * 4))) return bail(rc);
        if ((rc = launch_premix_fst4(T4.p, W4.p, m->Mf, m->Mb, m->A4, m->V, m->C, m->S, m->SP, m->geom.SR)))
            return bail(rc);
    }
    if ((rc = dev_upload(m, &m->h0, d->h0, m->S, m->SP, od))) return bail(rc);
    if ((rc = dev_upload(m, &m->hT, d->hT, m->S, m->SP, od))) return bail(rc);
    if ((rc = setup_priority(m, d->P, od))) return bail(rc);
    *out = m;
    return FARNN_OK;
}

// ---- create: onehot independent=1 -------------------------------------------------------------
extern "C" int farnn_onehot_ind1_create(const farnn_onehot_ind1_desc *d, int device, farnn_model **out) {
    if (!d || !out) return fail(FARNN_EINVAL, "null argument%s%s");
    *out = nullptr;
    if (d->V <= 0 || d->S <= 0 || d->C <= 0 || !d->T || !d->W || !d->Oten || !d->h0 || !d->hT)
        return fail(FARNN_EINVAL, "onehot_ind1: sizes must be positive and T/W/Oten/h0/hT non-null%s%s");
    int rc = select_device(device);
    if (rc) return rc;
    farnn_model *m = new (std::nothrow) farnn_model();
    if (!m) return fail(FARNN_ENOMEM, "host allocation failed%s%s");
    TunScope tun_scope(&m->tun);
    m->kind = KIND_IND1; m->device = device;
    m->V = d->V; m->S = d->S; m->C = d->C; m->K = d->C; m->Kp = round_up(m->K, 4); m->Kc = round_up(m->K, 64);
    m->nl = FARNN_NL_RELU;                       // relu always (model_onehot.py:266, :278)
    m->semiring = d->semiring; m->threshold = d->threshold; m->o_idx = d->o_idx;
    m->mask_by_output = d->mask_by_output;
    pick_chain_geometry(m);
    m->chain_ks = tun(TUN_KS);
    m->SP = m->geom.SP;
    const int od = d->weights_on_device;
    auto bail = [&](int code) { farnn_destroy(m); return code; };
    if (m->K > 1024) return bail(fail(FARNN_ERANGE, "more than 1024 label columns%s%s"));
    if (m->geom.NCH > 4) return bail(fail(FARNN_ERANGE, "more than 1024 states%s%s"));
    {
        const size_t nT = (size_t)m->V * m->S * m->S;
        TmpDev T, W, Ot;
        if ((rc = T.init(d->T, nT, od))) return bail(rc);
        if ((rc = W.init(d->W, (size_t)m->S * m->S, od))) return bail(rc);
        if ((rc = Ot.init(d->Oten, (size_t)m->C * m->S * m->S, od))) return bail(rc);
        const size_t nM = (size_t)m->V * m->geom.SR * m->SP;     // chain blocks (SR rows)
        const size_t nS = (size_t)m->V * m->S * m->SP;           // scoring blocks (S rows)
        if ((rc = dev_alloc(m, (void **)&m->Ms, nS * 4))) return bail(rc);
        if ((rc = upload_padded(m, &m->Oten, Ot.p, m->C * m->S, m->S, m->C * m->S, m->SP, 1))) return bail(rc);
        float *osum = nullptr;
        if (m->mask_by_output) {
            if ((rc = dev_alloc(m, (void **)&osum, (size_t)m->S * m->S * 4))) return bail(rc);
            colsum_kernel<<<(m->S * m->S + 255) / 256, 256>>>(Ot.p, osum, m->C, m->S * m->S);
            FARNN_HIP_TRY(hipGetLastError());
            if ((rc = dev_alloc(m, (void **)&m->Mf, nM * 4))) return bail(rc);
            if ((rc = dev_alloc(m, (void **)&m->Mb, nM * 4))) return bail(rc);
            if ((rc = launch_premix(T.p, W.p, osum, m->Mf, m->Mb, m->V, m->S, m->SP, m->geom.SR))) return bail(rc);
        } else {
            if ((rc = dev_alloc(m, (void **)&m->Mf, nM * 4))) return bail(rc);
            if ((rc = dev_alloc(m, (void **)&m->Mb, nM * 4))) return bail(rc);
            if ((rc = launch_premix(T.p, W.p, nullptr, m->Mf, m->Mb, m->V, m->S, m->SP, m->geom.SR))) return bail(rc);
        }
        if ((rc = launch_premix(T.p, W.p, nullptr, m->Ms, nullptr, m->V, m->S, m->SP, m->S))) return bail(rc);
    }
    if ((rc = dev_upload(m, &m->h0, d->h0, m->S, m->SP, od))) return bail(rc);
    if ((rc = dev_upload(m, &m->hT, d->hT, m->S, m->SP, od))) return bail(rc);
    if ((rc = setup_priority(m, d->P, od))) return bail(rc);
    *out = m;
    return FARNN_OK;
}

// ---- shared by the three decomposed creates: factor tables of the recurrence ---------------------
struct GateSrc { int farnn; const float *Wss1, *Wrs1, *bs1, *Wss2, *Wrs2, *bs2; };

static int check_gates(const GateSrc &g, const char *who) {
    if (g.farnn < 0 || g.farnn > 2) return fail(FARNN_EINVAL, "%s: farnn must be 0, 1 or 2%s", who, "");
    if (g.farnn >= 1 && (!g.Wss1 || !g.Wrs1 || !g.bs1))
        return fail(FARNN_EINVAL, "%s: farnn>=1 needs Wss1/Wrs1/bs1%s", who, "");
    if (g.farnn == 2 && (!g.Wss2 || !g.Wrs2 || !g.bs2))
        return fail(FARNN_EINVAL, "%s: farnn==2 needs Wss2/Wrs2/bs2%s", who, "");
    return FARNN_OK;
}

// Vgen [V,R], S1/S2 [S,R], W [S,S] dense row-major; each with its own host|device flag.
static int upload_chain_factors(farnn_model *m, const float *Vgen, int odV, const float *S1, const float *S2,
                                int odS, const float *W, int odW, const GateSrc &g, int odG) {
    DecompWeights &w = m->dw;
    int rc;
    w.S = m->S; w.SP = m->SP; w.R = m->R; w.Rp = m->Rp; w.V = m->V;
    w.farnn = g.farnn; w.nl = m->nl; w.semiring = m->semiring; w.sig_k = m->sig_k;
    float *tmp = nullptr;
    if ((rc = upload_padded(m, &tmp, Vgen, m->V, m->R, m->V, m->Rp, odV))) return rc; w.Vgen = tmp;
    if ((rc = upload_padded(m, &tmp, S1, m->S, m->R, m->S, m->Rp, odS))) return rc; w.S1 = tmp;
    if ((rc = upload_padded(m, &tmp, S2, m->S, m->R, m->S, m->Rp, odS))) return rc; w.S2 = tmp;
    if ((rc = upload_transposed(m, &tmp, S1, m->S, m->R, m->SP, odS))) return rc; w.S1T = tmp;
    if ((rc = upload_transposed(m, &tmp, S2, m->S, m->R, m->SP, odS))) return rc; w.S2T = tmp;
    if ((rc = upload_padded(m, &tmp, W, m->S, m->S, m->S, m->SP, odW))) return rc; w.W = tmp;
    if ((rc = upload_transposed(m, &tmp, W, m->S, m->S, m->SP, odW))) return rc; w.WT = tmp;
    if (g.farnn >= 1) {
        if ((rc = upload_padded(m, &tmp, g.Wss1, m->S, m->S, m->S, m->SP, odG))) return rc; w.Wss1 = tmp;
        if ((rc = upload_padded(m, &tmp, g.Wrs1, m->R, m->S, m->R, m->SP, odG))) return rc; w.Wrs1 = tmp;
        if ((rc = dev_upload(m, &tmp, g.bs1, m->S, m->SP, odG))) return rc; w.bs1 = tmp;
    }
    if (g.farnn == 2) {
        if ((rc = upload_padded(m, &tmp, g.Wss2, m->S, m->S, m->S, m->SP, odG))) return rc; w.Wss2 = tmp;
        if ((rc = upload_padded(m, &tmp, g.Wrs2, m->R, m->S, m->R, m->SP, odG))) return rc; w.Wrs2 = tmp;
        if ((rc = dev_upload(m, &tmp, g.bs2, m->S, m->SP, odG))) return rc; w.bs2 = tmp;
    }
    return FARNN_OK;
}

static int upload_ones_o(farnn_model *m) {
    std::vector<float> ones((size_t)m->SP, 1.0f);
    int rc = dev_upload(m, &m->o, ones.data(), m->SP, m->SP, 0);
    if (rc) return rc;
    m->dw.o = m->o;
    return FARNN_OK;
}

// ---- create: decomposed i-FST ------------------------------------------------------------------
// od_vgen / od_s12: Vgen resp. S1, S2 are device pointers whatever d->weights_on_device says (the folded creator)
static int decomp_ifst_create_impl(const farnn_decomp_ifst_desc *d, int device, farnn_model **out, int od_vgen, int od_s12) {
    if (!d || !out) return fail(FARNN_EINVAL, "null argument%s%s");
    *out = nullptr;
    if (d->V <= 0 || d->S <= 0 || d->R <= 0 || d->K <= 0 || !d->Vgen || !d->S1 || !d->S2 || !d->W ||
        !d->Cout || !d->h0 || !d->hT)
        return fail(FARNN_EINVAL, "decomp_ifst: sizes must be positive and factor pointers non-null%s%s");
    const GateSrc gates{d->farnn, d->Wss1, d->Wrs1, d->bs1, d->Wss2, d->Wrs2, d->bs2};
    if (int grc = check_gates(gates, "decomp_ifst")) return grc;
    if (d->nl < 0 || d->nl > FARNN_NL_RELUTANH) return fail(FARNN_EINVAL, "decomp_ifst: bad nl%s%s");
    int rc = select_device(device);
    if (rc) return rc;
    farnn_model *m = new (std::nothrow) farnn_model();
    if (!m) return fail(FARNN_ENOMEM, "host allocation failed%s%s");
    TunScope tun_scope(&m->tun);
    m->kind = KIND_DECOMP; m->device = device;
    m->V = d->V; m->S = d->S; m->R = d->R; m->K = d->K; m->Kp = round_up(d->K, 4); m->Kc = round_up(d->K, 64);
    m->C = d->use_crf ? d->K - 2 : d->K;
    m->SP = round_up(d->S, 4); m->Rp = round_up(d->R, 4);
    m->nl = d->nl; m->semiring = d->semiring; m->threshold = d->threshold; m->o_idx = d->o_idx;
    m->use_crf = d->use_crf ? 1 : 0; m->farnn_gate = d->farnn; m->sig_k = d->sigmoid_exponent;
    const int od = d->weights_on_device;
    auto bail = [&](int code) { farnn_destroy(m); return code; };
    if (m->K > 64 * SCORE_KCH) return bail(fail(FARNN_ERANGE, "more than 256 label columns%s%s"));
    if (m->S > 1024 || m->R > 4096) return bail(fail(FARNN_ERANGE, "decomp_ifst: S<=1024, R<=4096%s%s"));
    DecompWeights &w = m->dw;
    if ((rc = upload_chain_factors(m, d->Vgen, od | od_vgen, d->S1, d->S2, od | od_s12, d->W, od, gates, od))) return bail(rc);
    {   // o = sum_k Cout[k,:] (CE1, model_decompose_single.py:232); OT = Cout^T
        TmpDev Co;
        if ((rc = Co.init(d->Cout, (size_t)m->K * m->S, od))) return bail(rc);
        if ((rc = dev_alloc(m, (void **)&m->o, (size_t)m->SP * 4))) return bail(rc);
        if ((rc = dev_alloc(m, (void **)&m->OT, round_up_sz((size_t)m->S * m->Kc * 4, 1024)))) return bail(rc);
        FARNN_HIP_TRY(hipMemset(m->o, 0, (size_t)m->SP * 4));
        FARNN_HIP_TRY(hipMemset(m->OT, 0, round_up_sz((size_t)m->S * m->Kc * 4, 1024)));
        colsum_kernel<<<(m->S + 255) / 256, 256>>>(Co.p, m->o, m->K, m->S);
        int n = m->K * m->S;
        transpose_pad_kernel<<<(n + 255) / 256, 256>>>(Co.p, m->OT, m->K, m->S, m->Kc);
        FARNN_HIP_TRY(hipGetLastError());
        if ((rc = build_ot_image(m))) return bail(rc);
        FARNN_HIP_TRY(hipDeviceSynchronize());
        if ((rc = build_label_map(m))) return bail(rc);
        w.o = m->o;
    }
    if ((rc = dev_upload(m, &m->h0, d->h0, m->S, m->SP, od))) return bail(rc);
    if ((rc = dev_upload(m, &m->hT, d->hT, m->S, m->SP, od))) return bail(rc);
    w.h0 = m->h0; w.hT = m->hT;
    if ((rc = setup_priority(m, d->P, od))) return bail(rc);
    if ((rc = setup_crf(m, d->crf_trans, od))) return bail(rc);
    if ((rc = build_rows_pack(m))) return bail(rc);
    if ((rc = build_dense_blocks(m))) return bail(rc);
    *out = m;
    return FARNN_OK;
}

extern "C" int farnn_decomp_ifst_create(const farnn_decomp_ifst_desc *d, int device, farnn_model **out) {
    return decomp_ifst_create_impl(d, device, out, 0, 0);
}

// ---- the word table and --normalize_automata on the device ------------------------------------------------
// avg[c] = ||M[:, c]||_ord / rows   (utils.get_average, '-rank' modes; reference utils.py:202-225)
__global__ void col_avg_norm_kernel(const float *M, int rows, int cols, int ld, int ord, float *avg) {
    const int c = blockIdx.x;
    __shared__ float red[256];
    float acc = 0.0f;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        const float v = M[(long long)r * ld + c];
        acc += ord == 1 ? fabsf(v) : v * v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) avg[c] = (ord == 1 ? red[0] : sqrtf(red[0])) / (float)rows;
}

// factor = cbrt(v_avg s1_avg s2_avg); scale[0][c] = factor / v_avg, [1][c] = factor / s1_avg, [2][c] = factor / s2_avg  (init_params.py:285-297)
__global__ void norm_scales_kernel(const float *avg /*[3][R]*/, float *scale /*[3][R]*/, int R) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= R) return;
    const float f = cbrtf(avg[c] * avg[R + c] * avg[2 * R + c]);
    scale[c] = f / avg[c]; scale[R + c] = f / avg[R + c]; scale[2 * R + c] = f / avg[2 * R + c];
}

// G[i][j] = sum_r M[r][i] M[r][j]  (the Gram matrix of the columns, doubles: its largest eigenvalue is the squared spectral norm)
__global__ void gram_kernel(const float *M, int rows, int cols, double *G) {
    const int i = blockIdx.x, j = blockIdx.y;
    if (j > i) return;
    __shared__ double red[256];
    double acc = 0.0;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) acc += (double)M[(long long)r * cols + i] * (double)M[(long long)r * cols + j];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { G[(long long)i * cols + j] = red[0]; G[(long long)j * cols + i] = red[0]; }
}

// largest eigenvalue of a symmetric positive semi-definite n x n matrix (a Gram matrix; doubles; n <= a few hundred: create time).
// Only the top eigenvalue is needed (the spectral norm of a factor matrix): power iteration on A -- with A squared a few times
// first, so that the eigenvalue ratio that governs convergence is raised to the 2^k-th power -- instead of diagonalising the
// matrix (a cyclic Jacobi sweep is n^2/2 rotations of 4n updates; 60 sweeps at n = 250 were seconds of host time per create).
static double gram_largest_eigenvalue(std::vector<double> &A, int n) {
    if (n <= 0) return 0.0;
    auto matmul_sq = [&](std::vector<double> &M) {       // M <- M . M / trace-scale (keeps the numbers in range)
        double tr = 0.0;
        for (int i = 0; i < n; i++) tr += M[(size_t)i * n + i];
        if (!(tr > 0.0)) return 0.0;
        std::vector<double> N((size_t)n * n, 0.0);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < n; k++) {
                const double a = M[(size_t)i * n + k] / tr;
                if (a == 0.0) continue;
                for (int j = 0; j < n; j++) N[(size_t)i * n + j] += a * (M[(size_t)k * n + j] / tr);
            }
        M.swap(N);
        return tr;
    };
    // lambda_max(A) from the Rayleigh quotient of the dominant eigenvector of A^(2^k): same eigenvector
    std::vector<double> B = A;
    for (int k = 0; k < 6; k++)
        if (!(matmul_sq(B) > 0.0)) return 0.0;
    std::vector<double> v((size_t)n), w((size_t)n);
    for (int i = 0; i < n; i++) v[i] = 1.0 + 1e-3 * ((i * 2654435761u) % 1000);   // (not orthogonal to anything in particular)
    double lam = 0.0;
    for (int it = 0; it < 200; it++) {
        const std::vector<double> &M = it < 8 ? B : A;   // a few steps on A^(64) to land on the eigenvector, then refine on A itself
        double nrm = 0.0;
        for (int i = 0; i < n; i++) { double acc = 0.0; for (int j = 0; j < n; j++) acc += M[(size_t)i * n + j] * v[j]; w[i] = acc; nrm += acc * acc; }
        nrm = sqrt(nrm);
        if (!(nrm > 0.0)) return 0.0;
        for (int i = 0; i < n; i++) v[i] = w[i] / nrm;
        if (it >= 8) {
            double num = 0.0;                            // Rayleigh quotient v^T A v (v has unit length)
            for (int i = 0; i < n; i++) { double acc = 0.0; for (int j = 0; j < n; j++) acc += A[(size_t)i * n + j] * v[j]; num += v[i] * acc; }
            if (fabs(num - lam) <= 1e-14 * fabs(num)) { lam = num; break; }
            lam = num;
        }
    }
    return lam;
}

__global__ void scale_cols_kernel(float *M, long long n, int cols, const float *scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) M[i] *= scale[i % cols];
}

// Vgen[w][r] = V[w][r] cv[r] beta[r] + nl_add( sum_d E[w][d] G[d][r] cv[r] ) (1 - beta[r])      (model_decompose.py:222-241)
// cv = the normalisation scale of V_embed's columns (1 without): G = pinv(E) V_embed is linear in V_embed's columns
__global__ void fold_vgen_kernel(const float *Vemb, const float *E, const float *G, const float *beta, const float *cv,
                                 float *Vgen, int V, int R, int D, int add_nl) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)V * R) return;
    const int w = (int)(idx / R), r = (int)(idx % R);
    float g = 0.0f;
    for (int d = 0; d < D; d++) g = fmaf(E[(long long)w * D + d], G[(long long)d * R + r], g);
    const float c = cv ? cv[r] : 1.0f;
    const float b = beta[r];
    Vgen[idx] = Vemb[idx] * c * b + apply_nl(g * c, add_nl) * (1.0f - b);
}

extern "C" int farnn_decomp_ifst_create_folded(const farnn_decomp_ifst_desc *d, const farnn_vgen_fold *f, int device,
                                               farnn_model **out) {
    if (!d || !f || !out) return fail(FARNN_EINVAL, "null argument%s%s");
    *out = nullptr;
    if (d->V <= 0 || d->S <= 0 || d->R <= 0 || f->D <= 0 || !f->V_embed || !f->E || !f->G || !f->beta || !d->S1 || !d->S2)
        return fail(FARNN_EINVAL, "decomp_ifst_create_folded: V_embed / E / G / beta / S1 / S2 and positive sizes needed%s%s");
    if (f->add_nl < FARNN_NL_NONE || f->add_nl > FARNN_NL_SIGMOID) return fail(FARNN_EINVAL, "decomp_ifst_create_folded: bad add_nl%s%s");
    if (f->normalize < FARNN_NORM_NONE || f->normalize > FARNN_NORM_L2_RANK)
        return fail(FARNN_EINVAL, "decomp_ifst_create_folded: bad normalize mode%s%s");
    int rc = select_device(device);
    if (rc) return rc;
    const size_t V = d->V, R = d->R, S = d->S, D = f->D;
    EdgeTmp tmp;
    float *Vd = nullptr, *S1d = nullptr, *S2d = nullptr, *Vgen = nullptr, *avg = nullptr;
    TmpDev E, G, beta;
    auto copy_in = [&](float **dst, const float *src, size_t n, int on_dev) -> int {
        int r2 = tmp.get((void **)dst, n * 4);
        if (r2) return r2;
        FARNN_HIP_TRY(hipMemcpy(*dst, src, n * 4, on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
        return FARNN_OK;
    };
    if ((rc = copy_in(&Vd, f->V_embed, V * R, f->on_device)) || (rc = copy_in(&S1d, d->S1, S * R, d->weights_on_device)) ||
        (rc = copy_in(&S2d, d->S2, S * R, d->weights_on_device)) || (rc = E.init(f->E, V * D, f->on_device)) ||
        (rc = G.init(f->G, D * R, f->on_device)) || (rc = beta.init(f->beta, R, f->on_device)) ||
        (rc = tmp.get((void **)&Vgen, V * R * 4)) || (rc = tmp.get((void **)&avg, 6 * R * 4))) return rc;
    const float *cv = nullptr;
    if (f->normalize == FARNN_NORM_L1 || f->normalize == FARNN_NORM_L2) {
        // whole-matrix modes (utils.py:211-216): numpy's matrix 1-norm (the largest column sum) or 2-norm (the spectral norm) over
        // the element count -- one scalar per matrix.  The column sums / the R x R Gram matrix are formed on the device; R floats /
        // R x R doubles come back, never anything of size V x R.
        const float *mats[3] = {Vd, S1d, S2d};
        const size_t rows[3] = {V, S, S};
        double avgs[3];
        if (f->normalize == FARNN_NORM_L1) {
            for (int q = 0; q < 3; q++) col_avg_norm_kernel<<<(unsigned)R, 256>>>(mats[q], (int)rows[q], (int)R, (int)R, 1, avg + q * R);
            std::vector<float> hv(3 * R);
            FARNN_HIP_TRY(hipMemcpy(hv.data(), avg, 3 * R * 4, hipMemcpyDeviceToHost));
            for (int q = 0; q < 3; q++) {
                float mx = 0.0f;
                for (size_t c = 0; c < R; c++) mx = hv[q * R + c] > mx ? hv[q * R + c] : mx;       // (column sum / rows)
                avgs[q] = (double)mx / (double)R;
            }
        } else {
            double *Gd = nullptr;
            if ((rc = tmp.get((void **)&Gd, R * R * 8))) return rc;
            std::vector<double> Gh(R * R);
            for (int q = 0; q < 3; q++) {
                gram_kernel<<<dim3((unsigned)R, (unsigned)R), 256>>>(mats[q], (int)rows[q], (int)R, Gd);
                FARNN_HIP_TRY(hipMemcpy(Gh.data(), Gd, R * R * 8, hipMemcpyDeviceToHost));
                avgs[q] = sqrt(gram_largest_eigenvalue(Gh, (int)R)) / ((double)rows[q] * (double)R);
            }
        }
        if (!(avgs[0] > 0.0) || !(avgs[1] > 0.0) || !(avgs[2] > 0.0))
            return fail(FARNN_EINVAL, "decomp_ifst_create_folded: a factor matrix has zero norm%s%s");
        const double fac = cbrt(avgs[0] * avgs[1] * avgs[2]);
        std::vector<float> sc(3 * R);
        for (int q = 0; q < 3; q++)
            for (size_t c = 0; c < R; c++) sc[q * R + c] = (float)(fac / avgs[q]);
        FARNN_HIP_TRY(hipMemcpy(avg + 3 * R, sc.data(), 3 * R * 4, hipMemcpyHostToDevice));
        scale_cols_kernel<<<(unsigned)((S * R + 255) / 256), 256>>>(S1d, (long long)(S * R), (int)R, avg + 4 * R);
        scale_cols_kernel<<<(unsigned)((S * R + 255) / 256), 256>>>(S2d, (long long)(S * R), (int)R, avg + 5 * R);
        cv = avg + 3 * R;
    } else if (f->normalize != FARNN_NORM_NONE) {
        const int ord = f->normalize == FARNN_NORM_L1_RANK ? 1 : 2;
        col_avg_norm_kernel<<<(unsigned)R, 256>>>(Vd, (int)V, (int)R, (int)R, ord, avg);
        col_avg_norm_kernel<<<(unsigned)R, 256>>>(S1d, (int)S, (int)R, (int)R, ord, avg + R);
        col_avg_norm_kernel<<<(unsigned)R, 256>>>(S2d, (int)S, (int)R, (int)R, ord, avg + 2 * R);
        norm_scales_kernel<<<(unsigned)((R + 255) / 256), 256>>>(avg, avg + 3 * R, (int)R);
        scale_cols_kernel<<<(unsigned)((S * R + 255) / 256), 256>>>(S1d, (long long)(S * R), (int)R, avg + 4 * R);
        scale_cols_kernel<<<(unsigned)((S * R + 255) / 256), 256>>>(S2d, (long long)(S * R), (int)R, avg + 5 * R);
        cv = avg + 3 * R;
    }
    fold_vgen_kernel<<<(unsigned)((V * R + 255) / 256), 256>>>(Vd, E.p, G.p, beta.p, cv, Vgen, (int)V, (int)R, (int)D, f->add_nl);
    FARNN_HIP_TRY(hipGetLastError());
    FARNN_HIP_TRY(hipDeviceSynchronize());
    farnn_decomp_ifst_desc full = *d;
    full.Vgen = Vgen; full.S1 = S1d; full.S2 = S2d;
    return decomp_ifst_create_impl(&full, device, out, 1, 1);
}

// ---- create: decomposed independent=1 ----------------------------------------------------------
extern "C" int farnn_decomp_ind1_create(const farnn_decomp_ind1_desc *d, int device, farnn_model **out) {
    if (!d || !out) return fail(FARNN_EINVAL, "null argument%s%s");
    *out = nullptr;
    if (d->V <= 0 || d->S <= 0 || d->R <= 0 || d->RO <= 0 || d->K <= 0 || !d->Vgen || !d->S1 || !d->S2 ||
        !d->W || !d->Cout || !d->S1o || !d->S2o || !d->h0 || !d->hT)
        return fail(FARNN_EINVAL, "decomp_ind1: sizes must be positive and factor pointers non-null%s%s");
    const GateSrc gates{d->farnn, d->Wss1, d->Wrs1, d->bs1, d->Wss2, d->Wrs2, d->bs2};
    if (int grc = check_gates(gates, "decomp_ind1")) return grc;
    if (d->nl < 0 || d->nl > FARNN_NL_RELUTANH) return fail(FARNN_EINVAL, "decomp_ind1: bad nl%s%s");
    int rc = select_device(device);
    if (rc) return rc;
    farnn_model *m = new (std::nothrow) farnn_model();
    if (!m) return fail(FARNN_ENOMEM, "host allocation failed%s%s");
    TunScope tun_scope(&m->tun);
    m->kind = KIND_DECOMP1; m->device = device;
    m->V = d->V; m->S = d->S; m->R = d->R; m->RO = d->RO; m->K = d->K;
    m->Kp = round_up(d->K, 4); m->Kc = round_up(d->K, 64);
    m->C = d->use_crf ? d->K - 2 : d->K;
    m->SP = round_up(d->S, 4); m->Rp = round_up(d->R, 4); m->ROp = round_up(d->RO, 4);
    m->nl = d->nl; m->semiring = d->semiring; m->threshold = d->threshold; m->o_idx = d->o_idx;
    m->use_crf = d->use_crf ? 1 : 0; m->farnn_gate = d->farnn; m->sig_k = d->sigmoid_exponent;
    const int od = d->weights_on_device;
    auto bail = [&](int code) { farnn_destroy(m); return code; };
    if (m->K > 256) return bail(fail(FARNN_ERANGE, "more than 256 label columns%s%s"));
    if (decomp1_score_lds_bytes(m->S, m->SP, m->Rp, m->ROp, m->Kc) > 160 * 1024)
        return bail(fail(FARNN_ERANGE, "decomp_ind1: S*S*4 bytes of LDS needed per token (S too large)%s%s"));
    DecompWeights &w = m->dw;
    if ((rc = upload_chain_factors(m, d->Vgen, od, d->S1, d->S2, od, d->W, od, gates, od))) return bail(rc);
    {   // no per-state output scaling in this model: o = 1; the output sum masks the transitions instead
        if ((rc = upload_ones_o(m))) return bail(rc);
        TmpDev Co, S1o, S2o, Wo;
        if ((rc = Co.init(d->Cout, (size_t)m->K * m->RO, od))) return bail(rc);
        if ((rc = S1o.init(d->S1o, (size_t)m->S * m->RO, od))) return bail(rc);
        if ((rc = S2o.init(d->S2o, (size_t)m->S * m->RO, od))) return bail(rc);
        if ((rc = Wo.init(d->Wo, (size_t)m->S * m->S, od))) return bail(rc);
        float *osum = nullptr;
        if ((rc = dev_alloc(m, (void **)&osum, (size_t)m->S * m->SP * 4))) return bail(rc);
        FARNN_HIP_TRY(hipMemset(osum, 0, (size_t)m->S * m->SP * 4));
        output_sum_kernel<<<(m->S * m->S + 255) / 256, 256>>>(Co.p, S1o.p, S2o.p, Wo.p, osum, m->K, m->S, m->SP, m->RO);
        FARNN_HIP_TRY(hipGetLastError());
        FARNN_HIP_TRY(hipDeviceSynchronize());
        w.mask = osum;
    }
    if ((rc = upload_padded(m, &m->d1_S1o, d->S1o, m->S, m->RO, m->S, m->ROp, od))) return bail(rc);
    if ((rc = upload_padded(m, &m->d1_S2o, d->S2o, m->S, m->RO, m->S, m->ROp, od))) return bail(rc);
    if ((rc = upload_transposed(m, &m->d1_CoutT, d->Cout, m->K, m->RO, m->Kc, od))) return bail(rc);
    if ((rc = dev_upload(m, &m->h0, d->h0, m->S, m->SP, od))) return bail(rc);
    if ((rc = dev_upload(m, &m->hT, d->hT, m->S, m->SP, od))) return bail(rc);
    w.h0 = m->h0; w.hT = m->hT;
    if ((rc = setup_priority(m, d->P, od))) return bail(rc);
    if ((rc = setup_crf(m, d->crf_trans, od))) return bail(rc);
    if (m->RO <= 16 * D1M_MAXNT && m->S <= 16 * D1M_MAXKQ4 && (size_t)m->V * m->S * m->SP * 4 <= ((size_t)32 << 30)) {
        // per-word bss table for the MFMA scoring kernel (unmasked: the mask only enters the recurrence),
        // materialised row-major in a scratch buffer, then re-laid-out in MFMA operand order
        const int MT = (m->S + 15) / 16, NT = (m->RO + 15) / 16, KQ4 = MT;
        float *tmp = nullptr;
        FARNN_HIP_TRY(hipMalloc((void **)&tmp, (size_t)m->V * m->S * m->SP * 4));
        dim3 grid((m->S * m->SP + 255) / 256, m->V);
        materialise_blocks_kernel<<<grid, 256>>>(m->dw.Vgen, m->dw.S1, m->dw.S2, m->dw.W, nullptr, tmp, nullptr,
                                                 m->S, m->SP, m->S, m->R, m->Rp);
        const long long total = (long long)m->V * MT * KQ4 * 256;
        rc = dev_alloc(m, (void **)&m->d1_BSSp, (size_t)total * 4);
        if (!rc) rc = dev_alloc(m, (void **)&m->d1_S1oP, (size_t)MT * NT * 256 * 4);
        if (!rc) rc = dev_alloc(m, (void **)&m->d1_S2oP, (size_t)KQ4 * NT * 256 * 4);
        if (!rc) {
            pack_s2o_operand_kernel<<<(KQ4 * NT * 256 + 255) / 256, 256>>>(m->d1_S2o, m->d1_S2oP, KQ4 * NT * 256,
                                                                          m->S, m->RO, m->ROp, NT);
            pack_bss_operand_kernel<<<(unsigned)((total + 255) / 256), 256>>>(tmp, m->d1_BSSp, total, m->S, m->SP, MT, KQ4);
            pack_s1o_operand_kernel<<<(MT * NT * 256 + 255) / 256, 256>>>(m->d1_S1o, m->d1_S1oP, MT * NT * 256,
                                                                         m->S, m->RO, m->ROp, NT);
        }
        hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
        (void)hipFree(tmp);
        if (rc) return bail(rc);
        FARNN_HIP_TRY(e1);
        FARNN_HIP_TRY(e2);
    }
    if ((rc = build_dense_blocks(m))) return bail(rc);
    *out = m;
    return FARNN_OK;
}

// ---- create: decomposed independent=0 ----------------------------------------------------------
extern "C" int farnn_decomp_fst_create(const farnn_decomp_fst_desc *d, int device, farnn_model **out) {
    if (!d || !out) return fail(FARNN_EINVAL, "null argument%s%s");
    *out = nullptr;
    if (d->V <= 0 || d->S <= 0 || d->R <= 0 || d->RW <= 0 || d->K <= 0 || !d->Vgen || !d->C || !d->S1 ||
        !d->S2 || !d->Cw || !d->S1w || !d->S2w || !d->WW || !d->h0 || !d->hT)
        return fail(FARNN_EINVAL, "decomp_fst: sizes must be positive and factor pointers non-null%s%s");
    const GateSrc gates{d->farnn, d->Wss1, d->Wrs1, d->bs1, d->Wss2, d->Wrs2, d->bs2};
    if (int grc = check_gates(gates, "decomp_fst")) return grc;
    if (d->nl < 0 || d->nl > FARNN_NL_RELUTANH) return fail(FARNN_EINVAL, "decomp_fst: bad nl%s%s");
    int rc = select_device(device);
    if (rc) return rc;
    farnn_model *m = new (std::nothrow) farnn_model();
    if (!m) return fail(FARNN_ENOMEM, "host allocation failed%s%s");
    TunScope tun_scope(&m->tun);
    m->kind = KIND_DECOMP0; m->device = device;
    m->V = d->V; m->S = d->S; m->R = d->R; m->RW = d->RW; m->K = d->K;
    m->Kp = round_up(d->K, 4); m->Kc = round_up(d->K, 64);
    m->C = d->use_crf ? d->K - 2 : d->K;
    m->SP = round_up(d->S, 4); m->Rp = round_up(d->R, 4); m->RWp = round_up(d->RW, 4);
    m->nl = d->nl; m->semiring = d->semiring; m->threshold = d->threshold; m->o_idx = d->o_idx;
    m->use_crf = d->use_crf ? 1 : 0; m->farnn_gate = d->farnn; m->sig_k = d->sigmoid_exponent;
    const int od = d->weights_on_device;
    auto bail = [&](int code) { farnn_destroy(m); return code; };
    if (m->K > 256) return bail(fail(FARNN_ERANGE, "more than 256 label columns%s%s"));
    if (m->S > 1024 || m->R > 4096 || m->RW > 4096)
        return bail(fail(FARNN_ERANGE, "decomp_fst: S<=1024, R<=4096, RW<=4096%s%s"));
    {
        // recurrence inputs: table = Vgen * sum_c C (:253), W = sum_q (sum_c Cw) S1w S2w + WW (:319-324)
        TmpDev Cd, Cwd, S1wd, S2wd, WWd;
        if ((rc = Cd.init(d->C, (size_t)m->K * m->R, od))) return bail(rc);
        if ((rc = Cwd.init(d->Cw, (size_t)m->K * m->RW, od))) return bail(rc);
        if ((rc = S1wd.init(d->S1w, (size_t)m->S * m->RW, od))) return bail(rc);
        if ((rc = S2wd.init(d->S2w, (size_t)m->S * m->RW, od))) return bail(rc);
        if ((rc = WWd.init(d->WW, (size_t)m->S * m->S, od))) return bail(rc);
        float *table = nullptr, *wsum = nullptr;
        FARNN_HIP_TRY(hipMalloc((void **)&table, (size_t)m->V * m->R * 4));
        struct Free { float *&p; ~Free() { if (p) (void)hipFree(p); } } f1{table};
        FARNN_HIP_TRY(hipMalloc((void **)&wsum, (size_t)m->S * m->S * 4));
        Free f2{wsum};
        FARNN_HIP_TRY(hipMemcpy(table, d->Vgen, (size_t)m->V * m->R * 4,
                                od ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
        const long long n = (long long)m->V * m->R;
        scale_by_colsum_kernel<<<(unsigned)((n + 255) / 256), 256>>>(table, Cd.p, m->V, m->R, m->K);
        output_sum_kernel<<<(m->S * m->S + 255) / 256, 256>>>(Cwd.p, S1wd.p, S2wd.p, WWd.p, wsum, m->K, m->S, m->S, m->RW);
        FARNN_HIP_TRY(hipGetLastError());
        FARNN_HIP_TRY(hipDeviceSynchronize());
        if ((rc = upload_chain_factors(m, table, 1, d->S1, d->S2, od, wsum, 1, gates, od))) return bail(rc);
    }
    if ((rc = upload_ones_o(m))) return bail(rc);
    if ((rc = upload_padded(m, &m->d0_Vgen, d->Vgen, m->V, m->R, m->V, m->Rp, od))) return bail(rc);
    if ((rc = upload_transposed(m, &m->d0_CT, d->C, m->K, m->R, m->Kc, od))) return bail(rc);
    if ((rc = upload_padded(m, &m->d0_S1w, d->S1w, m->S, m->RW, m->S, m->RWp, od))) return bail(rc);
    if ((rc = upload_padded(m, &m->d0_S2w, d->S2w, m->S, m->RW, m->S, m->RWp, od))) return bail(rc);
    if ((rc = upload_transposed(m, &m->d0_CwT, d->Cw, m->K, m->RW, m->Kc, od))) return bail(rc);
    if ((rc = dev_upload(m, &m->h0, d->h0, m->S, m->SP, od))) return bail(rc);
    if ((rc = dev_upload(m, &m->hT, d->hT, m->S, m->SP, od))) return bail(rc);
    m->dw.h0 = m->h0; m->dw.hT = m->hT;
    if ((rc = setup_priority(m, d->P, od))) return bail(rc);
    if ((rc = setup_crf(m, d->crf_trans, od))) return bail(rc);
    if ((rc = build_rows_pack(m))) return bail(rc);
    *out = m;
    return FARNN_OK;
}

// [x | lengths] from the pinned (device-mapped) staging buffer into device memory, by a kernel on the tagging stream: an
// SDMA copy on the same stream costs an engine hand-over before and after the recurrence (measured: 38 us of copies and
// hand-overs per 256 x 64 batch against ~6 us for this kernel; the flat predictions need no copy at all: the decode
// epilogue stores them straight into mapped host memory)
__global__ void stage_in_kernel(const int64_t *__restrict__ src, int64_t *__restrict__ dst, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) *reinterpret_cast<int4 *>(dst + i) = *reinterpret_cast<const int4 *>(src + i);
    else if (i < n) dst[i] = src[i];
}

static void host_slot_free(farnn_model::HostSlot &h) {
    if (h.x_pin) (void)hipHostFree(h.x_pin);
    if (h.flat_pin) (void)hipHostFree(h.flat_pin);
    if (h.x_dev) (void)hipFree(h.x_dev);
    h.x_pin = h.flat_pin = h.x_dev = h.flat_dev = h.x_map = nullptr;
    h.capN = h.capB = 0;
}

// One stream, in order, no copy engine: a staging kernel pulls [x | lengths] out of mapped pinned memory, the tagging launch
// stores its flat predictions straight into mapped pinned memory, one event.  (Measured alternatives, us per 256 x 64
// batch: torch tensors + pinned copies from Python 94; three event-chained streams with SDMA copies 96-120, host bound by
// the extra runtime calls; one stream with SDMA copies 88, of which 38 are copies and engine hand-overs.)
extern "C" int farnn_tag_host_submit(farnn_model *m, const int64_t *x_host, const int64_t *len_host, int32_t B, int32_t L,
                                     int32_t *ticket, int64_t *n_flat) {
    if (!m || !x_host || !len_host || !ticket) return fail(FARNN_EINVAL, "tag_host_submit: null argument%s%s");
    if (B <= 0 || L <= 0) return fail(FARNN_EINVAL, "tag_host_submit: B and L must be positive%s%s");
    FARNN_HIP_TRY(hipSetDevice(m->device));
    if (!m->hs_run) FARNN_HIP_TRY(hipStreamCreateWithFlags(&m->hs_run, hipStreamNonBlocking));
    const int slot = m->hnext;
    farnn_model::HostSlot &h = m->hslot[slot];
    // a ticket nobody waited for (the caller dropped it): the slot is free again once its batch has completed
    if (h.busy && h.ev_out && hipEventQuery(h.ev_out) == hipSuccess) h.busy = false;
    if (h.busy) return fail(FARNN_EINVAL, "tag_host_submit: FARNN_HOST_SLOTS batches already in flight (wait for the oldest ticket first)%s%s");
    const size_t N = (size_t)B * L;
    if (N > h.capN || (size_t)B > h.capB) {
        FARNN_HIP_TRY(hipStreamSynchronize(m->hs_run));
        host_slot_free(h);
        FARNN_HIP_TRY(hipHostMalloc((void **)&h.x_pin, (N + B) * 8 + 16, hipHostMallocMapped));      // [x | lengths]
        FARNN_HIP_TRY(hipHostMalloc((void **)&h.flat_pin, N * 8, hipHostMallocMapped));
        FARNN_HIP_TRY(hipMalloc((void **)&h.x_dev, (N + B) * 8 + 16));
        FARNN_HIP_TRY(hipHostGetDevicePointer((void **)&h.x_map, h.x_pin, 0));
        FARNN_HIP_TRY(hipHostGetDevicePointer((void **)&h.flat_dev, h.flat_pin, 0));                 // device view of flat_pin
        h.capN = N; h.capB = (size_t)B;
    }
    if (!h.ev_out) FARNN_HIP_TRY(hipEventCreateWithFlags(&h.ev_out, hipEventDisableTiming));
    // workspace growth frees device memory: never while older batches still run on it
    if (B > m->wsB || L > m->wsL) {
        FARNN_HIP_TRY(hipStreamSynchronize(m->hs_run));
        int rc = farnn_reserve(m, B, L);
        if (rc) return rc;
    }
    memcpy(h.x_pin, x_host, N * 8);
    memcpy(h.x_pin + N, len_host, (size_t)B * 8);
    long long total = 0;
    for (int b = 0; b < B; b++) { const long long v = len_host[b]; total += v < 0 ? 0 : (v > L ? L : v); }
    h.total = total;
    {
        const long long n = (long long)(N + B);
        stage_in_kernel<<<(unsigned)((n / 2 + 256) / 256), 256, 0, m->hs_run>>>(h.x_map, h.x_dev, n);
        FARNN_HIP_TRY(hipGetLastError());
    }
    int rc = farnn_tag(m, h.x_dev, h.x_dev + N, B, L, FARNN_MODE_LOCAL, nullptr, h.flat_dev, nullptr, m->hs_run);
    if (rc) return rc;
    FARNN_HIP_TRY(hipEventRecord(h.ev_out, m->hs_run));
    h.busy = true;
    h.gen = (h.gen + 1) & 0x7fffffu;
    m->hnext = (slot + 1) % FARNN_HOST_SLOTS;
    *ticket = (int32_t)((unsigned)slot | (h.gen << 8));
    if (n_flat) *n_flat = total;
    return FARNN_OK;
}

extern "C" int farnn_tag_host_wait(farnn_model *m, int32_t ticket, int64_t *flat_out, int64_t *n_out) {
    if (!m || ticket < 0 || (ticket & 0xff) >= FARNN_HOST_SLOTS) return fail(FARNN_EINVAL, "tag_host_wait: bad ticket%s%s");
    farnn_model::HostSlot &h = m->hslot[ticket & 0xff];
    // (a ticket names ONE submit: slot | generation << 8.  A ticket whose batch was already waited for, or whose slot was
    //  reclaimed and handed to a later submit, is refused -- it never consumes the newer batch.)
    if (!h.busy || h.gen != ((unsigned)ticket >> 8))
        return fail(FARNN_EINVAL, "tag_host_wait: no batch in flight under this ticket (stale or already waited for)%s%s");
    FARNN_HIP_TRY(hipSetDevice(m->device));
    FARNN_HIP_TRY(hipEventSynchronize(h.ev_out));
    if (flat_out && h.total > 0) memcpy(flat_out, h.flat_pin, (size_t)h.total * 8);
    if (n_out) *n_out = h.total;
    h.busy = false;
    return FARNN_OK;
}

// utils.flatten (reference utils.py:153-164) for a host int64 [B][L] array: the valid prefix of every row, batch-major.
// Host-side helper of the same boundary (the flat gold labels forward_local returns beside the predictions).
extern "C" int64_t farnn_flatten_host(const int64_t *a, const int64_t *len_host, int32_t B, int32_t L, int64_t *out) {
    int64_t n = 0;
    if (!a || !len_host || !out) return -1;
    for (int b = 0; b < B; b++) {
        long long v = len_host[b];
        v = v < 0 ? 0 : (v > L ? L : v);
        memcpy(out + n, a + (size_t)b * L, (size_t)v * 8);
        n += v;
    }
    return n;
}

extern "C" void farnn_destroy(farnn_model *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    for (auto &h : m->hslot) {
        host_slot_free(h);
        if (h.ev_out) (void)hipEventDestroy(h.ev_out);
    }
    if (m->hs_run) (void)hipStreamDestroy(m->hs_run);
    prof_fold(m);
    for (hipEvent_t e : m->prof.pool) (void)hipEventDestroy(e);
    for (void *p : m->owned) (void)hipFree(p);
    if (m->A) (void)hipFree(m->A);
    if (m->Bk) (void)hipFree(m->Bk);
    if (m->offs) (void)hipFree(m->offs);
    if (m->order) (void)hipFree(m->order);
    if (m->crf_scores) (void)hipFree(m->crf_scores);
    if (m->d1_br) (void)hipFree(m->d1_br);
    if (m->hs) (void)hipFree(m->hs);
    if (m->ev_order) (void)hipEventDestroy(m->ev_order);
    delete m;
}

// ---- introspection ---------------------------------------------------------------------------
extern "C" int farnn_abi_version(void) { return FARNN_ABI_VERSION; }
extern "C" int farnn_ab_build(void) {
#if defined(FARNN_AB)
    return 1;
#else
    return 0;
#endif
}

extern "C" int farnn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *farnn_last_error(void) { return g_err; }

extern "C" int farnn_num_columns(const farnn_model *m) { return m ? m->K : 0; }

extern "C" double farnn_algorithmic_bytes(const farnn_model *m, int64_t valid_tokens) {
    if (!m) return 0.0;
    const double S = m->S, C = m->C, R = m->R, K = m->K;
    double per_tok = 0.0, once = 0.0;
    switch (m->kind) {
        case KIND_IFST: per_tok = 2.0 * S * S * 4 + 12; break;
        case KIND_IND1: per_tok = 3.0 * S * S * 4 + 12; once = C * S * S * 4; break;
        case KIND_FST4: per_tok = (C + 2.0) * S * S * 4 + 12; break;
        case KIND_DECOMP0:
        case KIND_DECOMP1:
        case KIND_DECOMP:
            per_tok = R * 4 + 12;
            once = (2.0 * S * R + S * S + K * S) * 4;
            break;
    }
    return per_tok * (double)valid_tokens + once;
}

extern "C" double farnn_kernel_algorithmic_bytes(const farnn_model *m, int32_t which, int64_t valid_tokens) {
    if (!m) return 0.0;
    const double S = m->S, C = m->C, R = m->R, K = m->K, n = (double)valid_tokens;
    if (which == KERN_CHAIN) {
        if (m->compact_on) return (2.0 * S * m->bmNS * 8 + 8) * n;      // one bit-packed block per direction + the token id
        if (!m->dense_decomp && (m->kind == KIND_DECOMP || m->kind == KIND_DECOMP1 || m->kind == KIND_DECOMP0))
            return (R * 4 + 8) * n + (2.0 * S * R + S * S) * 4;
        // one block per direction + the token id (+ the tag when the decode is this kernel's epilogue)
        return (2.0 * S * S * 4 + 8 + (m->last_fused ? 4 : 0)) * n + (m->last_fused ? K * S * 4 : 0);
    }
    if (which == KERN_SCORE) {
        switch (m->kind) {
            case KIND_FST4: return (C * S * S * 4 + 4) * n;             // the 4-D scoring stream
            case KIND_IND1: return (S * S * 4 + 4) * n + C * S * S * 4;
            default: return 4 * n + K * S * 4;                          // tags out (+ the output matrix once)
        }
    }
    return 0.0;
}


#if defined(FARNN_PROBES)
// profiling build only (not part of include/farnn.h): the per-workgroup stamps of the last compact_tag_kernel launch under FARNN_DBG=2048
extern "C" int farnn_debug_ct_stamps(long long *out, int n_workgroups) {
    if (!out || n_workgroups < 0 || n_workgroups > farnn::CT_STAMP_MAX) return FARNN_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return FARNN_EIO;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(farnn::g_ct_stamps), sizeof(long long) * 16 * (size_t)n_workgroups) == hipSuccess ? FARNN_OK : FARNN_EIO;
}
#endif
