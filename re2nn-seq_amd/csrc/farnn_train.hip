// libfarnn_hip.so -- the training step of the decomposed i-FST (farnn_train_*; include/farnn.h).
// Its own translation unit: the kernels of train.hip.h compile beside the tagging path.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <new>
#include <utility>
#include <vector>

#include "common.hip.h"
#include "host_util.hip.h"
#include "train.hip.h"

using namespace farnn;

// ---- training step (decomposed i-FST) ------------------------------------------
struct farnn_train_ctx {
    Tunables tun;                 // the FARNN_* switches as they stood when the context was created (host_util.hip.h)
    farnn_train_dims d;
    int device = 0;
    float *ws = nullptr;          // per-batch workspace (zeroed every step)
    float *part = nullptr;        // partial products of the parameter-gradient reductions
    size_t part_floats = 0;
    size_t ws_floats = 0;
    int wsB = 0, wsL = 0;
    float *S1T = nullptr, *S2T = nullptr, *WT = nullptr, *Osum = nullptr, *dOsum = nullptr;
    float *Wss1T = nullptr, *Wss2T = nullptr, *Wrs1T = nullptr, *Wrs2T = nullptr;   // gate transposes (farnn > 0)
    float *VgenT = nullptr, *GV = nullptr;   // [R][V] and 2 x [V][S]: the gates' input halves Vgen Wrs (farnn > 0)
    int n_cu = 256;               // compute units of the device
    float *ones = nullptr;        // [ones_n] of 1.0f: bias gradients as a product with a column of ones
    size_t ones_n = 0;
    int profiling = 0;
    double prof_ms = 0.0;
    int64_t prof_n = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    volatile int *err_host = nullptr;   // pinned, device-mapped: the kernels set bit 0 on a bad label (only then is it touched)
    int *err_dev = nullptr;             // the device's address of err_host
};

extern "C" int farnn_train_create(const farnn_train_dims *d, int device, farnn_train_ctx **out) {
    if (!d || !out) return fail(FARNN_EINVAL, "train_create: null argument%s%s");
    *out = nullptr;
    if (d->V <= 0 || d->S <= 0 || d->R <= 0 || d->K <= 0) return fail(FARNN_EINVAL, "train_create: bad dimensions%s%s");
    if (d->nl < FARNN_NL_NONE || d->nl > FARNN_NL_RELUTANH) return fail(FARNN_EINVAL, "train_create: bad nonlinearity%s%s");
    // the CRF loss kernel keeps exp(transitions) in LDS: K = 190 is the hard limit (L = 4); at L = 64 it is K = 140
    // (checked per call, farnn_decomp_ifst_train_step returns FARNN_ERANGE before enqueuing anything)
    if (d->use_crf && (d->K < 4 || train_crf_lds_bytes(d->K, 4, true) > 160 * 1024))
        return fail(FARNN_ERANGE, "train_create: CRF needs 4..190 score columns%s%s");
    if (d->farnn < 0 || d->farnn > 2) return fail(FARNN_EINVAL, "train_create: farnn must be 0, 1 or 2%s%s");
    int rc;
    if ((rc = select_device(device))) return rc;
    farnn_train_ctx *c = new farnn_train_ctx();
    c->d = *d; c->device = device;
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu;
    }
    const size_t S = d->S, R = d->R;
    float *blk = nullptr;
    if (hipMalloc((void **)&blk, (2 * S * R + S * S + 2 * S + (d->farnn ? 2 * S * S + 2 * S * R : 0)) * sizeof(float)) != hipSuccess) {
        delete c;
        return fail(FARNN_ENOMEM, "train_create: out of device memory%s%s");
    }
    c->S1T = blk; c->S2T = blk + S * R; c->WT = c->S2T + S * R; c->Osum = c->WT + S * S; c->dOsum = c->Osum + S;
    if (d->farnn) {
        const size_t Vv = d->V;
        if (hipMalloc((void **)&c->VgenT, Vv * R * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&c->GV, 6 * Vv * S * sizeof(float)) != hipSuccess) {
            if (c->VgenT) (void)hipFree(c->VgenT);
            (void)hipFree(blk);
            delete c;
            return fail(FARNN_ENOMEM, "train_create: out of device memory%s%s");
        }
    }
    if (d->farnn) { c->Wss1T = c->dOsum + S; c->Wss2T = c->Wss1T + S * S; c->Wrs1T = c->Wss2T + S * S; c->Wrs2T = c->Wrs1T + S * R; }
    if (hipHostMalloc((void **)&c->err_host, sizeof(int), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->err_dev, (void *)c->err_host, 0) != hipSuccess) {
        farnn_train_destroy(c);
        return fail(FARNN_ENOMEM, "train_create: out of memory%s%s");
    }
    *c->err_host = 0;
    *out = c;
    return FARNN_OK;
}

extern "C" void farnn_train_destroy(farnn_train_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto &e : c->pending) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (c->ws) (void)hipFree(c->ws);
    if (c->part) (void)hipFree(c->part);
    if (c->ones) (void)hipFree(c->ones);
    if (c->VgenT) (void)hipFree(c->VgenT);
    if (c->GV) (void)hipFree(c->GV);
    if (c->S1T) (void)hipFree(c->S1T);
    if (c->err_host) (void)hipHostFree((void *)c->err_host);
    delete c;
}

extern "C" int farnn_train_set_profiling(farnn_train_ctx *c, int32_t enable) {
    if (!c) return fail(FARNN_EINVAL, "train_set_profiling: null context%s%s");
    c->profiling = enable;
    return FARNN_OK;
}

extern "C" int farnn_train_time(farnn_train_ctx *c, double *total_ms, int64_t *steps) {
    if (!c || !total_ms || !steps) return fail(FARNN_EINVAL, "train_time: null argument%s%s");
    FARNN_HIP_TRY(hipSetDevice(c->device));
    FARNN_HIP_TRY(hipDeviceSynchronize());
    for (auto &e : c->pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { c->prof_ms += ms; c->prof_n++; }
        (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
    }
    c->pending.clear();
    *total_ms = c->prof_ms; *steps = c->prof_n;
    c->prof_ms = 0.0; c->prof_n = 0;
    return FARNN_OK;
}

static void atb_add(AtbJobs &jobs, const float *A, const float *Bm, float *out, long long N, int M, int J) {
    if (N <= 0) return;
    if (jobs.n >= ATB_MAX_JOBS) { jobs.total_wgs = -1; return; }       // checked by the caller: never drop a product silently
    AtbJob &j = jobs.j[jobs.n];
    j.A = A; j.B = Bm; j.out = out; j.N = N; j.M = M; j.J = J;
    j.tiles_m = (M + 63) / 64; j.tiles_j = (J + 63) / 64;
    j.nsplit = (int)((N + jobs.chunk - 1) / jobs.chunk);
    j.wg0 = jobs.total_wgs; j.out0 = jobs.total_out;
    j.part_off = jobs.n ? jobs.j[jobs.n - 1].part_off + (long long)jobs.j[jobs.n - 1].nsplit * jobs.j[jobs.n - 1].M * jobs.j[jobs.n - 1].J : 0;
    jobs.total_wgs += j.tiles_m * j.tiles_j * j.nsplit;
    jobs.total_out += M * J;
    jobs.n++;
}
static size_t atb_partial_floats(const AtbJobs &jobs) {
    if (!jobs.n) return 0;
    const AtbJob &l = jobs.j[jobs.n - 1];
    return (size_t)(l.part_off + (long long)l.nsplit * l.M * l.J);
}

extern "C" int farnn_decomp_ifst_train_step(farnn_train_ctx *c, const farnn_train_weights *w, const int64_t *x,
                                            const int64_t *lengths, const int64_t *labels, int32_t B, int32_t L,
                                            int64_t valid_tokens, const farnn_train_outputs *o, void *stream) {
    if (!c || !w || !x || !lengths || !labels || !o) return fail(FARNN_EINVAL, "train_step: null argument%s%s");
    TunScope tun_scope(&c->tun);
    if (!w->Vgen || !w->S1 || !w->S2 || !w->W || !w->C || !w->h0 || !w->hT)
        return fail(FARNN_EINVAL, "train_step: null weight%s%s");
    if (!o->loss || !o->dVgen || !o->dS1 || !o->dS2 || !o->dW || !o->dC || !o->dh0 || !o->dhT || !o->tags)
        return fail(FARNN_EINVAL, "train_step: null output%s%s");
    const bool crf = c->d.use_crf != 0;
    if (crf && (!w->crf_trans || !o->dtrans)) return fail(FARNN_EINVAL, "train_step: CRF transitions / their gradient missing%s%s");
    const int farnn = c->d.farnn;
    if (farnn >= 1 && (!w->Wss1 || !w->Wrs1 || !w->bs1 || !o->dWss1 || !o->dWrs1 || !o->dbs1))
        return fail(FARNN_EINVAL, "train_step: update-gate weights / gradients missing%s%s");
    if (farnn == 2 && (!w->Wss2 || !w->Wrs2 || !w->bs2 || !o->dWss2 || !o->dWrs2 || !o->dbs2))
        return fail(FARNN_EINVAL, "train_step: reset-gate weights / gradients missing%s%s");
    if (B <= 0 || L <= 0 || valid_tokens <= 0) return fail(FARNN_EINVAL, "train_step: B, L and valid_tokens must be positive%s%s");
    // checked before anything is enqueued: the CRF kernel keeps exp(transitions) [K][K+1] and two message tables
    // [L][K] in LDS (K = 130 at L = 64: 144 KiB; K = 140 is the limit at L = 64, K = 190 at L = 4)
    if (crf && train_crf_lds_bytes(c->d.K, L, true) > 160 * 1024)
        return fail(FARNN_ERANGE, "train_step: CRF tag set too large for this sequence length (K(K+1)*4 + 9*L*K + ... bytes of LDS must fit 160 KiB)%s%s");
    if (c->d.S > TR_VPT * TR_THREADS / TR_NSEQ || c->d.R > TR_VPT * TR_THREADS / TR_NSEQ)
        return fail(FARNN_ERANGE, "train_step: more than 512 states or rank above 512%s%s");
    {   // the chain kernels address their per-step arrays [B (L+1)][S | R] and the per-word tables [V][S | R] by 32-bit element
        // offsets against scalar base pointers (train.hip.h)
        const unsigned long long wide = (unsigned long long)(c->d.S > c->d.R ? c->d.S : c->d.R);
        if ((unsigned long long)B * (L + 1) * wide >= (1ull << 30) || (unsigned long long)c->d.V * wide >= (1ull << 30))
            return fail(FARNN_ERANGE, "train_step: B (L+1) max(S, R) and V max(S, R) must stay below 2^30 elements%s%s");
    }
    FARNN_HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (*c->err_host) {        // set by an earlier step's kernels straight in pinned host memory (no sync here)
        FARNN_HIP_TRY(hipStreamSynchronize(s));
        *c->err_host = 0;
        return fail(FARNN_EINVAL, "train_step: an earlier step saw a label outside 0..K-1 at a valid position (torch's CrossEntropyLoss raises on it); that step counted it as label 0%s%s");
    }
    const size_t S = c->d.S, R = c->d.R, K = c->d.K, V = c->d.V;
    const size_t N1 = (size_t)B * (L + 1), N0 = (size_t)B * L;
    const size_t need = N1 * (8 * S + 4 * R) + N0 * (K + S) + (crf ? N0 * K + (size_t)B * K * K : 0) +
                        (farnn ? N1 * (11 * S + 2 * R) : 0);
    if (need > c->ws_floats) {
        if (c->ws) { FARNN_HIP_TRY(hipDeviceSynchronize()); (void)hipFree(c->ws); c->ws = nullptr; c->ws_floats = 0; }
        if (hipMalloc((void **)&c->ws, need * sizeof(float)) != hipSuccess)
            return fail(FARNN_ENOMEM, "train_step: out of device memory for the workspace%s%s");
        c->ws_floats = need;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profiling && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) (void)hipEventRecord(e0, s);

    TrainParams p;
    memset(&p, 0, sizeof(p));
    p.Vgen = w->Vgen; p.S1 = w->S1; p.S2 = w->S2; p.W = w->W; p.C = w->C; p.h0 = w->h0; p.hT = w->hT; p.P = w->P;
    p.S1T = c->S1T; p.S2T = c->S2T; p.WT = c->WT; p.Osum = c->Osum;
    p.x = x; p.len = lengths; p.labels = labels; p.err = c->err_dev;
    float *q = c->ws;
    p.A = q; q += N1 * S; p.Bk = q; q += N1 * S; p.GA = q; q += N1 * S; p.GB = q; q += N1 * S;
    p.Zf = q; q += N1 * S; p.Zb = q; q += N1 * S; p.BBAR = q; q += N1 * S; p.PRE = q; q += N1 * S;
    p.D1f = q; q += N1 * R; p.D1b = q; q += N1 * R; p.Tf = q; q += N1 * R; p.Tb = q; q += N1 * R;
    p.DS = q; q += N0 * K; p.AB = q; q += N0 * S;
    if (crf) { p.SC = q; q += N0 * K; p.dtrans_part = q; q += (size_t)B * K * K; p.trans = w->crf_trans; }
    p.farnn = farnn; p.sig_k = c->d.sigmoid_exponent;
    if (farnn) {
        p.ZGf = q; q += N1 * S; p.ZGb = q; q += N1 * S; p.RGf = q; q += N1 * S; p.RGb = q; q += N1 * S;
        p.CDf = q; q += N1 * S; p.CDb = q; q += N1 * S;
        p.DAZf = q; q += N1 * S; p.DAZb = q; q += N1 * S; p.DARf = q; q += N1 * S; p.DARb = q; q += N1 * S;
        p.HBARf = q; q += N1 * S;
        p.VRf = q; q += N1 * R; p.VRb = q; q += N1 * R;
        p.Wss1 = w->Wss1; p.Wrs1 = w->Wrs1; p.bs1 = w->bs1; p.Wss2 = w->Wss2; p.Wrs2 = w->Wrs2; p.bs2 = w->bs2;
        p.Wss1T = c->Wss1T; p.Wss2T = c->Wss2T; p.Wrs1T = c->Wrs1T; p.Wrs2T = c->Wrs2T;
        if (c->ones_n < N1) {
            if (c->ones) { FARNN_HIP_TRY(hipDeviceSynchronize()); (void)hipFree(c->ones); c->ones = nullptr; c->ones_n = 0; }
            if (hipMalloc((void **)&c->ones, N1 * sizeof(float)) != hipSuccess)
                return fail(FARNN_ENOMEM, "train_step: out of device memory%s%s");
            c->ones_n = N1;
            std::vector<float> hones(N1, 1.0f);
            FARNN_HIP_TRY(hipMemcpy(c->ones, hones.data(), N1 * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    p.dVgen = o->dVgen; p.dOsum = c->dOsum; p.dh0 = o->dh0; p.dhT = o->dhT; p.loss = o->loss; p.tags = o->tags;
    p.B = B; p.L = L; p.V = (int)V; p.S = (int)S; p.R = (int)R; p.K = (int)K; p.nl = c->d.nl; p.o_idx = c->d.o_idx;
    p.threshold = c->d.threshold; p.inv_tokens = 1.0f / (float)valid_tokens;

    FARNN_HIP_TRY(hipMemsetAsync(c->ws, 0, need * sizeof(float), s));
    {
        PrepJobs pj;
        memset(&pj, 0, sizeof(pj));
        auto add = [&](int kind, const float *src, float *dst, size_t rows, size_t cols) {
            if (pj.n >= PREP_MAX_JOBS) { pj.total = -1; return; }
            PrepJob &j = pj.j[pj.n++];
            j.kind = kind; j.src = src; j.dst = dst; j.rows = (int)rows; j.cols = (int)cols; j.e0 = pj.total;
            // every job starts on a 256-thread block boundary; a transpose takes one block per 32x32 tile
            const size_t ne = kind == 1 ? ((rows + 31) / 32) * ((cols + 31) / 32) * 256
                                        : (((kind == 2 ? cols : rows * cols) + 255) / 256) * 256;
            if (pj.total < 0 || ne > (size_t)0x7fffffff - (size_t)pj.total) { pj.total = -1; return; }   // 32-bit element index
            pj.total += (int)ne;
        };
        add(0, nullptr, o->loss, 1, 1); add(0, nullptr, o->dVgen, V, R); add(0, nullptr, o->dS1, S, R);
        add(0, nullptr, o->dS2, S, R); add(0, nullptr, o->dW, S, S); add(0, nullptr, o->dC, K, S);
        add(0, nullptr, o->dh0, 1, S); add(0, nullptr, o->dhT, 1, S); add(0, nullptr, c->dOsum, 1, S);
        add(1, w->S1, c->S1T, S, R); add(1, w->S2, c->S2T, S, R); add(1, w->W, c->WT, S, S);
        add(2, w->C, c->Osum, K, S);
        if (farnn) {
            add(0, nullptr, o->dWss1, S, S); add(0, nullptr, o->dWrs1, R, S); add(0, nullptr, o->dbs1, 1, S);
            add(1, w->Wss1, c->Wss1T, S, S); add(1, w->Wrs1, c->Wrs1T, R, S);
            add(1, w->Vgen, c->VgenT, V, R); add(0, nullptr, c->GV, 4 * V, S);      // GV1 | GV2 | dGV1 | dGV2
            if (farnn == 2) {
                add(0, nullptr, o->dWss2, S, S); add(0, nullptr, o->dWrs2, R, S); add(0, nullptr, o->dbs2, 1, S);
                add(1, w->Wss2, c->Wss2T, S, S); add(1, w->Wrs2, c->Wrs2T, R, S);
            }
        }
        if (pj.total < 0) return fail(FARNN_ERANGE, "train_step: too many preparation jobs%s%s");
        train_prep_kernel<<<(pj.total + 255) / 256, 256, 0, s>>>(pj);
    }
    if (farnn) {
        // GV = Vgen Wrs as A^T B with the rank as the reduction index: A = Vgen^T [R][V], B = Wrs [R][S]
        AtbJobs gj;
        memset(&gj, 0, sizeof(gj));
        gj.chunk = 128;
        atb_add(gj, c->VgenT, w->Wrs1, c->GV, (long long)R, (int)V, (int)S);
        if (farnn == 2) atb_add(gj, c->VgenT, w->Wrs2, c->GV + V * S, (long long)R, (int)V, (int)S);
        const size_t gpf = atb_partial_floats(gj);
        if (gpf > c->part_floats) {
            if (c->part) { FARNN_HIP_TRY(hipDeviceSynchronize()); (void)hipFree(c->part); c->part = nullptr; c->part_floats = 0; }
            if (hipMalloc((void **)&c->part, gpf * sizeof(float)) != hipSuccess)
                return fail(FARNN_ENOMEM, "train_step: out of device memory for the gate-input products%s%s");
            c->part_floats = gpf;
        }
        gj.partial = c->part;
        atb_partial_kernel<<<gj.total_wgs, 256, 0, s>>>(gj);
        atb_reduce_kernel<<<(gj.total_out + 255) / 256, 256, 0, s>>>(gj);
        p.GV1 = c->GV; p.GV2 = c->GV + V * S; p.dGV1 = c->GV + 2 * V * S; p.dGV2 = c->GV + 3 * V * S;
    }

    const size_t SR = S > R ? S : R;
    const size_t SPd0 = ((S + 3) & ~(size_t)3) + 8;
    const size_t lds_lw = 8 * (SPd0 + 2 * K) * sizeof(float), lds_lc = ((K * (S + 1) + 3) & ~(size_t)3) * sizeof(float);
    const bool clds = lds_lw + lds_lc <= 150 * 1024;
    const size_t lds_l = lds_lw + (clds ? lds_lc : 0);
    const size_t nwv = TR_THREADS / 64;
    const size_t SPd = ((S + 3) & ~(size_t)3) + 8, RPd = ((R + 3) & ~(size_t)3) + 8;
    // LDS of the vectors, partial sums and token lists of a chain workgroup with ns sequences
    auto vecf = [&](size_t ns) { return (ns * SPd + ns * RPd + ns * nwv * SR + ns * nwv * S + ns * (size_t)L +
                                         (farnn ? ns * SPd + 2 * ns * nwv * S : 0)) * sizeof(float); };
    auto vecb = [&](size_t ns) { return (2 * ns * SPd + ns * RPd + 2 * ns * nwv * SR + ns * nwv * S + ns * (size_t)L +
                                         (farnn ? 2 * ns * SPd : 0)) * sizeof(float); };
    const size_t mat_f = ((2 * S * R + S * S + 3) & ~(size_t)3) * sizeof(float), mat_b = ((3 * S * R + S * S + 3) & ~(size_t)3) * sizeof(float);
    const bool ldsw_f = vecf(TR_NSEQ) + mat_f <= 160 * 1024 && !tun(TUN_TRAIN_NOLDS);
    const bool ldsw_b = vecb(TR_NSEQ) + mat_b <= 160 * 1024 && !tun(TUN_TRAIN_NOLDS);
    // sequences per workgroup: two with the matrices in LDS; four when they are read through L2 every step (that mode
    // is bound by the L2 rate, and every element read then feeds four sequences) if the batch still fills the chip
    const size_t lds_cap = 156 * 1024;
    auto pick_ns = [&](bool ldsw, size_t vec4) -> int {
        const int forced = tun(TUN_TRAIN_NSEQ);
        if (ldsw) return TR_NSEQ;
        const bool fits = TR_NSEQ_L2 * SR <= (size_t)TR_VPT * TR_THREADS && vec4 + 16 <= lds_cap;
        if (forced == 2 || !fits) return TR_NSEQ;
        // (rounds 2-3 kept the gated four-sequence kernels with two register slots per thread -- 4 R or 4 S above 512, e.g. the
        // shipped rank 250 -- away: they spilled 80-320 bytes per lane.  Round 4 retired the scratch (train.hip.h: 32-bit offsets,
        // no hoisted per-call-site addresses): at B = 1024, rank 250, farnn 2 four sequences per workgroup run 5.3 ms against 7.1)
        // measured at rank 250: with 256 sequences four per workgroup leave half the CUs idle (2.98 vs 2.63 ms per step),
        // with 1024 they win (6.5 vs 8.3 ms): four once two-sequence workgroups would outnumber the CUs two to one
        return (forced == 4 || (size_t)B >= 2 * (size_t)c->n_cu) ? TR_NSEQ_L2 : TR_NSEQ;
    };
    const int ns_f = pick_ns(ldsw_f, vecf(TR_NSEQ_L2)), ns_b = pick_ns(ldsw_b, vecb(TR_NSEQ_L2));
    const size_t vec_f = vecf(ns_f), vec_b = vecb(ns_b);
    // through-L2 kernels keep as many of their S x S matrices in LDS as fit (wildcard matrix, then the gates' Wss)
    const size_t ssb = S * S * sizeof(float);
    const size_t want_ss = farnn == 2 ? 3 : (farnn == 1 ? 2 : 1);
    p.nss_f = ldsw_f || vec_f + 16 > lds_cap ? 0 : (int)std::min(want_ss, (lds_cap - vec_f - 16) / ssb);
    p.nss_b = ldsw_b || vec_b + 16 > lds_cap ? 0 : (int)std::min(want_ss, (lds_cap - vec_b - 16) / ssb);
    if (tun(TUN_TRAIN_NOLDS) > 1) p.nss_f = p.nss_b = 0;
    const size_t lds_f = vec_f + (ldsw_f ? mat_f : 16 + p.nss_f * ssb), lds_b = vec_b + (ldsw_b ? mat_b : 16 + p.nss_b * ssb);
    int rc;
    // instantiation: weights in LDS or through L2, with or without the gate state, slots per thread, sequences per workgroup
#define FARNN_TRAIN_CHAIN3(KERN, LDSWV, G, NSV, LDSB)                                                                  \
    do {                                                                                                             \
        const dim3 cgrid((B + NSV - 1) / NSV, 2);                                                                    \
        const bool twoS = NSV * S > (size_t)TR_THREADS, twoR = NSV * R > (size_t)TR_THREADS;                        \
        if (twoS)      { if ((rc = raise_lds_limit(KERN<LDSWV, G, 2, 2, NSV>, LDSB))) return rc; KERN<LDSWV, G, 2, 2, NSV><<<cgrid, TR_THREADS, LDSB, s>>>(p); } \
        else if (twoR) { if ((rc = raise_lds_limit(KERN<LDSWV, G, 1, 2, NSV>, LDSB))) return rc; KERN<LDSWV, G, 1, 2, NSV><<<cgrid, TR_THREADS, LDSB, s>>>(p); } \
        else           { if ((rc = raise_lds_limit(KERN<LDSWV, G, 1, 1, NSV>, LDSB))) return rc; KERN<LDSWV, G, 1, 1, NSV><<<cgrid, TR_THREADS, LDSB, s>>>(p); } \
    } while (0)
#define FARNN_TRAIN_CHAIN(KERN, LDSWV, NSR, LDSB)                                                         \
    do {                                                                                                  \
        if (LDSWV || NSR == TR_NSEQ) {                                                                    \
            if (farnn) FARNN_TRAIN_CHAIN3(KERN, LDSWV, true, TR_NSEQ, LDSB);                              \
            else       FARNN_TRAIN_CHAIN3(KERN, LDSWV, false, TR_NSEQ, LDSB);                             \
        } else {                                                                                          \
            if (farnn) FARNN_TRAIN_CHAIN3(KERN, false, true, TR_NSEQ_L2, LDSB);                           \
            else       FARNN_TRAIN_CHAIN3(KERN, false, false, TR_NSEQ_L2, LDSB);                          \
        }                                                                                                 \
    } while (0)
    if (ldsw_f) FARNN_TRAIN_CHAIN(train_forward_kernel, true, ns_f, lds_f);
    else        FARNN_TRAIN_CHAIN(train_forward_kernel, false, ns_f, lds_f);
    {
        int dev = 0, ncu = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const unsigned lgrid = (unsigned)std::min<size_t>(ncu > 0 ? ncu : 256, (N0 + 7) / 8);
#define FARNN_LAUNCH_LOSS(PH)                                                                          \
        if (clds) {                                                                                   \
            if ((rc = raise_lds_limit(train_loss_kernel<true, PH>, lds_l))) return rc;                \
            train_loss_kernel<true, PH><<<lgrid, 512, lds_l, s>>>(p);                                 \
        } else {                                                                                      \
            if ((rc = raise_lds_limit(train_loss_kernel<false, PH>, lds_l))) return rc;               \
            train_loss_kernel<false, PH><<<lgrid, 512, lds_l, s>>>(p);                                \
        }
        if (!crf) {
            FARNN_LAUNCH_LOSS(0)
        } else {
            // emissions -> CRF forward-backward (loss, d loss / d emissions, transition counts, Viterbi tags) -> adjoints
            FARNN_LAUNCH_LOSS(1)
            if (train_crf_lds_bytes(K, L, false) <= 160 * 1024) {
                const size_t lds_c = train_crf_lds_bytes(K, L, false);
                if ((rc = raise_lds_limit(train_crf_kernel<false>, lds_c))) return rc;
                train_crf_kernel<false><<<B, 256, lds_c, s>>>(p);
            } else {        // large tag sets: transitions, expected counts and emissions stay in global memory
                const size_t lds_c = train_crf_lds_bytes(K, L, true);
                if ((rc = raise_lds_limit(train_crf_kernel<true>, lds_c))) return rc;
                train_crf_kernel<true><<<B, 256, lds_c, s>>>(p);
            }
            crf_reduce_kernel<<<(unsigned)((K * K + 255) / 256), 256, 0, s>>>(p.dtrans_part, o->dtrans, B, (int)(K * K));
            FARNN_LAUNCH_LOSS(2)
        }
#undef FARNN_LAUNCH_LOSS
    }
    if (ldsw_b) FARNN_TRAIN_CHAIN(train_backward_kernel, true, ns_b, lds_b);
    else        FARNN_TRAIN_CHAIN(train_backward_kernel, false, ns_b, lds_b);
#undef FARNN_TRAIN_CHAIN
#undef FARNN_TRAIN_CHAIN3
    float *dGVT = nullptr;
    if (farnn) {                                       // dGV^T ([S][V]) as the A operand of dVgen += dGV Wrs^T
        dGVT = c->GV + 4 * V * S;
        PrepJobs tj;
        memset(&tj, 0, sizeof(tj));
        for (int gsel = 0; gsel < farnn; gsel++) {
            PrepJob &j = tj.j[tj.n++];
            j.kind = 1; j.src = c->GV + (2 + gsel) * V * S; j.dst = dGVT + gsel * S * V; j.rows = (int)V; j.cols = (int)S; j.e0 = tj.total;
            tj.total += (int)(((V + 31) / 32) * ((S + 31) / 32) * 256);
        }
        train_prep_kernel<<<(tj.total + 255) / 256, 256, 0, s>>>(tj);
    }
    // parameter gradients = tall-skinny products over the per-token rows (rows of non-tokens are zero)
    AtbJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    jobs.chunk = 128;
    atb_add(jobs, p.Zf, p.Tf, o->dS2, (long long)N1, (int)S, (int)R);                 // dS2 += Zf^T (v*rr)
    if (!farnn) {
        atb_add(jobs, p.A, p.D1f + R, o->dS1, (long long)N1 - 1, (int)S, (int)R);     // dS1 += f_{t-1}^T (u*v)
        atb_add(jobs, p.A, p.Zf + S, o->dW, (long long)N1 - 1, (int)S, (int)S);       // dW  += f_{t-1}^T z
    } else {                                                                            // the chain input is hbar_t, stored per row
        atb_add(jobs, p.HBARf, p.D1f, o->dS1, (long long)N1, (int)S, (int)R);
        atb_add(jobs, p.HBARf, p.Zf, o->dW, (long long)N1, (int)S, (int)S);
    }
    atb_add(jobs, p.Zb, p.Tb, o->dS1, (long long)N1, (int)S, (int)R);                 // backward chain: roles of S1, S2 swap
    atb_add(jobs, p.BBAR, p.D1b, o->dS2, (long long)N1, (int)S, (int)R);
    atb_add(jobs, p.Zb, p.BBAR, o->dW, (long long)N1, (int)S, (int)S);                // pre_j += sum_s bbar_s W[j][s]
    atb_add(jobs, p.DS, p.AB, o->dC, (long long)N0, (int)K, (int)S);                  // dC += ds^T (alpha*beta)
    if (farnn) {
        // gates read the raw previous state (stash shifted by one row) and v_t: dWss = h_{t-1}^T da, dWrs = v^T da, dbs = 1^T da
        atb_add(jobs, p.A, p.DAZf + S, o->dWss1, (long long)N1 - 1, (int)S, (int)S);
        atb_add(jobs, p.Bk, p.DAZb + S, o->dWss1, (long long)N1 - 1, (int)S, (int)S);
        atb_add(jobs, w->Vgen, p.dGV1, o->dWrs1, (long long)V, (int)R, (int)S);             // dWrs = Vgen^T dGV
        atb_add(jobs, dGVT, c->Wrs1T, o->dVgen, (long long)S, (int)V, (int)R);              // dVgen += dGV Wrs^T
        atb_add(jobs, c->ones, p.DAZf, o->dbs1, (long long)N1, 1, (int)S);
        atb_add(jobs, c->ones, p.DAZb, o->dbs1, (long long)N1, 1, (int)S);
        if (farnn == 2) {
            atb_add(jobs, p.A, p.DARf + S, o->dWss2, (long long)N1 - 1, (int)S, (int)S);
            atb_add(jobs, p.Bk, p.DARb + S, o->dWss2, (long long)N1 - 1, (int)S, (int)S);
            atb_add(jobs, w->Vgen, p.dGV2, o->dWrs2, (long long)V, (int)R, (int)S);
            atb_add(jobs, dGVT + S * V, c->Wrs2T, o->dVgen, (long long)S, (int)V, (int)R);
            atb_add(jobs, c->ones, p.DARf, o->dbs2, (long long)N1, 1, (int)S);
            atb_add(jobs, c->ones, p.DARb, o->dbs2, (long long)N1, 1, (int)S);
        }
    }
    if (jobs.total_wgs < 0) return fail(FARNN_ERANGE, "train_step: too many gradient products for one launch%s%s");
    const size_t pf = atb_partial_floats(jobs);
    if (pf > c->part_floats) {
        if (c->part) { FARNN_HIP_TRY(hipDeviceSynchronize()); (void)hipFree(c->part); c->part = nullptr; c->part_floats = 0; }
        if (hipMalloc((void **)&c->part, pf * sizeof(float)) != hipSuccess)
            return fail(FARNN_ENOMEM, "train_step: out of device memory for the gradient partials%s%s");
        c->part_floats = pf;
    }
    jobs.partial = c->part;
    atb_partial_kernel<<<jobs.total_wgs, 256, 0, s>>>(jobs);
    atb_reduce_kernel<<<(jobs.total_out + 255) / 256, 256, 0, s>>>(jobs);
    add_row_to_all_kernel<<<(unsigned)((K * S + 255) / 256), 256, 0, s>>>(o->dC, c->dOsum, (int)K, (int)S);
    FARNN_HIP_TRY(hipGetLastError());
    if (e0 && e1) { (void)hipEventRecord(e1, s); c->pending.emplace_back(e0, e1); }
    return FARNN_OK;
}
