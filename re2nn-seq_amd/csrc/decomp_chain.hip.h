// K12/K13 -- the decomposed (low-rank) i-FST recurrence, optionally GRU-gated.
//
// Reference: FARNN_S_D_W_I_S.get_forward_score (model_decompose_single.py:138-200), driven by the
// time loop of forward_local (:236-249).  Per step, with v = Vgen[x_t] (the generalized word
// vector, model_decompose.py:222-241, precomputed as a table because weights are frozen here):
//
//   gates (farnn>=1):  z = sig(k (h Wss1 + v Wrs1 + bs1));  r = sig(k (h Wss2 + v Wrs2 + bs2))
//                      hb = (1-r) h_init + r h   (farnn==2)   |   hb = h   (farnn<=1)
//   fwd:  nx = ((hb S1) * v) S2^T + hb W ;  nx = nl(nx * o)
//   bwd:  hb = hb * o ;  nx = nl( ((hb S2) * v) S1^T + hb W^T )
//   h' = nx (farnn==0)  |  (1-z) h + z nx
//
// The factors are shared by every sequence (unlike the onehot path there is no per-token matrix),
// so this kernel is latency/VALU-bound, not HBM-bound: per token it reads R*4 bytes of Vgen.
// One workgroup per (sequence, direction); every matrix is walked row-major with the output index
// on the lanes, so all weight reads are coalesced and served by L1/L2.
#pragma once
#include <stdlib.h>
#include "common.hip.h"

namespace farnn {

struct DecompWeights {
    const float *Vgen = nullptr;                    // [V][Rp]
    const float *S1 = nullptr, *S2 = nullptr;       // [S][Rp]
    const float *S1T = nullptr, *S2T = nullptr;     // [R][SP]
    const float *W = nullptr, *WT = nullptr;        // [S][SP]
    const float *Wss1 = nullptr, *Wrs1 = nullptr, *bs1 = nullptr;   // [S][SP], [R][SP], [SP]
    const float *Wss2 = nullptr, *Wrs2 = nullptr, *bs2 = nullptr;
    const float *o = nullptr, *h0 = nullptr, *hT = nullptr;         // [SP]
    const float *mask = nullptr;    // [S][SP] independent=1: output sum multiplied into the per-step
                                    // transition matrix (model_decompose_independent.py:167-168)
    // fast-path copies, row strides SPo / Rpo chosen so that (stride/4) is odd: a lane that walks
    // one ROW with 16-byte LDS reads is then bank-conflict-free against its 15 group neighbours
    const float *fS1 = nullptr, *fS2 = nullptr;     // [S][Rpo]
    const float *fS1T = nullptr, *fS2T = nullptr;   // [R][SPo]
    const float *fW = nullptr, *fWT = nullptr;      // [S][SPo]
    int SPo = 0, Rpo = 0;
    int V = 0, S = 0, SP = 0, R = 0, Rp = 0;
    int farnn = 0, nl = 0, semiring = 0;
    float sig_k = 1.0f;
};

struct DecompParams {
    DecompWeights w;
    const int64_t *x, *len;
    float *A, *Bk;
    int B, L, full;
};

constexpr int DECOMP_THREADS = 256;

__global__ void __launch_bounds__(DECOMP_THREADS)
decomp_chain_kernel(const DecompParams p) {
    extern __shared__ __align__(16) float smem[];
    const DecompWeights &w = p.w;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int item = blockIdx.x, b = item >> 1, dir = item & 1;
    const int len = (int)p.len[b];
    const int nsteps = p.full ? p.L : len;
    const int S = w.S, SP = w.SP, R = w.R, Rp = w.Rp;

    const int Lr = (p.L + 3) & ~3;
    int *tok = reinterpret_cast<int *>(smem);     // [Lr]
    float *h = smem + Lr;                         // [SP] current state
    float *hb = h + SP;                           // [SP] gated / pre-scaled state fed to the factors
    float *z = hb + SP;                           // [SP] update gate
    float *v = z + SP;                            // [Rp] word vector
    float *rr = v + Rp;                           // [Rp] (hb . S1) * v

    for (int k = tid; k < nsteps; k += nt) {
        int idx = (dir == 0) ? k : (k < len ? len - 1 - k : k);
        tok[k] = (int)p.x[(long long)b * p.L + idx];
    }
    const float *hinit = dir == 0 ? w.h0 : w.hT;
    float *stash = (dir == 0 ? p.A : p.Bk) + (long long)b * (p.L + 1) * SP;
    for (int j = tid; j < SP; j += nt) { float t = j < S ? hinit[j] : 0.0f; h[j] = t; stash[j] = t; }
    __syncthreads();

    const float *Sa = dir == 0 ? w.S1 : w.S2;        // hb . Sa            [S][Rp]
    const float *SbT = dir == 0 ? w.S2T : w.S1T;     // (.) . Sb^T         [R][SP]
    const float *Wd = dir == 0 ? w.W : w.WT;         // hb . W  or hb . W^T [S][SP]
    const bool maxsr = w.semiring == FARNN_SEMIRING_MAX;
    const bool materialise = maxsr || w.mask != nullptr;     // build Tr[s][j] from the factors per step

    for (int t = 0; t < nsteps; t++) {
        const float *vg = w.Vgen + (long long)tok[t] * Rp;
        for (int r = tid; r < Rp; r += nt) v[r] = vg[r];
        __syncthreads();
        // ---- gates (:143-154) ----------------------------------------------------------------
        for (int j = tid; j < S; j += nt) {
            float hbj = h[j];
            if (w.farnn >= 1) {
                float a1 = 0.0f, a2 = 0.0f;
                for (int s = 0; s < S; s++) {
                    const float hs = h[s];
                    a1 = fmaf(hs, w.Wss1[(long long)s * SP + j], a1);
                    if (w.farnn == 2) a2 = fmaf(hs, w.Wss2[(long long)s * SP + j], a2);
                }
                float c1 = 0.0f, c2 = 0.0f;
                for (int r = 0; r < R; r++) {
                    const float vr = v[r];
                    c1 = fmaf(vr, w.Wrs1[(long long)r * SP + j], c1);
                    if (w.farnn == 2) c2 = fmaf(vr, w.Wrs2[(long long)r * SP + j], c2);
                }
                z[j] = 1.0f / (1.0f + expf(-((a1 + c1 + w.bs1[j]) * w.sig_k)));
                if (w.farnn == 2) {
                    float rt = 1.0f / (1.0f + expf(-((a2 + c2 + w.bs2[j]) * w.sig_k)));
                    hbj = (1.0f - rt) * hinit[j] + rt * hbj;
                }
            }
            if (dir == 1) hbj *= w.o[j];                                    // :156-157
            hb[j] = hbj;
        }
        __syncthreads();
        float *srow = stash + (long long)(t + 1) * SP;
        if (!materialise) {
            // ---- rr = (hb . Sa) * v  (:169-170 / :174-175) ------------------------------------
            for (int r = tid; r < R; r += nt) {
                float a = 0.0f;
                for (int s = 0; s < S; s++) a = fmaf(hb[s], Sa[(long long)s * Rp + r], a);
                rr[r] = a * v[r];
            }
            __syncthreads();
            // ---- nx = rr . Sb^T + hb . W  (:171-173 / :176-178) --------------------------------
            for (int j = tid; j < S; j += nt) {
                float lang = 0.0f, wild = 0.0f;
                for (int r = 0; r < R; r++) lang = fmaf(rr[r], SbT[(long long)r * SP + j], lang);
                for (int s = 0; s < S; s++) wild = fmaf(hb[s], Wd[(long long)s * SP + j], wild);
                float nx = lang + wild;
                if (dir == 0) nx *= w.o[j];                                  // :180-181
                nx = apply_nl(nx, w.nl);
                float hn = (w.farnn == 0) ? nx : (1.0f - z[j]) * h[j] + z[j] * nx;   // :193-196
                srow[j] = hn;
                h[j] = hn;                 // only thread j reads h[j] in this phase
            }
        } else {
            // ---- materialised transition (max-times semiring, model_decompose_single.py:159-166, and
            // the independent=1 model, model_decompose_independent.py:165-173):
            //     Tr[s][j] = (sum_r v_r S1[s][r] S2[j][r] + W[s][j]) * mask[s][j]
            for (int j = tid; j < S; j += nt) {
                float best = maxsr ? -INFINITY : 0.0f;
                for (int s = 0; s < S; s++) {
                    // forward uses Tr[s][j]; backward uses Tr^T, i.e. Tr[j][s]
                    const int fr = dir == 0 ? s : j, to = dir == 0 ? j : s;
                    float tr = 0.0f;
                    for (int r = 0; r < R; r++)
                        tr = fmaf(v[r] * w.S1[(long long)fr * Rp + r], w.S2[(long long)to * Rp + r], tr);
                    tr += w.W[(long long)fr * SP + to];
                    if (w.mask) tr *= w.mask[(long long)fr * SP + to];
                    best = maxsr ? fmaxf(best, hb[s] * tr) : fmaf(hb[s], tr, best);
                }
                float nx = best;
                if (dir == 0) nx *= w.o[j];
                nx = apply_nl(nx, w.nl);
                float zj = (w.farnn == 0) ? 0.0f : z[j];
                float hn = (w.farnn == 0) ? nx : (1.0f - zj) * h[j] + zj * nx;
                srow[j] = hn;
                h[j] = hn;
            }
        }
        __syncthreads();
    }
}

// ---- fast path: farnn == 0, sum semiring, factors LDS-resident -------------------------------
// The three matrices a direction needs are shared by every sequence, so a workgroup pulls them
// into LDS once (LDS-DMA) and then runs TWO sequences of the same direction, one wavefront each,
// with no barrier in the time loop.  Every product is a ROW dot product owned by one lane:
//     fwd:  rr[r] = v[r] * <S1T[r,:], hb>      nx[j] = <S2[j,:], rr> + <WT[j,:], hb>
//     bwd:  rr[r] = v[r] * <S2T[r,:], hb>      nx[j] = <S1[j,:], rr> + <W[j,:],  hb>
// so a lane walks its row with 16-byte LDS reads (odd row stride in 16-byte units: conflict-free)
// against 16-byte LDS broadcasts of the vector, four independent accumulators per row -- the
// per-step cost is LDS issue, not L2 latency (the generic kernel above re-reads ~85 KB of factors
// from L2 per step and workgroup).
struct DecompFastParams {
    DecompWeights w;
    const int64_t *x, *len;
    const int *order;
    float *A, *Bk;
    int B, L, full;
};

constexpr int DECOMP_FAST_WAVES = 2;

// dot(row[0..n), vec[0..n)) with n a multiple of 4, vec read as LDS broadcasts
__device__ __forceinline__ float row_dot(const float *row, const float *vec, int n) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    // 8 x 16 bytes of each operand in flight per wait: a lone wavefront has no other wave to hide
    // the LDS latency behind, so the loads of a whole block are issued before the first FMA
    for (; k + 32 <= n; k += 32) {
        float4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { a[u] = ld4(row + k + 4 * u); b[u] = ld4(vec + k + 4 * u); }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc.x = fmaf(a[u].x, b[u].x, acc.x); acc.y = fmaf(a[u].y, b[u].y, acc.y);
            acc.z = fmaf(a[u].z, b[u].z, acc.z); acc.w = fmaf(a[u].w, b[u].w, acc.w);
        }
    }
    for (; k < n; k += 4) {
        const float4 a = ld4(row + k), b = ld4(vec + k);
        acc.x = fmaf(a.x, b.x, acc.x); acc.y = fmaf(a.y, b.y, acc.y);
        acc.z = fmaf(a.z, b.z, acc.z); acc.w = fmaf(a.w, b.w, acc.w);
    }
    return (acc.x + acc.y) + (acc.z + acc.w);
}

__global__ void __launch_bounds__(DECOMP_FAST_WAVES * 64)
decomp_fast_kernel(const DecompFastParams p) {
    extern __shared__ __align__(16) float smem[];
    const DecompWeights &w = p.w;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.x & 1, pair = blockIdx.x >> 1;
    const int S = w.S, SP = w.SP, R = w.R, Rp = w.Rp, SPo = w.SPo, Rpo = w.Rpo;
    const int Lr = (p.L + 3) & ~3;

    // ---- LDS carve: shared factors (whole 1 KiB DMA pieces), then per-wavefront state ---------
    const int nA = (R * SPo * 4 + 1023) / 1024, nB = (S * Rpo * 4 + 1023) / 1024, nW = (S * SPo * 4 + 1023) / 1024;
    float *Al = smem;                                 // [R][SPo]  S1T (fwd) / S2T (bwd)
    float *Bl = Al + nA * 256;                        // [S][Rpo]  S2  (fwd) / S1  (bwd)
    float *Wl = Bl + nB * 256;                        // [S][SPo]  WT  (fwd) / W   (bwd)
    float *ol = Wl + nW * 256;                        // [SP]
    float *per = ol + SP + (size_t)wv * (Lr + 2 * SP + Rp);
    int *tok = reinterpret_cast<int *>(per);          // [Lr]
    float *hb = per + Lr;                             // [2][SP] state as fed to the factors (bwd: * o)
    float *rr = hb + 2 * SP;                          // [Rp]

    {   // factors -> LDS, pieces round-robin over the wavefronts
        const char *srcs[3] = {reinterpret_cast<const char *>(dir == 0 ? w.fS1T : w.fS2T),
                               reinterpret_cast<const char *>(dir == 0 ? w.fS2 : w.fS1),
                               reinterpret_cast<const char *>(dir == 0 ? w.fWT : w.fW)};
        const float *dsts[3] = {Al, Bl, Wl};
        const int cnts[3] = {nA, nB, nW};
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dsts[m]);
            for (int k = wv; k < cnts[m]; k += DECOMP_FAST_WAVES)
                lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, srcs[m], lds0 + (unsigned)k * 1024u);
        }
    }
    for (int j = tid; j < SP; j += blockDim.x) ol[j] = j < S ? w.o[j] : 0.0f;

    const int slot = pair * DECOMP_FAST_WAVES + wv;
    const bool have = slot < p.B;
    const int b = have ? (p.order ? p.order[slot] : slot) : 0;
    const int len = have ? (int)p.len[b] : 0;
    const int nsteps = have ? (p.full ? p.L : len) : 0;
    for (int k = lane; k < nsteps; k += WAVE) {
        const int idx = (dir == 0) ? k : (k < len ? len - 1 - k : k);
        tok[k] = (int)p.x[(long long)b * p.L + idx];
    }
    const float *hinit = dir == 0 ? w.h0 : w.hT;
    float *stash = (dir == 0 ? p.A : p.Bk) + (long long)b * (p.L + 1) * SP;
    for (int j = lane; j < SP; j += WAVE) {
        const float t0 = j < S ? hinit[j] : 0.0f;
        if (have) stash[j] = t0;
        hb[j] = (dir == 1 && j < S) ? t0 * w.o[j] : t0;              // (:156-157) backward input pre-scaled
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wavefront's DMA pieces landed
    __syncthreads();
    if (!have) return;

    int cur = 0;
    for (int t = 0; t < nsteps; t++) {
        const float *hc = hb + cur * SP;
        float *hn = hb + (cur ^ 1) * SP;
        const float *vg = w.Vgen + (long long)__builtin_amdgcn_readfirstlane(tok[t]) * Rp;
        // ---- rr = (hb . Sa) * v   (:169-170 / :174-175); lane owns r -------------------------
        for (int r0 = lane; r0 < Rp; r0 += WAVE) {
            const bool ok = r0 < R;
            const float vv = ok ? vg[r0] : 0.0f;
            const float d = row_dot(Al + (ok ? r0 : 0) * SPo, hc, SP);   // pad entries of hb are 0
            rr[r0] = ok ? d * vv : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- nx = rr . Sb^T + hb . W   (:171-173 / :176-178); lane owns output j --------------
        float *srow = stash + (long long)(t + 1) * SP;
        for (int j0 = lane; j0 < SP; j0 += WAVE) {
            const int j = j0 < S ? j0 : S - 1;
            const float lang = row_dot(Bl + j * Rpo, rr, Rp);           // pad entries of rr are 0
            const float wild = row_dot(Wl + j * SPo, hc, SP);
            float nx = lang + wild;
            const float ov = ol[j];
            if (dir == 0) nx *= ov;                                       // :180-181
            nx = apply_nl(nx, w.nl);
            if (j0 < S) {
                srow[j0] = nx;
                hn[j0] = dir == 0 ? nx : nx * ov;
            } else {
                hn[j0] = 0.0f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        cur ^= 1;
    }
}

inline size_t decomp_fast_lds_bytes(const DecompWeights &w, int L) {
    const int Lr = (L + 3) & ~3;
    size_t pieces = ((size_t)w.R * w.SPo * 4 + 1023) / 1024 + ((size_t)w.S * w.Rpo * 4 + 1023) / 1024 +
                    ((size_t)w.S * w.SPo * 4 + 1023) / 1024;
    return pieces * 1024 + ((size_t)w.SP + (size_t)DECOMP_FAST_WAVES * (Lr + 2 * w.SP + w.Rp)) * 4;
}

inline int launch_decomp_chain(const DecompWeights &w, const int64_t *x, const int64_t *len,
                               const int *order, float *A, float *Bk, int B, int L, int full, hipStream_t s) {
    const bool fast_ok = w.farnn == 0 && w.semiring == FARNN_SEMIRING_SUM && !w.mask &&
                         decomp_fast_lds_bytes(w, L) <= 150 * 1024 && !getenv("FARNN_DECOMP_GENERIC");
    if (fast_ok) {
        DecompFastParams p;
        p.w = w; p.x = x; p.len = len; p.order = order; p.A = A; p.Bk = Bk; p.B = B; p.L = L; p.full = full;
        const size_t lds = decomp_fast_lds_bytes(w, L);
        if (lds > 48 * 1024)
            FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(decomp_fast_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int pairs = (B + DECOMP_FAST_WAVES - 1) / DECOMP_FAST_WAVES;
        decomp_fast_kernel<<<dim3(2 * pairs), dim3(DECOMP_FAST_WAVES * 64), lds, s>>>(p);
        FARNN_HIP_TRY(hipGetLastError());
        return FARNN_OK;
    }
    DecompParams p;
    p.w = w; p.x = x; p.len = len; p.A = A; p.Bk = Bk; p.B = B; p.L = L; p.full = full;
    const int Lr = (L + 3) & ~3;
    size_t lds = ((size_t)Lr + 3 * (size_t)w.SP + 2 * (size_t)w.Rp) * sizeof(float);
    if (lds > 160 * 1024) return fail(FARNN_ERANGE, "decomposed chain: LDS budget exceeded%s%s");
    if (lds > 48 * 1024)
        FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(decomp_chain_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    decomp_chain_kernel<<<dim3(2 * B), dim3(DECOMP_THREADS), lds, s>>>(p);
    FARNN_HIP_TRY(hipGetLastError());
    return FARNN_OK;
}

}  // namespace farnn
