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
    const int len = clamp_len(p.len[b], p.L);
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
        tok[k] = clamp_tok(p.x[(long long)b * p.L + idx], w.V);
    }
    const float *hinit = dir == 0 ? w.h0 : w.hT;
    float *stash = (dir == 0 ? p.A : p.Bk) + (long long)b * (p.L + 1) * SP;
    for (int j = tid; j < SP; j += nt) { float t = j < S ? hinit[j] : 0.0f; h[j] = t; stash[j] = t; }
    // pad columns of every row this chain will write (the workspace is strided with the call's L and not re-zeroed)
    if (SP > S)
        for (int q = tid; q < nsteps * (SP - S); q += nt) stash[(long long)(1 + q / (SP - S)) * SP + S + q % (SP - S)] = 0.0f;
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

inline int launch_decomp_chain(const DecompWeights &w, const int64_t *x, const int64_t *len,
                               const int *order, float *A, float *Bk, int B, int L, int full, hipStream_t s) {
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
