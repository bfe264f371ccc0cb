// K12r "register" kernel -- the decomposed recurrence (FARNN_S_D_W_I_S.get_forward_score,
// model_decompose_single.py:138-200; farnn = 0, sum semiring) with the WEIGHTS IN REGISTERS.
//
//   fwd:  rr = v_t * (S1^T . h) ;  h' = nl( (S2 . rr + W^T . h) * o )
//   bwd:  hb = h * o ;  rr = v_t * (S2^T . hb) ;  h' = nl( S1 . rr + W . hb )
//
// Why: the factors are shared by every sequence, the whole batch is 0.74 GFLOP (4.7 us at the f32 peak), and a step is
// two dependent matrix-vector products of ~21 k multiply-adds.  K12 (decomp_rows_kernel) keeps the packed rows of one
// direction in LDS (105 KiB at rank 50): ONE chain per CU, 512 chains on 256 CUs in two rounds, and per step the eight
// wavefronts re-read 85 KB of weights through the LDS pipe: ~1 us per step, 125 us per batch, 3.8 % of the f32 rate.
// Measured dead end on the way here (profiles/r02_decomp_wave_probe.txt): one wavefront per chain with shared LDS rows and
// no barrier at all -- a lone wavefront issues a 16-byte LDS read every ~16 cycles, a quarter of the LDS rate, and its
// 120 row reads per step alone took longer (2.2 us per step) than K12's whole step.
// Here a chain is FOUR wavefronts (one per SIMD) and every lane keeps its slice of the packed rows in registers for the
// whole sequence: 21.2 k weights / 256 lanes = 83 registers (112 with the layout's padding).  A step reads only the
// input vectors from LDS (~2 KB instead of 85 KB), does 56 packed FMAs per lane, three quad reductions on the DPP network
// and two workgroup barriers of four wavefronts.  No LDS-resident weights means the workgroups are small: two chains per
// CU (8 wavefronts, 2 per SIMD at ~200 VGPRs), so the 512 chains of a 256-sequence batch run in ONE round.
//
// Layout: the packed rows of K12 (P2[dir] = Sa^T rows, P3[dir] = [Sb | W(^T)] rows, o folded in, row stride ld2 / ld3),
// read once from global memory (L2) at set-up.  Four adjacent lanes share a row (lane k of the quad owns the 16-byte
// pieces k and k+4 of every 32-column chunk), a wavefront covers 16 rows, the workgroup 64 rows per pass: P2 (R <= 64
// rows) is one pass, P3 (S rows) NP3 passes.  Bound: the serial step chain (issue + LDS latency + two barriers per step).
#pragma once
#include "common.hip.h"
#include "decomp_rows.hip.h"

namespace farnn {

constexpr int DG_WAVES = 4;                   // wavefronts per chain
constexpr int DG_THREADS = DG_WAVES * 64;
constexpr int DG_ROWS = DG_THREADS / 4;       // rows per pass (64)

struct DecompRegsParams {
    const float *P2[2];           // [R][ld2]   per direction
    const float *P3[2];           // [S][ld3]   per direction
    int ld2, ld3;
    const float *Vgen;            // [V][Rp]
    const float *h0, *hT;
    const int64_t *x, *len;
    const int *order;             // folded launch order (batch_prep) or nullptr
    int sort;                     // 1: no order array, the workgroup selects its sequence by length rank itself
    float *A, *Bk;
    int B, L, S, SP, R, Rp, nl, full, V;
    int dbg;                      // FARNN_DBG & 4096: workgroup 0 prints its per-phase cycle counts (diagnostic)
};

// tanh on the hardware exponential and reciprocal: |error| ~2e-7 against the 1e-4 parity bar
__device__ __forceinline__ float dg_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));          // in (0, 1]
    return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}
__device__ __forceinline__ float dg_nl(float x, int nl) {
    switch (nl) {
        case FARNN_NL_RELU: return fmaxf(x, 0.0f);
        case FARNN_NL_TANH: return dg_tanh(x);
        case FARNN_NL_RELUTANH: return dg_tanh(fmaxf(x, 0.0f));
        default: return x;
    }
}

// NCH2 / NCH3: 32-column chunks of the two input vectors (h: S columns; [rr | h]: Rp + S columns); NP3: passes of P3;
// CS: leading chunks of [rr | h] that hold rr entries (they can only be read behind the step's first barrier)
// (Tried in r02 and dropped: both chains of a sequence in one eight-wavefront workgroup with the scores + decode as its
// epilogue -- one launch per step, but the two chains in step through shared barriers ran 63.5 us against 56.9, and the
// two score tiles of the longest sequence, serial on one CU behind it, cost more than the separate launch: 76.5 us per
// step against 69.1.)
template <int NCH2, int NCH3, int NP3, int CS>
__global__ void __launch_bounds__(DG_THREADS, 2)
decomp_regs_kernel(const DecompRegsParams p) {
    extern __shared__ __align__(16) float smem[];
    const int wtid = threadIdx.x;
    const int tid = wtid, lane = tid & 63;
    const int dir = (int)blockIdx.x & 1;
    const int seq = (int)blockIdx.x >> 1;
    const int S = p.S, SP = p.SP, R = p.R, Rp = p.Rp;
    constexpr int c2p = NCH2 * DR_CHUNK, c3p = NCH3 * DR_CHUNK;
    const int Lr = (p.L + 3) & ~3;

    // ---- LDS: the input vectors (ping-pong: the leaders write the next state while slower wavefronts still read) -----
    float *H = smem;                                          // [2][c2p]  h: input of P2
    float *X3 = H + 2 * c2p;                                  // [2][c3p]  rr | h: input of P3
    int *tok = reinterpret_cast<int *>(X3 + 2 * c3p);         // [Lr]
    int *scratch = tok + Lr;                                  // select_by_length_rank: L + 17 ints

    int b = p.order ? p.order[seq] : seq;
    if (p.sort) b = select_by_length_rank(p.len, p.B, p.L, folded_rank(seq, p.B), scratch, tid, DG_THREADS);
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;

    // ---- this lane's slice of the packed rows: registers for the whole sequence -------------------------------------------
    const int k = lane & 3, rslot = tid >> 2;                 // row slot 0..63 of a pass
    const bool own2 = rslot < R;
    v4f w2[2 * NCH2], w3[NP3][2 * NCH3];
    {
        const float *src = p.P2[dir] + (long long)(own2 ? rslot : R - 1) * p.ld2 + k * 4;
#pragma unroll
        for (int c = 0; c < NCH2; c++) {
            w2[2 * c] = *reinterpret_cast<const v4f *>(src + c * DR_CHUNK);
            w2[2 * c + 1] = *reinterpret_cast<const v4f *>(src + c * DR_CHUNK + 16);
        }
#pragma unroll
        for (int i = 0; i < NP3; i++) {
            const int row = i * DG_ROWS + rslot;
            const float *s3 = p.P3[dir] + (long long)(row < S ? row : S - 1) * p.ld3 + k * 4;
#pragma unroll
            for (int c = 0; c < NCH3; c++) {
                w3[i][2 * c] = *reinterpret_cast<const v4f *>(s3 + c * DR_CHUNK);
                w3[i][2 * c + 1] = *reinterpret_cast<const v4f *>(s3 + c * DR_CHUNK + 16);
            }
        }
    }
    const float *hinit = dir == 0 ? p.h0 : p.hT;
    float *stash = (dir == 0 ? p.A : p.Bk) + (long long)b * (p.L + 1) * SP;
    for (int q = tid; q < nsteps; q += DG_THREADS) {
        const int idx = (dir == 0) ? q : (q < len ? len - 1 - q : q);
        tok[q] = clamp_tok(p.x[(long long)b * p.L + idx], p.V);
    }
    for (int j = tid; j < 2 * c2p; j += DG_THREADS) H[j] = (j < S) ? hinit[j] : 0.0f;           // buffer 0 = h_0
    for (int j = tid; j < 2 * c3p; j += DG_THREADS) X3[j] = (j >= Rp && j < Rp + S) ? hinit[j - Rp] : 0.0f;
    for (int j = tid; j < SP; j += DG_THREADS) stash[j] = j < S ? hinit[j] : 0.0f;                 // state 0
    __syncthreads();
    if (nsteps <= 0) return;

    // word-vector entry of this lane's P2 row, TWO steps ahead and with no branch around the load (idle lanes read entry 0):
    // vmcnt retires in order, so a value is only ever waited for when a younger load and the stash stores are already in
    // flight behind it -- one step ahead the compiler's wait for it also drained the load just issued (a full L2 round trip
    // on every step's critical path)
    const int vcol = own2 ? rslot : 0;
    auto v_addr = [&](int tk) -> const float * { return p.Vgen + (long long)tk * Rp + vcol; };
    float v0 = *v_addr(tok[0]), v1 = *v_addr(tok[nsteps > 1 ? 1 : 0]);
    int tk2 = tok[nsteps > 2 ? 2 : nsteps - 1];
    const int nl_mode = p.nl;
    static_assert(NP3 <= 4, "one lane of the quad per row pass");
    static_assert(CS >= 1 && CS <= NCH3, "rr chunks");
    long long cyc[4] = {0, 0, 0, 0};
    for (int t = 0; t < nsteps; t++) {
        long long c0 = FARNN_PROBE_ON(p.dbg & 4096) ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const int cur = t & 1, nxt = cur ^ 1;
        const float v2 = *v_addr(tk2);                        // step t+2's entry: in flight for two steps
        tk2 = tok[t + 3 < nsteps ? t + 3 : nsteps - 1];      // (consumed at the next step's start)
        lds_cfloat *Hc = (lds_cfloat *)(H + cur * c2p) + k * 4;
        float *X3c = X3 + cur * c3p, *X3n = X3 + nxt * c3p, *Hn = H + nxt * c2p;
        // ---- phase A (needs h only): P2, rr[r] = v[r] * <Sa[:, r], h>, and the part of P3 that does not depend on rr -- the
        // chunks of [rr | h] that hold state entries only (W(^T) . h and the tail of nothing else) --------------------------
        v2f pl3[NP3], ph3[NP3];
        {
            v2f tl = v2f{0.f, 0.f}, th = v2f{0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NCH2; c++) {
                const v4f x0 = *(lds_cv4f *)(Hc + c * DR_CHUNK), x1 = *(lds_cv4f *)(Hc + c * DR_CHUNK + 16);
                tl = __builtin_elementwise_fma(v2f{w2[2 * c].x, w2[2 * c].y}, v2f{x0.x, x0.y}, tl);
                th = __builtin_elementwise_fma(v2f{w2[2 * c].z, w2[2 * c].w}, v2f{x0.z, x0.w}, th);
                tl = __builtin_elementwise_fma(v2f{w2[2 * c + 1].x, w2[2 * c + 1].y}, v2f{x1.x, x1.y}, tl);
                th = __builtin_elementwise_fma(v2f{w2[2 * c + 1].z, w2[2 * c + 1].w}, v2f{x1.z, x1.w}, th);
            }
            const v2f tt = tl + th;
            const float acc = quad_sum(tt.x + tt.y);
            if (k == 0 && own2) X3c[rslot] = acc * v0;
            lds_cfloat *xq = (lds_cfloat *)X3c + k * 4;
#pragma unroll
            for (int i = 0; i < NP3; i++) { pl3[i] = v2f{0.f, 0.f}; ph3[i] = v2f{0.f, 0.f}; }
#pragma unroll
            for (int c = 0; c < NCH3; c++) {
                if (c < CS) continue;                         // the chunks that hold rr entries wait for the barrier
                const v4f x0 = *(lds_cv4f *)(xq + c * DR_CHUNK), x1 = *(lds_cv4f *)(xq + c * DR_CHUNK + 16);
#pragma unroll
                for (int i = 0; i < NP3; i++) {
                    pl3[i] = __builtin_elementwise_fma(v2f{w3[i][2 * c].x, w3[i][2 * c].y}, v2f{x0.x, x0.y}, pl3[i]);
                    ph3[i] = __builtin_elementwise_fma(v2f{w3[i][2 * c].z, w3[i][2 * c].w}, v2f{x0.z, x0.w}, ph3[i]);
                    pl3[i] = __builtin_elementwise_fma(v2f{w3[i][2 * c + 1].x, w3[i][2 * c + 1].y}, v2f{x1.x, x1.y}, pl3[i]);
                    ph3[i] = __builtin_elementwise_fma(v2f{w3[i][2 * c + 1].z, w3[i][2 * c + 1].w}, v2f{x1.z, x1.w}, ph3[i]);
                }
            }
        }
        if (FARNN_PROBE_ON(p.dbg & 4096)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long c1 = __builtin_amdgcn_s_memtime(); cyc[0] += c1 - c0; c0 = c1; }
        wg_barrier_lds();
        if (FARNN_PROBE_ON(p.dbg & 4096)) { const long long c1 = __builtin_amdgcn_s_memtime(); cyc[1] += c1 - c0; c0 = c1; }
        // ---- phase B: the rr chunks of P3, h'[j] = nl(<[Sb[j, :] | Wd[:, j]], [rr | h]>), into the other buffers and the stash
        {
            lds_cfloat *xq = (lds_cfloat *)X3c + k * 4;
#pragma unroll
            for (int c = 0; c < NCH3; c++) {
                if (c >= CS) continue;
                const v4f x0 = *(lds_cv4f *)(xq + c * DR_CHUNK), x1 = *(lds_cv4f *)(xq + c * DR_CHUNK + 16);
#pragma unroll
                for (int i = 0; i < NP3; i++) {
                    pl3[i] = __builtin_elementwise_fma(v2f{w3[i][2 * c].x, w3[i][2 * c].y}, v2f{x0.x, x0.y}, pl3[i]);
                    ph3[i] = __builtin_elementwise_fma(v2f{w3[i][2 * c].z, w3[i][2 * c].w}, v2f{x0.z, x0.w}, ph3[i]);
                    pl3[i] = __builtin_elementwise_fma(v2f{w3[i][2 * c + 1].x, w3[i][2 * c + 1].y}, v2f{x1.x, x1.y}, pl3[i]);
                    ph3[i] = __builtin_elementwise_fma(v2f{w3[i][2 * c + 1].z, w3[i][2 * c + 1].w}, v2f{x1.z, x1.w}, ph3[i]);
                }
            }
            // every lane of a quad gets the row sums; lane k finishes row pass k (k < NP3), so the non-linearity of all the
            // passes runs once, on different lanes
            float mine = 0.0f;
#pragma unroll
            for (int i = 0; i < NP3; i++) {
                const v2f tt = pl3[i] + ph3[i];
                const float acc = quad_sum(tt.x + tt.y);
                mine = (k == i) ? acc : mine;
            }
            const int row = k * DG_ROWS + rslot;
            if (k < NP3 && row < SP) {
                float hn = 0.0f;                              // pad columns of the stash stay zero
                if (row < S) {
                    hn = dg_nl(mine, nl_mode);
                    Hn[row] = hn;
                    X3n[Rp + row] = hn;
                }
                stash[(long long)(t + 1) * SP + row] = hn;
            }
        }
        v0 = v1; v1 = v2;
        if (FARNN_PROBE_ON(p.dbg & 4096)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long c1 = __builtin_amdgcn_s_memtime(); cyc[2] += c1 - c0; c0 = c1; }
        wg_barrier_lds();
        if (FARNN_PROBE_ON(p.dbg & 4096)) { const long long c1 = __builtin_amdgcn_s_memtime(); cyc[3] += c1 - c0; }
    }
    if (FARNN_PROBE_ON(p.dbg & 4096) && blockIdx.x < 2 && (tid & 63) == 0)
        printf("regs kernel wg %d wave %d: %d steps, cycles per step: A %lld  barrier %lld  B %lld  barrier %lld\n", (int)blockIdx.x, wtid >> 6,
               nsteps, cyc[0] / nsteps, cyc[1] / nsteps, cyc[2] / nsteps, cyc[3] / nsteps);
}

struct RegsPlan { int nch2, nch3, np3, cs; size_t lds; };

// the (chunks of h, chunks of [rr | h], passes of P3) triples that are instantiated: S <= 128, rank <= 64
#define FARNN_REGS_GEOMETRIES(X)                                                                              \
    X(1, 1, 1, 1) X(1, 2, 1, 1) X(1, 2, 1, 2) X(1, 3, 1, 2) X(2, 2, 1, 1) X(2, 3, 1, 1) X(2, 3, 1, 2) X(2, 4, 1, 2)   \
    X(3, 3, 2, 1) X(3, 4, 2, 1) X(3, 4, 2, 2) X(3, 5, 2, 2) X(4, 4, 2, 1) X(4, 5, 2, 1) X(4, 5, 2, 2) X(4, 6, 2, 2)
inline bool regs_has_geometry(int nch2, int nch3, int np3, int cs) {
#define FARNN_REGS_HAS(A_, B_, C_, D_) if (nch2 == A_ && nch3 == B_ && np3 == C_ && cs == D_) return true;
    FARNN_REGS_GEOMETRIES(FARNN_REGS_HAS)
#undef FARNN_REGS_HAS
    return false;
}

// Can the register kernel serve this model?  (farnn = 0, rank <= 64 so that P2 is one pass, an instantiated geometry)
inline bool regs_plan(const DecompRowsPack &k, const DecompWeights &w, int L, RegsPlan &pl) {
    if (!k.ok || w.farnn != 0 || k.n1 != 0 || k.n2 != w.R || k.n3 != w.S || w.R > DG_ROWS) return false;
    if (getenv("FARNN_DECOMP_NOREGS")) return false;
    pl.nch2 = k.nch2; pl.nch3 = k.nch3; pl.np3 = (w.S + DG_ROWS - 1) / DG_ROWS;
    pl.cs = (w.Rp + DR_CHUNK - 1) / DR_CHUNK;
    if (!regs_has_geometry(pl.nch2, pl.nch3, pl.np3, pl.cs)) return false;
    const int Lr = (L + 3) & ~3;
    pl.lds = ((size_t)2 * k.nch2 * DR_CHUNK + (size_t)2 * k.nch3 * DR_CHUNK + Lr + L + 32) * 4;
    return pl.lds <= 64 * 1024;
}

inline int launch_decomp_regs(const DecompRowsPack &k, const DecompWeights &w, const RegsPlan &pl, const int64_t *x,
                              const int64_t *len, const int *order, int sort, float *A, float *Bk, int B, int L,
                              int full, hipStream_t s) {
    DecompRegsParams p;
    p.P2[0] = k.P2[0]; p.P2[1] = k.P2[1]; p.P3[0] = k.P3[0]; p.P3[1] = k.P3[1];
    p.ld2 = k.ld2; p.ld3 = k.ld3;
    p.Vgen = w.Vgen; p.h0 = w.h0; p.hT = w.hT; p.x = x; p.len = len; p.order = order; p.sort = sort; p.A = A; p.Bk = Bk;
    p.B = B; p.L = L; p.S = w.S; p.SP = w.SP; p.R = w.R; p.Rp = w.Rp; p.nl = w.nl; p.full = full; p.V = w.V;
    { const char *e = getenv("FARNN_DBG"); p.dbg = e ? atoi(e) : 0; }
#define FARNN_REGS_CASE(A_, B_, C_, D_)                                                                       \
    if (pl.nch2 == A_ && pl.nch3 == B_ && pl.np3 == C_ && pl.cs == D_) {                                      \
        if (pl.lds > 48 * 1024)                                                                               \
            FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(decomp_regs_kernel<A_, B_, C_, D_>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));        \
        decomp_regs_kernel<A_, B_, C_, D_><<<dim3(2 * B), dim3(DG_THREADS), pl.lds, s>>>(p);                  \
        FARNN_HIP_TRY(hipGetLastError());                                                                     \
        return FARNN_OK;                                                                                      \
    }
    FARNN_REGS_GEOMETRIES(FARNN_REGS_CASE)
#undef FARNN_REGS_CASE
    return fail(FARNN_ERANGE, "decomp register kernel: no instantiation for this geometry%s%s");
}

}  // namespace farnn
