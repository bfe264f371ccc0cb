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
#include "beside.hip.h"

namespace farnn {

constexpr int DG_WAVES = 4;                   // wavefronts per chain
constexpr int DG_THREADS = DG_WAVES * 64;
constexpr int DG_ROWS = DG_THREADS / 4;       // rows per pass (64)
constexpr int DG_NG = 7;                      // state groups of 16 the scoring stage reaches here (S <= 112)

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
    BesideParams bs;              // SCORE instantiations: the scores + decode stage that runs beside the recurrence (beside.hip.h)
};

// tanh on the hardware exponential and reciprocal (|error| ~2e-7 absolute), the odd series below 1/16 (common.hip.h: tanh_series)
__device__ __forceinline__ float dg_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));          // in (0, 1]
    const float big = copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
    return fabsf(x) < TANH_SERIES_BELOW ? tanh_series(x) : big;
}
// Branch-free (round 6): the mode is a kernel argument, and as a `switch` it was three taken scalar branches on the step's chain
// (~20 cycles each: scripts/probe/issue_rate.hip) in front of a dozen instructions; then selects on its bits; now two scalars
// (common.hip.h: NlMode -- one max, one bit-field insert).  The tanh is always computed (finite for every input: e in [0, 1]).
__device__ __forceinline__ float dg_nl(float x, NlMode m) {
    const float y = nl_floor(x, m);
    return nl_pick(dg_tanh(y), y, m);
}

// four FMAs of a 16-byte piece into two pairs of partial sums: two v_pk_fma_f32.  (-DFARNN_DG_SCALAR_FMA builds them as four
// scalar v_fmac_f32 -- measured on the config-2 batch: 66.4 us per launch against 53.0; with two issue-bound wavefronts per
// SIMD the packed form wins here, unlike in chain_regs.hip.h's latency-bound step.)
__device__ __forceinline__ void dg_fma4(v2f &lo, v2f &hi, const v4f &w, const v4f &x) {
#if defined(FARNN_DG_SCALAR_FMA)
    asm("v_fmac_f32 %0, %1, %2" : "+v"(lo.x) : "v"(w.x), "v"(x.x));
    asm("v_fmac_f32 %0, %1, %2" : "+v"(lo.y) : "v"(w.y), "v"(x.y));
    asm("v_fmac_f32 %0, %1, %2" : "+v"(hi.x) : "v"(w.z), "v"(x.z));
    asm("v_fmac_f32 %0, %1, %2" : "+v"(hi.y) : "v"(w.w), "v"(x.w));
#else
    lo = __builtin_elementwise_fma(v2f{w.x, w.y}, v2f{x.x, x.y}, lo);
    hi = __builtin_elementwise_fma(v2f{w.z, w.w}, v2f{x.z, x.w}, hi);
#endif
}

// the sum of eight adjacent lanes, in every one of them (two quad levels, then the mirrored quad of the half row)
__device__ __forceinline__ float oct_sum(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true));    // row_half_mirror
    return x;
}

// FOUR row sums over the eight lanes of a group, reduce-scattered (round 6): three DPP levels in all -- the mirrored partner (lanes
// 0-3 keep passes {0, 1}, lanes 4-7 keep {2, 3}), the partner two lanes away (bit 1 of the lane picks one of the two), the neighbour
// (which holds the same pass) -- four DPP adds and six selects where four oct_sums took twelve and four.  Lane k ends with the total
// of pass oct_pass(k); both lanes of a pair {2m, 2m + 1} hold it.
__device__ __forceinline__ int oct_pass(int k) { return ((k >> 2) << 1) | ((k >> 1) & 1); }
__device__ __forceinline__ float oct_reduce_scatter4(float v0, float v1, float v2, float v3, int k) {
    auto dpp = [](float x, auto ctrl) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true)); };
    const bool hi = (k & 4) != 0, b1 = (k & 2) != 0;
    float ka = hi ? v2 : v0, kb = hi ? v3 : v1;
    const float sa = hi ? v0 : v2, sb = hi ? v1 : v3;                         // what the mirrored partner keeps
    ka += dpp(sa, std::integral_constant<int, 0x141>{});                      // row_half_mirror
    kb += dpp(sb, std::integral_constant<int, 0x141>{});
    float kk = b1 ? kb : ka;
    const float ss = b1 ? ka : kb;                                            // what the partner two lanes away keeps
    kk += dpp(ss, std::integral_constant<int, 0x4E>{});                       // quad_perm [2,3,0,1]
    kk += dpp(kk, std::integral_constant<int, 0xB1>{});                       // quad_perm [1,0,3,2]: the neighbour holds the same pass
    return kk;
}

// NCH2 / NCH3: 32-column chunks of the two input vectors (h: S columns; [rr | h]: Rp + S columns); NP3: passes of P3;
// CS: leading chunks of [rr | h] that hold rr entries (they can only be read behind the step's first barrier)
// (Tried in r02 and dropped: both chains of a sequence in one eight-wavefront workgroup with the scores + decode as its
// epilogue -- one launch per step, but the two chains in step through shared barriers ran 63.5 us against 56.9, and the
// two score tiles of the longest sequence, serial on one CU behind it, cost more than the separate launch: 76.5 us per
// step against 69.1.)
//
// SCORE (round 3): ONE launch per tagging step.  The states also go to LDS (`hist`, the whole sequence), the LAST wavefront
// copies every finished row to the stash with write-through stores (it is the only wavefront that stores: the producer half
// of the hand-off needs one drain, by that wavefront, before each signal), publishes the progress word once the other
// direction's tiles have the rows they need (pubmax) and again at the end; when the chain is done the four wavefronts run
// beside.hip.h's end-of-chain protocol: the tiles of this workgroup's half, two per pass, then the arrival.
template <int NCH2, int NCH3, int NP3, int CS, bool SCORE>
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
    // SCORE: behind them (16-byte aligned) the sequence's own states, the tiles' products and scores, the protocol's words
    const int c16 = SCORE ? p.bs.sp.c16 : 0, Kc = SCORE ? p.bs.sp.Kc : 0;
    float *hist = smem + ((2 * c2p + 2 * c3p + Lr + p.L + 32 + 3) & ~3);       // [L + 1][SP]
    float *ab = hist + (SCORE ? (p.L + 1) * SP : 0);          // [2][16][16 c16 + 4]
    float *scl = ab + 2 * RG_TT * (16 * c16 + 4);             // [2][16][Kc]
    int *misc = reinterpret_cast<int *>(scl + 2 * RG_TT * Kc);    // [32]

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
    int kmid = 0, pubmax = 0;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int WCOPY = DG_WAVES - 1;                       // SCORE: the wavefront that copies state rows to the stash
    unsigned lm_pk0 = 0u, lm_pk1 = 0u;                        // the output matrix as a label map (label_map.hip.h), when it is one
    unsigned epoch = 0u;                                      // the launch's epoch, from device memory (beside.hip.h, bs_launch_epoch)
    if (SCORE) {
        if (wv == 0 || wv == WCOPY) epoch = bs_launch_epoch(p.bs.done, lane);
        if (bs_label_map_path(p.bs.sp)) lm_load_packed(p.bs.sp.lm, lane, lm_pk0, lm_pk1);
        bs_halves(dir, len, nsteps, kmid, pubmax);
        for (int j = tid; j < (nsteps + 1) * SP; j += DG_THREADS) hist[j] = j < S ? hinit[j] : 0.0f;   // row 0; pad columns zero
        if (tid < 32) misc[tid] = tid == RGM_ACQ ? -1 : 0;
        if (wv == 0) {
            const int fo = bs_flat_offset(p.bs, b, lane);
            if (lane == 0) misc[RGM_FOFF] = fo;
        }
        if (wv == 1 && dir == 0) {                            // pad positions of LOCAL mode: tag -1, zero score rows
            for (int i = nsteps + lane; i < p.L; i += WAVE)
                if (p.bs.sp.tags) p.bs.sp.tags[(long long)b * p.L + i] = -1;
            if (p.bs.sp.scores)
                for (long long e = (long long)nsteps * p.bs.sp.K + lane; e < (long long)p.L * p.bs.sp.K; e += WAVE)
                    p.bs.sp.scores[(long long)b * p.L * p.bs.sp.K + e] = 0.0f;
        }
    } else {
        for (int j = tid; j < SP; j += DG_THREADS) stash[j] = j < S ? hinit[j] : 0.0f;             // state 0
    }
    __syncthreads();
    // SCORE: row r of `hist` -> the stash, write-through, by the copying wavefront; published = the progress word's last value
    int published = -1;
    // (a row is READ from LDS in one step and STORED in the next: stored at once, the store would wait for the LDS round trip
    //  at the top of every step, and the other three wavefronts with it at the step's barrier.  SP <= 128: one pair per lane.)
    const int cj = 2 * lane < SP ? 2 * lane : 0;
    float c0 = 0.f, c1 = 0.f;
    auto read_row = [&](int r) { c0 = hist[r * SP + cj]; c1 = hist[r * SP + cj + 1]; };
    auto store_row = [&](int r) { if (2 * lane < SP) st2_agent(stash + (long long)r * SP + cj, c0, c1); };
    auto copy_row = [&](int r) { read_row(r); store_row(r); };
    auto publish = [&](int r) {                               // every store of this (the only storing) wavefront has left, then the word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            __hip_atomic_store(p.bs.prog + (long long)dir * p.B + b, ((unsigned long long)epoch << 32) | (unsigned)r,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        published = r;
    };
    if (SCORE && wv == WCOPY) copy_row(0);
    const int t_poll = nsteps - 6 > 0 ? nsteps - 6 : 0;
    int polled = -1;
    if (nsteps <= 0) {
        if (SCORE) {
            bs_finish<DG_WAVES, DG_NG, 0, -1>(p.bs, b, dir, len, nsteps, kmid, hist, ab, scl, hist, misc, wv, lane, lm_pk0, lm_pk1, epoch,
                [&]() {}, [&]() { if (wv == WCOPY) publish(0); });
        }
        return;
    }

    // word-vector entry of this lane's P2 row, TWO steps ahead and with no branch around the load (idle lanes read entry 0):
    // vmcnt retires in order, so a value is only ever waited for when a younger load and the stash stores are already in
    // flight behind it -- one step ahead the compiler's wait for it also drained the load just issued (a full L2 round trip
    // on every step's critical path)
    const int vcol = own2 ? rslot : 0;
    auto v_addr = [&](int tk) -> const float * { return p.Vgen + (long long)tk * Rp + vcol; };
    float v0 = *v_addr(tok[0]), v1 = *v_addr(tok[nsteps > 1 ? 1 : 0]);
    int tk2 = tok[nsteps > 2 ? 2 : nsteps - 1];
    const NlMode nl_mode = nl_mode_of(p.nl);
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
        // (the copying wavefront's row and the poll of the other direction's progress: behind phase A's products -- decomp_regs8_kernel)
        if (__builtin_expect(SCORE && wv == WCOPY && t >= 1, 0)) {         // (one wavefront's: out of the others' line -- a taken branch is ~20 cycles)
            if (t >= 2) store_row(t - 1);                     // read one step ago
            read_row(t);                                      // the state the last step finished (complete behind its barrier)
            if (t - 1 == pubmax && published < pubmax) publish(pubmax);   // the other direction's tiles need no row beyond this one
        }
        if (__builtin_expect(SCORE && wv == 0 && (unsigned)(t - t_poll) <= 1u, 0)) {   // (two steps of one wavefront's chain)
            // the other direction's progress, looked at a few steps before this chain ends: polled in one step, acted on in the
            // next (the load has long landed), so that the end-of-chain protocol finds its acquire done
            if (t == t_poll && lane == 0) polled = bs_read_prog(p.bs.prog + (long long)(dir ^ 1) * p.B + b, epoch);
            if (t == t_poll + 1) {
                const int pr = __builtin_amdgcn_readfirstlane(polled);
                if (pr >= 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    if (lane == 0) misc[RGM_ACQ] = pr;
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
                if (SCORE) hist[(t + 1) * SP + row] = hn;     // (the copying wavefront takes it to the stash one step later)
                else stash[(long long)(t + 1) * SP + row] = hn;
            }
        }
        v0 = v1; v1 = v2;
        if (FARNN_PROBE_ON(p.dbg & 4096)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long c1 = __builtin_amdgcn_s_memtime(); cyc[2] += c1 - c0; c0 = c1; }
        wg_barrier_lds();
        if (FARNN_PROBE_ON(p.dbg & 4096)) { const long long c1 = __builtin_amdgcn_s_memtime(); cyc[3] += c1 - c0; }
    }
    if (SCORE) {
        if (wv == WCOPY) {                                    // (the loop's last barrier is behind us: the rows are complete)
            if (nsteps >= 2) store_row(nsteps - 1);
            copy_row(nsteps);
        }
        bs_finish<DG_WAVES, DG_NG, 0, -1>(p.bs, b, dir, len, nsteps, kmid, hist, ab, scl, hist, misc, wv, lane, lm_pk0, lm_pk1, epoch,
            [&]() {}, [&]() { if (wv == WCOPY) publish(nsteps); });
    }
    if (FARNN_PROBE_ON(p.dbg & 4096) && blockIdx.x < 2 && (tid & 63) == 0)
        printf("regs kernel wg %d wave %d: %d steps, cycles per step: A %lld  barrier %lld  B %lld  barrier %lld\n", (int)blockIdx.x, wtid >> 6,
               nsteps, cyc[0] / nsteps, cyc[1] / nsteps, cyc[2] / nsteps, cyc[3] / nsteps);
}

// ---- the same kernel with EIGHT lanes per row (round 3).  A wavefront pays 35-65 cycles for every LDS instruction it issues
// (chain_regs.hip.h's ablations), and the four-lane form issues 2 x (NCH2 + NCH3) = 18 sixteen-byte reads of the input vectors
// per step: what bounds its step.  With eight lanes on a row, lane k owns piece k of every 32-column chunk: ONE read per chunk
// (9 per step), the same weights per lane (a pass covers 32 rows: P2 takes NP2 = ceil(R / 32) passes, P3 NP3 = ceil(S / 32)),
// one more DPP level per row sum.  Lane k finishes row pass k of either product (NP2, NP3 <= 8).
// LMO: the label-map instantiation of the scores + decode stage (beside.hip.h: the matrix-core tile code is not compiled in)
template <int NCH2, int NCH3, int NP2, bool SCORE, bool LMO = false>
__global__ void __launch_bounds__(DG_THREADS, 2)
decomp_regs8_kernel(const DecompRegsParams p) {
    constexpr int NP3 = NCH2, CS = NP2;                       // S <= 32 NCH2 rows of P3; the rr entries fill the first NP2 chunks of [rr | h]
    constexpr int DG8 = DG_THREADS / 8;                       // rows per pass (32)
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
    // SCORE: behind them (16-byte aligned) the sequence's own states, the tiles' products and scores, the protocol's words
    const int c16 = SCORE ? p.bs.sp.c16 : 0, Kc = SCORE ? p.bs.sp.Kc : 0;
    float *hist = smem + ((2 * c2p + 2 * c3p + Lr + p.L + 32 + 3) & ~3);       // [L + 1][SP]
    float *ab = hist + (SCORE ? (p.L + 1) * SP : 0);          // [2][16][16 c16 + 4]
    float *scl = ab + 2 * RG_TT * (16 * c16 + 4);             // [2][16][Kc]
    int *misc = reinterpret_cast<int *>(scl + 2 * RG_TT * Kc);    // [32]

    int b = p.order ? p.order[seq] : seq;
    if (p.sort) b = select_by_length_rank(p.len, p.B, p.L, folded_rank(seq, p.B), scratch, tid, DG_THREADS);
    const int len = clamp_len(p.len[b], p.L);
    const int nsteps = p.full ? p.L : len;

    // ---- this lane's slice of the packed rows: registers for the whole sequence -------------------------------------------
    const int k = lane & 7, rslot = tid >> 3;                 // row slot 0..31 of a pass
    v4f w2[NP2][NCH2], w3[NP3][NCH3];
    {
#pragma unroll
        for (int i = 0; i < NP2; i++) {
            const int row = i * DG8 + rslot;
            const float *src = p.P2[dir] + (long long)(row < R ? row : R - 1) * p.ld2 + k * 4;
#pragma unroll
            for (int c = 0; c < NCH2; c++) w2[i][c] = *reinterpret_cast<const v4f *>(src + c * DR_CHUNK);
        }
#pragma unroll
        for (int i = 0; i < NP3; i++) {
            const int row = i * DG8 + rslot;
            const float *s3 = p.P3[dir] + (long long)(row < S ? row : S - 1) * p.ld3 + k * 4;
#pragma unroll
            for (int c = 0; c < NCH3; c++) w3[i][c] = *reinterpret_cast<const v4f *>(s3 + c * DR_CHUNK);
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
    int kmid = 0, pubmax = 0;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int WCOPY = DG_WAVES - 1;                       // SCORE: the wavefront that copies state rows to the stash
    unsigned lm_pk0 = 0u, lm_pk1 = 0u;                        // the output matrix as a label map (label_map.hip.h), when it is one
    unsigned epoch = 0u;                                      // the launch's epoch, from device memory (beside.hip.h, bs_launch_epoch)
    if (SCORE) {
        if (wv == 0 || wv == WCOPY) epoch = bs_launch_epoch(p.bs.done, lane);
        if (bs_label_map_path(p.bs.sp)) lm_load_packed(p.bs.sp.lm, lane, lm_pk0, lm_pk1);
        bs_halves(dir, len, nsteps, kmid, pubmax);
        for (int j = tid; j < (nsteps + 1) * SP; j += DG_THREADS) hist[j] = j < S ? hinit[j] : 0.0f;   // row 0; pad columns zero
        if (tid < 32) misc[tid] = tid == RGM_ACQ ? -1 : 0;
        if (wv == 0) {
            const int fo = bs_flat_offset(p.bs, b, lane);
            if (lane == 0) misc[RGM_FOFF] = fo;
        }
        if (wv == 1 && dir == 0) {                            // pad positions of LOCAL mode: tag -1, zero score rows
            for (int i = nsteps + lane; i < p.L; i += WAVE)
                if (p.bs.sp.tags) p.bs.sp.tags[(long long)b * p.L + i] = -1;
            if (p.bs.sp.scores)
                for (long long e = (long long)nsteps * p.bs.sp.K + lane; e < (long long)p.L * p.bs.sp.K; e += WAVE)
                    p.bs.sp.scores[(long long)b * p.L * p.bs.sp.K + e] = 0.0f;
        }
    } else {
        for (int j = tid; j < SP; j += DG_THREADS) stash[j] = j < S ? hinit[j] : 0.0f;             // state 0
    }
    __syncthreads();
    // SCORE: row r of `hist` -> the stash, write-through, by the copying wavefront; published = the progress word's last value
    int published = -1;
    // (a row is READ from LDS in one step and STORED in the next: stored at once, the store would wait for the LDS round trip
    //  at the top of every step, and the other three wavefronts with it at the step's barrier.  SP <= 128: one pair per lane.)
    const int cj = 2 * lane < SP ? 2 * lane : 0;
    float c0 = 0.f, c1 = 0.f;
    auto read_row = [&](int r) { c0 = hist[r * SP + cj]; c1 = hist[r * SP + cj + 1]; };
    auto store_row = [&](int r) { if (2 * lane < SP) st2_agent(stash + (long long)r * SP + cj, c0, c1); };
    auto copy_row = [&](int r) { read_row(r); store_row(r); };
    auto publish = [&](int r) {                               // every store of this (the only storing) wavefront has left, then the word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            __hip_atomic_store(p.bs.prog + (long long)dir * p.B + b, ((unsigned long long)epoch << 32) | (unsigned)r,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        published = r;
    };
    if (SCORE && wv == WCOPY) copy_row(0);
    const int t_poll = nsteps - 6 > 0 ? nsteps - 6 : 0;
    int polled = -1;
    if (nsteps <= 0) {
        if (SCORE) {
            bs_finish<DG_WAVES, DG_NG, 0, -1, false, LMO>(p.bs, b, dir, len, nsteps, kmid, hist, ab, scl, hist, misc, wv, lane, lm_pk0, lm_pk1, epoch,
                [&]() {}, [&]() { if (wv == WCOPY) publish(0); });
        }
        return;
    }

    // word-vector entry of this lane's P2 row, TWO steps ahead and with no branch around the load (idle lanes read entry 0):
    // vmcnt retires in order, so a value is only ever waited for when a younger load and the stash stores are already in
    // flight behind it -- one step ahead the compiler's wait for it also drained the load just issued (a full L2 round trip
    // on every step's critical path)
    const int row2 = k * DG8 + rslot;                         // the P2 row this lane finishes (pass k)
    const bool own2 = k < NP2 && row2 < R;
    const int vcol = own2 ? row2 : 0;
    auto v_addr = [&](int tk) -> const float * { return p.Vgen + (long long)tk * Rp + vcol; };
    float v0 = *v_addr(tok[0]), v1 = *v_addr(tok[nsteps > 1 ? 1 : 0]);
    int tk2 = tok[nsteps > 2 ? 2 : nsteps - 1];
    const NlMode nl_mode = nl_mode_of(p.nl);
    static_assert(NP3 <= 8 && NP2 <= 8, "one lane of the eight per row pass");
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
        // (round 6) the chunks of [rr | h] that hold state entries only are READ here -- they are the last step's, complete behind its
        // barrier -- but their products with P3 run in phase B, under the round trip of the rr chunks' reads: in front of the
        // barrier (rounds 2-5) those 8 (NCH3 - CS) packed FMAs per lane delayed every wavefront's arrival at it
        constexpr int NH3 = NCH3 - CS > 0 ? NCH3 - CS : 0;
        v4f xh[NH3 > 0 ? NH3 : 1];
        {
            v2f tl[NP2], th[NP2];
#pragma unroll
            for (int i = 0; i < NP2; i++) { tl[i] = v2f{0.f, 0.f}; th[i] = v2f{0.f, 0.f}; }
#pragma unroll
            for (int c = 0; c < NCH2; c++) {
                const v4f x0 = *(lds_cv4f *)(Hc + c * DR_CHUNK);
#pragma unroll
                for (int i = 0; i < NP2; i++) {
                    dg_fma4(tl[i], th[i], w2[i][c], x0);
                }
            }
            float acc2 = 0.0f;
#pragma unroll
            for (int i = 0; i < NP2; i++) {
                const v2f tt = tl[i] + th[i];
                const float a = oct_sum(tt.x + tt.y);
                acc2 = (k == i) ? a : acc2;
            }
            if (own2) X3c[row2] = acc2 * v0;
            // (issued BEHIND the products of P2 and the rr store, as explicit instructions: in front of them the three reads delayed
            //  the state chunks P2 waits for)
            const unsigned xq_h = (unsigned)(size_t)(X3c + k * 4);
            static_for<0, NH3>([&](auto c) { lds_read16_at<(CS + decltype(c)::value) * DR_CHUNK * 4>(xh[decltype(c)::value], xq_h); });
        }
        // (the copying wavefront's row and the poll of the other direction's progress: behind phase A's products, so that the step
        //  starts with its LDS reads -- at the step's top this bookkeeping sat between the barrier and them: 59.6 -> 56.5 us per
        //  launch.  The word-vector prefetch stays at the top: moved here too it cost 0.7 us.)
        if (__builtin_expect(SCORE && wv == WCOPY && t >= 1, 0)) {         // (one wavefront's: out of the others' line -- a taken branch is ~20 cycles)
            if (t >= 2) store_row(t - 1);                     // read one step ago
            read_row(t);                                      // the state the last step finished (complete behind its barrier)
            if (t - 1 == pubmax && published < pubmax) publish(pubmax);   // the other direction's tiles need no row beyond this one
        }
        if (__builtin_expect(SCORE && wv == 0 && (unsigned)(t - t_poll) <= 1u, 0)) {   // (two steps of one wavefront's chain)
            // the other direction's progress, looked at a few steps before this chain ends: polled in one step, acted on in the
            // next (the load has long landed), so that the end-of-chain protocol finds its acquire done
            if (t == t_poll && lane == 0) polled = bs_read_prog(p.bs.prog + (long long)(dir ^ 1) * p.B + b, epoch);
            if (t == t_poll + 1) {
                const int pr = __builtin_amdgcn_readfirstlane(polled);
                if (pr >= 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    if (lane == 0) misc[RGM_ACQ] = pr;
                }
            }
        }
        if (FARNN_PROBE_ON(p.dbg & 4096)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long c1 = __builtin_amdgcn_s_memtime(); cyc[0] += c1 - c0; c0 = c1; }
        wg_barrier_lds();
        if (FARNN_PROBE_ON(p.dbg & 4096)) { const long long c1 = __builtin_amdgcn_s_memtime(); cyc[1] += c1 - c0; c0 = c1; }
        // ---- phase B: the rr chunks of P3, h'[j] = nl(<[Sb[j, :] | Wd[:, j]], [rr | h]>), into the other buffers and the stash
        {
            // the rr chunks' reads are ISSUED (explicit instructions: they stay in front), the state chunks' products run under
            // their round trip, then the rr chunks' products
            const unsigned xq_a = (unsigned)(size_t)(X3c + k * 4);
            v4f xr[CS];
            static_for<0, CS>([&](auto c) { lds_read16_at<decltype(c)::value * DR_CHUNK * 4>(xr[decltype(c)::value], xq_a); });
#pragma unroll
            for (int j = 0; j < NH3; j++) asm volatile("" : "+v"(xh[j]));         // (behind the reads' issue)
#pragma unroll
            for (int i = 0; i < NP3; i++) { pl3[i] = v2f{0.f, 0.f}; ph3[i] = v2f{0.f, 0.f}; }
#pragma unroll
            for (int c = CS; c < NCH3; c++) {
#pragma unroll
                for (int i = 0; i < NP3; i++) dg_fma4(pl3[i], ph3[i], w3[i][c], xh[c - CS]);
            }
            static_for<0, CS>([&](auto c) { lds_wait_for<CS - 1 - decltype(c)::value>(xr[decltype(c)::value]); });
#pragma unroll
            for (int c = 0; c < CS; c++) {
#pragma unroll
                for (int i = 0; i < NP3; i++) dg_fma4(pl3[i], ph3[i], w3[i][c], xr[c]);
            }
            // the four passes' row sums are reduce-scattered over the eight lanes of the group (oct_reduce_scatter4): lane k ends
            // with pass oct_pass(k); the even lane of a pair finishes it, so the non-linearity of all the passes runs once
            static_assert(NP3 <= 4, "oct_reduce_scatter4: at most four passes of P3");
            float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NP3; i++) { const v2f tt = pl3[i] + ph3[i]; rs[i] = tt.x + tt.y; }
            const float mine = oct_reduce_scatter4(rs[0], rs[1], rs[2], rs[3], k);
            const int pass3 = oct_pass(k);
            const int row = pass3 * DG8 + rslot;
            if ((k & 1) == 0 && pass3 < NP3 && row < SP) {
                float hn = 0.0f;                              // pad columns of the stash stay zero
                if (row < S) {
                    hn = dg_nl(mine, nl_mode);
                    Hn[row] = hn;
                    X3n[Rp + row] = hn;
                }
                if (SCORE) hist[(t + 1) * SP + row] = hn;     // (the copying wavefront takes it to the stash one step later)
                else stash[(long long)(t + 1) * SP + row] = hn;
            }
        }
        v0 = v1; v1 = v2;
        if (FARNN_PROBE_ON(p.dbg & 4096)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long c1 = __builtin_amdgcn_s_memtime(); cyc[2] += c1 - c0; c0 = c1; }
        wg_barrier_lds();
        if (FARNN_PROBE_ON(p.dbg & 4096)) { const long long c1 = __builtin_amdgcn_s_memtime(); cyc[3] += c1 - c0; }
    }
    if (SCORE) {
        if (wv == WCOPY) {                                    // (the loop's last barrier is behind us: the rows are complete)
            if (nsteps >= 2) store_row(nsteps - 1);
            copy_row(nsteps);
        }
        bs_finish<DG_WAVES, DG_NG, 0, -1, false, LMO>(p.bs, b, dir, len, nsteps, kmid, hist, ab, scl, hist, misc, wv, lane, lm_pk0, lm_pk1, epoch,
            [&]() {}, [&]() { if (wv == WCOPY) publish(nsteps); });
    }
    if (FARNN_PROBE_ON(p.dbg & 4096) && blockIdx.x < 2 && (tid & 63) == 0)
        printf("regs8 kernel wg %d wave %d: %d steps, cycles per step: A %lld  barrier %lld  B %lld  barrier %lld\n", (int)blockIdx.x, wtid >> 6,
               nsteps, cyc[0] / nsteps, cyc[1] / nsteps, cyc[2] / nsteps, cyc[3] / nsteps);
}

struct RegsPlan { int nch2, nch3, np3, cs; size_t lds, lds_score; int np2; bool eight; };

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
    if (tun(TUN_DECOMP_NOREGS)) return false;
    pl.nch2 = k.nch2; pl.nch3 = k.nch3; pl.np3 = (w.S + DG_ROWS - 1) / DG_ROWS;
    pl.cs = (w.Rp + DR_CHUNK - 1) / DR_CHUNK;
    // the eight-lanes-per-row form (half the LDS reads per step) where it is instantiated: S in 65..128, R <= 64
    pl.np2 = (w.R + 31) / 32;
    pl.eight = !tun(TUN_DECOMP_FOUR) && (pl.nch2 == 3 || pl.nch2 == 4) && pl.np2 >= 1 && pl.np2 <= 2 &&
               (w.Rp + 31) / 32 == pl.np2 && (w.S + 31) / 32 == pl.nch2 && (pl.nch3 == pl.nch2 + pl.np2 || pl.nch3 == pl.nch2 + pl.np2 - 1);
    if (!pl.eight && !regs_has_geometry(pl.nch2, pl.nch3, pl.np3, pl.cs)) return false;
    const int Lr = (L + 3) & ~3;
    const size_t base = (size_t)2 * k.nch2 * DR_CHUNK + (size_t)2 * k.nch3 * DR_CHUNK + Lr + L + 32;
    pl.lds = base * 4;
    pl.lds_score = 0;                                         // set by the caller when the scoring stage can ride along
    return pl.lds <= 64 * 1024;
}

// LDS of the SCORE instantiation: + the sequence's states, two tiles' products and scores, the protocol's words
inline size_t regs_score_lds(const RegsPlan &pl, int L, int SP, int c16, int Kc) {
    const size_t base = (pl.lds / 4 + 3) & ~(size_t)3;
    return (base + (size_t)(L + 1) * SP + (size_t)2 * RG_TT * (16 * c16 + 4) + (size_t)2 * RG_TT * Kc + 32) * 4;
}

// bs != nullptr: the SCORE instantiation (the caller has checked regs_score_lds <= 80 KiB, S <= 16 DG_NG, K <= 256)
inline int launch_decomp_regs(const DecompRowsPack &k, const DecompWeights &w, const RegsPlan &pl, const int64_t *x,
                              const int64_t *len, const int *order, int sort, float *A, float *Bk, int B, int L,
                              int full, hipStream_t s, const BesideParams *bs = nullptr) {
    DecompRegsParams p;
    memset(&p.bs, 0, sizeof(p.bs));
    if (bs) p.bs = *bs;
    const size_t lds = bs ? pl.lds_score : pl.lds;
    p.P2[0] = k.P2[0]; p.P2[1] = k.P2[1]; p.P3[0] = k.P3[0]; p.P3[1] = k.P3[1];
    p.ld2 = k.ld2; p.ld3 = k.ld3;
    p.Vgen = w.Vgen; p.h0 = w.h0; p.hT = w.hT; p.x = x; p.len = len; p.order = order; p.sort = sort; p.A = A; p.Bk = Bk;
    p.B = B; p.L = L; p.S = w.S; p.SP = w.SP; p.R = w.R; p.Rp = w.Rp; p.nl = w.nl; p.full = full; p.V = w.V;
    p.dbg = tun(TUN_DBG);
#define FARNN_REGS8_CASE(A_, B_, C_)                                                                          \
    if (pl.eight && pl.nch2 == A_ && pl.nch3 == B_ && pl.np2 == C_) {                                         \
        if (bs && bs_label_map_path(bs->sp)) {                                                                \
            FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(decomp_regs8_kernel<A_, B_, C_, true, true>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));        \
            decomp_regs8_kernel<A_, B_, C_, true, true><<<dim3(2 * B), dim3(DG_THREADS), lds, s>>>(p);        \
        } else if (bs) {                                                                                      \
            FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(decomp_regs8_kernel<A_, B_, C_, true>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));        \
            decomp_regs8_kernel<A_, B_, C_, true><<<dim3(2 * B), dim3(DG_THREADS), lds, s>>>(p);              \
        } else {                                                                                              \
            if (lds > 48 * 1024)                                                                              \
                FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(decomp_regs8_kernel<A_, B_, C_, false>), \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));    \
            decomp_regs8_kernel<A_, B_, C_, false><<<dim3(2 * B), dim3(DG_THREADS), lds, s>>>(p);             \
        }                                                                                                     \
        FARNN_HIP_TRY(hipGetLastError());                                                                     \
        return FARNN_OK;                                                                                      \
    }
    FARNN_REGS8_CASE(3, 3, 1) FARNN_REGS8_CASE(3, 4, 1) FARNN_REGS8_CASE(3, 4, 2) FARNN_REGS8_CASE(3, 5, 2)
    FARNN_REGS8_CASE(4, 4, 1) FARNN_REGS8_CASE(4, 5, 1) FARNN_REGS8_CASE(4, 5, 2) FARNN_REGS8_CASE(4, 6, 2)
#undef FARNN_REGS8_CASE
#define FARNN_REGS_CASE(A_, B_, C_, D_)                                                                       \
    if (pl.nch2 == A_ && pl.nch3 == B_ && pl.np3 == C_ && pl.cs == D_) {                                      \
        if (bs) {                                                                                             \
            FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(decomp_regs_kernel<A_, B_, C_, D_, true>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));        \
            decomp_regs_kernel<A_, B_, C_, D_, true><<<dim3(2 * B), dim3(DG_THREADS), lds, s>>>(p);           \
        } else {                                                                                              \
            if (lds > 48 * 1024)                                                                              \
                FARNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(decomp_regs_kernel<A_, B_, C_, D_, false>), \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));    \
            decomp_regs_kernel<A_, B_, C_, D_, false><<<dim3(2 * B), dim3(DG_THREADS), lds, s>>>(p);          \
        }                                                                                                     \
        FARNN_HIP_TRY(hipGetLastError());                                                                     \
        return FARNN_OK;                                                                                      \
    }
    FARNN_REGS_GEOMETRIES(FARNN_REGS_CASE)
#undef FARNN_REGS_CASE
    return fail(FARNN_ERANGE, "decomp register kernel: no instantiation for this geometry%s%s");
}

}  // namespace farnn
