// The CRF decode with the partitions' history in LDS (K2b): viterbi_hist_body / viterbi_hist_kernel, its LDS sizes, and the
// flat-offset helper the decode kernels share.  A header of its own because two translation units hold it: farnn_hip.hip (the
// stand-alone kernel behind a recurrence kernel) and chain_viterbi.hip (the same body as the epilogue of the recurrence: ONE
// launch per tagging step of a CRF model).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include "common.hip.h"
#include "score_params.hip.h"

namespace farnn {

// flat-output offset of sequence b without a prepared prefix array: the sum of the (clamped) lengths in
// front of it (utils.py:153-164).  Called by every thread of the workgroup; B <= 1024.
__device__ __forceinline__ long long flat_offset_in_kernel(const int64_t *len, int b, int L, int tid, int nthreads) {
    __shared__ int fo_w[16];
    int part = 0;
    for (int j = tid; j < b; j += nthreads) {
        const int v = (int)len[j];
        part += v < 0 ? 0 : (v > L ? L : v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, WAVE);
    if ((tid & 63) == 0) fo_w[tid >> 6] = part;
    __syncthreads();
    int tot = 0;
    for (int ww = 0; ww < (nthreads >> 6); ww++) tot += fo_w[ww];
    __syncthreads();
    return (long long)tot;
}


// The dynamic programme's wavefronts: eight lanes per tag pair on full wavefronts (eight pairs each); up to four pairs left over go
// to ONE tail wavefront that spreads a pair over GT lanes (sources at stride GT); five to seven run as one more full wavefront.
__host__ __device__ inline int viterbi_tail_lanes(int rem) { return rem <= 1 ? 64 : rem == 2 ? 32 : rem <= 4 ? 16 : 8; }
__host__ __device__ inline int viterbi_dp_threads(int K) { return (((((K + 1) >> 1) * 8) + 63) & ~63); }

// The fused form's alpha*beta products, staged where the history will be: row-major [token][SPq] (round 6; rounds 2-5 staged them
// transposed, [state][L + 16]: the 26 lanes that share a token wrote 320 floats apart -- one bank, a 26-way conflict on every one
// of the staging's stores: 7.3 k cycles at S = 104, L = 64).  SPq / 4 is odd, so the sixteen token rows of a matrix-core A fragment
// start in sixteen different groups of four banks and its four state columns fill the group: reads and writes conflict-free.
// A row holds whole groups of 16 states (the pad columns are zeroed): the matrix-core loop reads its A entries at fixed offsets from
// ONE address it advances by a group, with no clamp into the row.
__host__ __device__ inline int viterbi_products_stride(int SP) { const int q = (SP + 15) & ~15; return ((q >> 2) & 1) ? q : q + 4; }
__host__ __device__ inline size_t viterbi_products_floats(int SP, int L) {
    return (size_t)((L + 15) & ~15) * viterbi_products_stride(SP);
}
// floats of the history area of viterbi_hist_kernel: [L][Kp] partitions; the fused form first stages the products there
__host__ __device__ inline size_t viterbi_hist_floats(int Kp, int SP, int L, bool fused) {
    const size_t a = (size_t)L * Kp, c = fused ? viterbi_products_floats(SP, L) : 0;
    return a > c ? a : c;
}

// 1 KiB pieces of the LDS area behind the scores: the transposed transition table -- and, for the fused form with a dense output
// matrix, first the matrix-core image of that matrix ((K / 16) x (S / 16) pieces).  Rounds 2-5 kept the image in LDS only when it
// fitted the table's own size; K = 75 tags over S = 104 states (the shipped configurations: a 23-piece table, a 35-piece image)
// read every B fragment from L2 instead -- two dependent round trips per pair of state groups, 10.8 k cycles for a product whose
// matrix-core time is 5.4 k.  The area is now the larger of the two whenever the LDS holds it (62 + 12 KB at that shape).
__host__ __device__ inline int viterbi_table_pieces(int K, int Kp, int SP, int L, bool fused) {
    const int tr_pieces = (K * Kp * 4 + 1023) / 1024;
    if (!fused) return tr_pieces;
    const int otm_pieces = ((K + 15) >> 4) * ((SP + 15) >> 4);
    if (otm_pieces <= tr_pieces) return tr_pieces;
    const size_t with_image = viterbi_hist_floats(Kp, SP, L, true) * 4 + ((size_t)L * Kp * 4 + 1023) / 1024 * 1024 + (size_t)otm_pieces * 1024;
    return with_image <= 158 * 1024 ? otm_pieces : tr_pieces;
}

// History variant of the DP (used when the LDS holds it): the forward pass keeps only the partition
// VALUES of every step; the back-pointers the reference stores (crf.py:147-149) are recomputed lazily along the ONE
// path the backtrace follows: bp_t[j] = first argmax_i ((f_t[j] + tr[i][j]) + part_{t-1}[i]) is the same f32
// expression on the same values, so the path is bit-identical.  The transposed transition table stays in LDS for
// that second pass.
// Forward pass layout (r02; FARNN_DBG=8192 prints the phase cycle counts).  The step is VALU-issue bound on the one CU
// that owns the sequence: two adds (crf.py:123,145) and half a v_max3 per (source, tag), K*K pairs -- 660 cycles per step
// at K = 130 if the four SIMDs were perfectly balanced, plus ~300 of LDS latency and barrier.  EIGHT lanes share a PAIR of
// destination tags (j0, j0+1); lane g of the group owns the sources 32*k + 4*g .. +3 of every 32-source block k < IB4 (a
// 16-byte LDS read per block, the eight lanes cover a contiguous 128 bytes: no bank conflicts) plus up to four leftover
// sources 32*IB4 + g + 8x (K - 32*IB4 < 32): at most one wasted source slot per lane for any K (r01: four lanes per tag,
// block sizes 8/16/36/52/64 -- K = 75 ran 36 slots per lane for 19 useful ones).  The two tags ride in the halves of
// packed-f32 registers (v_pk_add_f32 issues at half rate on gfx950, so this saves registers and LDS reads, not issue
// slots); the group is combined by three v_max with DPP operands per tag.  Tag pairs that do not fill a wavefront (K = 130:
// the START/STOP pair) go to a TAIL wavefront that spreads them over all its lanes and runs at raised priority -- its
// step is a short latency chain that otherwise runs behind the SIMD's older, issue-bound wavefronts.
// Measured on the config-3 batch (K = 130, 64 positions; cycles at ~2.3 GHz): set-up + scores 25.6 k, forward pass
// 1 190 per step (K = 128: 910), backtrace 590 per step (r01: 1 140 -- the keyed 64-bit DPP argmax).  Tried and dropped:
// hoisting (f + tr) of the coming step behind the LDS write (no gain: the idle time there is ~100 cycles), conditional
// leftover reads (waits inside branches: +130 per step).
// FUSED: the workgroup also computes the clamped scores of its sequence (what score_tile_kernel would
// have written to crf_scores) straight into LDS -- one kernel from stash to tags, no score round trip
// through HBM: alpha*beta products staged transposed in the (not yet used) history area, then a
// register-blocked [tokens x S].[S x K] product, 4 tokens x 4 tags per lane, against the L2-resident
// transposed output matrix.  Same fmaf chain in s order as score_tile_kernel: identical bits.
// The kernel's body as a function of (LDS base, thread index, thread count, sequence): viterbi_hist_kernel runs it on one
// workgroup per sequence; chain_viterbi_kernel runs it behind the two chains of the sequence, on the first `nthreads` threads
// of their workgroup (the others have left: a workgroup barrier counts the wavefronts that are still there).
// ldsF / ldsB (FUSED only): the sequence's forward / backward state rows [L + 1][SP] where they still lie in LDS (they must not
// overlap the products' area, the first SP * (L + 16) floats), else nullptr: the rows are read from the stash.  image_staged:
// the output matrix's matrix-core image is already in the transition table's LDS area.
// lm_pk: the caller's copy of this lane's two packed label-map words (label_map.hip.h), fetched long ago (chain_viterbi_kernel: at
// the kernel's start), or nullptr: they are fetched here, an L2 round trip in front of the first token
// ns (round 6): the threads that run the body up to the end of the SCORES (ns >= nthreads, whole wavefronts): the dynamic programme
// takes nthreads = 8 lanes per tag pair (K = 75: five wavefronts, two of them on one SIMD), the scores want every SIMD's matrix
// core and more loads in flight -- the stand-alone kernel is launched with up to sixteen wavefronts, threads nthreads .. ns - 1
// leave behind the scores' last barrier (a workgroup barrier counts the wavefronts that are still there).
template <int IB4, bool FUSED>
__device__ __forceinline__ void viterbi_hist_body(const ScoreParams &p, float *smem, const int tid, const int nthreads, const int b,
                                                  const float *ldsF = nullptr, const float *ldsB = nullptr, const bool image_staged = false,
                                                  const unsigned *lm_pk = nullptr, const int *pre = nullptr, const int ns_in = 0) {
    const int ns = (FUSED && ns_in > nthreads) ? ns_in : nthreads;
    // pre (chain_viterbi_kernel): {the sequence's length, its flat-output offset} in LDS, worked out by an idle wavefront while the
    // chains ran -- else two global round trips (the length, then the lengths in front of it) open the decode
    static_assert(IB4 <= 6, "K >= 224: the transition table does not fit the LDS beside a history (launch_viterbi: viterbi_kernel)");
    constexpr int IB = IB4 * 4;
    const int lane = tid & 63, w = tid >> 6;
    // The loads that open the decode are ISSUED together and used behind the flat offset's barriers (round 6): the sequence's length,
    // the label-map words and the lengths in front of the sequence used to be three dependent L2 round trips in a row at the head of
    // every workgroup (~2 k cycles each; then the state rows, a fourth).
    const long long pt0 = FARNN_PROBE_ON(p.dbg & 8192) ? (long long)__builtin_amdgcn_s_memtime() : 0;      // the body's entry
    const long long len_raw = pre ? 0 : p.len[b];
    unsigned lm_own[2] = {0u, 0u};
    if (FUSED && p.lm.on && !lm_pk) lm_load_packed(p.lm, lane, lm_own[0], lm_own[1]);
    (void)p.full;
    const int K = p.K, Kp = p.Kp;
    const int PW = Kp;                                   // partition row stride (K rounded up to 4), pads -inf
    float *hist = smem;                                  // [L][PW] partitions of every step (FUSED: first the products)
    const int sc_pieces = (p.L * Kp * 4 + 1023) / 1024, tr_pieces = (K * Kp * 4 + 1023) / 1024;
    float *scl = hist + viterbi_hist_floats(Kp, p.SP, p.L, FUSED);   // [L][Kp] clamped scores of this sequence (whole KiB)
    float *trl = scl + sc_pieces * 256;                  // [K][Kp] trT: trl[j][i] = transitions[i][j]
    const float *sc = p.crf_scores + (long long)b * p.L * Kp;
    const long long foff = p.offs ? p.offs[b] : (p.flat ? (pre ? (long long)pre[1] : flat_offset_in_kernel(p.len, b, p.L, tid, ns)) : 0);
    const int n = pre ? pre[0] : clamp_len(len_raw, p.L);
    const long long pf0 = FARNN_PROBE_ON(p.dbg & 8192) ? (long long)__builtin_amdgcn_s_memtime() : 0;     // (the opening loads have landed)
    const int START = K - 2, STOP = K - 1;
    const float ninf = -INFINITY;
    const int wu = __builtin_amdgcn_readfirstlane(w), nwaves = nthreads >> 6, nsw = ns >> 6;
    const bool probe = FARNN_PROBE_ON(p.dbg & 8192) && n == p.L;       // diagnostic: cycle counts of the phases of a full-length sequence
    long long pc0 = probe ? pt0 : 0, pc1 = 0, pc2 = 0, pc3 = 0, pa = 0, pb = 0, pw = 0;

    // ---- who does what: full wavefronts own eight tag pairs each (eight lanes per pair); the pairs left over go to one
    // TAIL wavefront that spreads them over all its lanes (GT = 64, 32, 16 or 8 lanes per pair, sources at stride GT), so
    // that e.g. K = 130 (64 pairs + START/STOP) costs the ninth wavefront 3 source slots per lane instead of 17
    const int npairs = (K + 1) >> 1, nfull = npairs >> 3, rem = npairs & 7;
    // (round 6: five to seven left-over pairs -- K = 75: 38 pairs, six left over -- would give the tail eight lanes per pair too: the
    //  same source slots as a full wavefront through the tail's simpler, unpipelined loop, which then set the step: 795 cycles.  The
    //  last wavefront runs the full wavefronts' loop instead, its spare groups idle: 635.  Measured behind it and dropped, both at
    //  625-640: the left-over pairs on three light tail wavefronts of 32 lanes per pair; (f + tr) computed under the reads' round
    //  trip.  The step is one wavefront's chain -- reads, ~250 cycles of adds and maxima, DPP, write, barrier -- not a SIMD's load.)
    const int GT = viterbi_tail_lanes(rem);
    const bool tail = wu >= nfull && GT > 8;             // wave-uniform
    const int g = tail ? (lane & (GT - 1)) : (tid & 7);  // lane of its group
    const int grp = tail ? lane / GT : 0;
    const int pair = tail ? nfull * 8 + grp : (tid >> 3);
    const int j0 = 2 * pair;
    const bool own0 = j0 < K && (!tail || grp < rem), own1 = own0 && j0 + 1 < K;
    const bool writer = own0 && (tail ? g == GT - 1 : g == 0);
    const int XS = (K - 8 * IB + 7) >> 3;                // full wavefronts: leftover source slots per lane (0..4)
    const int nst = (K + GT - 1) / GT;                   // tail wavefront: source slots per lane (<= IB + 4)
    constexpr int NSL = IB + 4;
    v2f trs[NSL];                                        // tr[i][j0], tr[i][j0+1] of this lane's sources
#define FARNN_TRS_SET(SL, X, Y) do { trs[SL] = v2f{(X), (Y)}; } while (0)
    int ixs[4];                                          // full: leftover sources (clamped into the row; their tr is -inf)
    // (round 6: requested HERE, in front of the scores -- the entries depend on the lane only, and their L2 round trip used to open
    //  the decode behind the scores' last barrier.  Round 4 had measured this at -0.8 us for the stand-alone kernel and dropped it
    //  for the one-launch form's sake, which has since moved to the A/B build.)
    if (tid < nthreads) {
        const float *row0 = p.trT + (long long)(own0 ? j0 : 0) * Kp, *row1 = p.trT + (long long)(own1 ? j0 + 1 : 0) * Kp;
        if (!tail) {
#pragma unroll
            for (int k4 = 0; k4 < IB4; k4++) {           // sources 32*k4 + 4*g + u: a 128-byte span per read, no bank conflicts
                const float4 a = ld4(row0 + k4 * 32 + g * 4), c = ld4(row1 + k4 * 32 + g * 4);
                trs[k4 * 4 + 0] = v2f{own0 ? a.x : ninf, own1 ? c.x : ninf}; trs[k4 * 4 + 1] = v2f{own0 ? a.y : ninf, own1 ? c.y : ninf};
                trs[k4 * 4 + 2] = v2f{own0 ? a.z : ninf, own1 ? c.z : ninf}; trs[k4 * 4 + 3] = v2f{own0 ? a.w : ninf, own1 ? c.w : ninf};
            }
#pragma unroll
            for (int xk = 0; xk < 4; xk++) {
                const int i = 8 * IB + 8 * xk + g;
                const bool ok = i < K;
                ixs[xk] = ok ? i : K - 1;
                FARNN_TRS_SET(IB + xk, (ok && own0) ? row0[ixs[xk]] : ninf, (ok && own1) ? row1[ixs[xk]] : ninf);
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < NSL; sl++) {
                const int i = g + GT * sl;
                const bool ok = i < K;
                FARNN_TRS_SET(sl, (ok && own0) ? row0[ok ? i : K - 1] : ninf, (ok && own1) ? row1[ok ? i : K - 1] : ninf);
            }
#pragma unroll
            for (int xk = 0; xk < 4; xk++) ixs[xk] = 0;
        }
    }
    const v2f t_start = (writer && tid < nthreads) ? v2f{p.trT[(long long)j0 * Kp + START], own1 ? p.trT[(long long)(j0 + 1) * Kp + START] : 0.0f}
                               : v2f{0.f, 0.f};                                                  // before the table DMA
    // set-up without a register round trip: the scores and (behind them) the transition table stream
    // into LDS by LDS-DMA; the table is only needed by the backtrace, so its pieces stay in flight
    // during the forward pass (counted vmcnt: this wavefront's table pieces are its youngest operations)
    if (!FUSED) {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)scl);
        const int need = (n * Kp * 4 + 1023) / 1024;
        for (int k = wu; k < need; k += nwaves)
            lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, reinterpret_cast<const char *>(sc), lds0 + (unsigned)k * 1024u);
    } else if (p.lm.on) {
        // the output matrix is a label map (label_map.hip.h): a wavefront turns a token's two state rows into its emission row in
        // ~50 instructions -- no products staged, no matrix-core pass, no image of the output matrix
        const int SP = p.SP;
        LabelMapRegs lr;
        if (lm_pk) lm_unpack(p.lm, lm_pk[0], lm_pk[1], lr);
        else lm_unpack(p.lm, lm_own[0], lm_own[1], lr);
        if (ldsF) {
            // two tokens at a time (their scans interleave: label_map.hip.h), every row entry fetched before the scan
            const bool two = p.lm.nq > 1;                 // (selected, not multiplied by 0: label_map.hip.h)
            for (int tok = wu; tok < n; tok += 2 * nsw) {
                const int tk2 = tok + nsw < n ? tok + nsw : tok;                   // (an odd token out: scored twice, stored once)
                const float *fa = ldsF + (tok + 1) * SP, *ba = ldsB + (n - (tok + 1)) * SP;
                const float *fb = ldsF + (tk2 + 1) * SP, *bb = ldsB + (n - (tk2 + 1)) * SP;
                const float xa0 = fa[lr.st0] * ba[lr.st0], xa1 = two ? fa[lr.st1] * ba[lr.st1] : 0.0f;
                const float xb0 = fb[lr.st0] * bb[lr.st0], xb1 = two ? fb[lr.st1] * bb[lr.st1] : 0.0f;
                float ya0, ya1, yb0, yb1;
                lm_scan_scores2(lr, xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1);
                lm_store_emissions(p.lm, lr, ya0, ya1, scl + (size_t)tok * Kp, Kp, lane);
                if (tk2 != tok) lm_store_emissions(p.lm, lr, yb0, yb1, scl + (size_t)tk2 * Kp, Kp, lane);
            }
        } else {
            // rows from the stash: every load of this wavefront's tokens in flight before the first is used
            const float *Ab = p.A + (long long)b * (p.L + 1) * SP, *Bb = p.Bk + (long long)b * (p.L + 1) * SP;
            constexpr int NT = 4;                        // tokens per batch
            for (int t0 = wu * NT; t0 < n; t0 += nsw * NT) {
                float a0[NT], b0[NT], a1[NT], b1[NT];
#pragma unroll
                for (int u = 0; u < NT; u++) {
                    const int tok = t0 + u < n ? t0 + u : n - 1;
                    const float *ar = Ab + (long long)(tok + 1) * SP, *br = Bb + (long long)(n - (tok + 1)) * SP;
                    a0[u] = ar[lr.st0]; b0[u] = br[lr.st0];
                    a1[u] = ar[lr.st1]; b1[u] = br[lr.st1];
                }
                if (probe && t0 == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pw = (long long)__builtin_amdgcn_s_memtime(); }   // (the first batch's rows have landed)
#pragma unroll
                for (int u = 0; u < NT; u++) {
                    if (t0 + u >= n) break;
                    float y0, y1;
                    lm_scan_scores(lr, a0[u] * b0[u], p.lm.nq > 1 ? a1[u] * b1[u] : 0.0f, y0, y1);
                    lm_store_emissions(p.lm, lr, y0, y1, scl + (size_t)(t0 + u) * Kp, Kp, lane);
                }
            }
        }
        __syncthreads();
        if (probe) { pa = (long long)__builtin_amdgcn_s_memtime(); pb = pa; }
    } else {
        const int SP = p.SP, SP4 = SP >> 2, SPq = viterbi_products_stride(SP);   // row stride of the products: viterbi_products_floats
        // the matrix-core image of the output matrix (K2: one 1 KiB piece per (column block, state group)) borrows the
        // transition table's LDS area until the scores are done, when it fits
        const int otm_pieces = ((K + 15) >> 4) * p.c16;
        const bool otm_lds = otm_pieces <= viterbi_table_pieces(K, Kp, SP, p.L, true);
        if (otm_lds && !image_staged) {
            const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)trl);
            for (int k = wu; k < otm_pieces; k += nsw)
                lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, reinterpret_cast<const char *>(p.OTm), lds0 + (unsigned)k * 1024u);
        }
        __shared__ int vh_queue[4];                      // next unit of each SIMD's queue (product_lds)
        if (tid < 4) vh_queue[tid] = tid;
        float *ab = hist;                                // [tokens][SPq] alpha*beta (aliases hist); rows of tokens >= n are never
                                                         // written: a matrix-core row depends on its own A row only, and is not stored
        const float *Ab = p.A + (long long)b * (p.L + 1) * SP;
        const float *Bb = p.Bk + (long long)b * (p.L + 1) * SP;
        // consecutive lanes take consecutive quads of ONE token's row: coalesced 16-byte loads, consecutive 16-byte LDS stores
        {
            const int pq = (SPq - SP) >> 2;              // pad quads of a row: zero (the output matrix's image is zero there too: 0 x 0)
            for (int idx = tid; idx < n * pq; idx += ns) {
                const int tok = idx / pq;
                *reinterpret_cast<v4f *>(ab + tok * SPq + SP + (idx - tok * pq) * 4) = v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (ldsF) {                                      // the chains ran in this workgroup: their rows never left the LDS
            lds_cfloat *lf = (lds_cfloat *)ldsF, *lb = (lds_cfloat *)ldsB;
            for (int idx = tid; idx < n * SP4; idx += ns) {
                const int tok = idx / SP4, s4 = (idx - tok * SP4) * 4;
                const v4f a4 = *(lds_cv4f *)(lf + (tok + 1) * SP + s4), b4 = *(lds_cv4f *)(lb + (n - (tok + 1)) * SP + s4);
                *reinterpret_cast<v4f *>(ab + tok * SPq + s4) = v4f{a4.x * b4.x, a4.y * b4.y, a4.z * b4.z, a4.w * b4.w};
            }
        } else
        for (int idx = tid; idx < n * SP4; idx += ns) {
            const int tok = idx / SP4, s4 = (idx - tok * SP4) * 4;
            // alpha = state after tok+1 tokens; beta = backward state before token tok+1 (:415-420)
            const float4 a4 = ld4(Ab + (long long)(tok + 1) * SP + s4), b4 = ld4(Bb + (long long)(n - (tok + 1)) * SP + s4);
            *reinterpret_cast<v4f *>(ab + tok * SPq + s4) = v4f{a4.x * b4.x, a4.y * b4.y, a4.z * b4.z, a4.w * b4.w};
        }
        __syncthreads();                                 // (drains vmcnt too: the image has landed)
        if (probe) pa = (long long)__builtin_amdgcn_s_memtime();
        // scores[n][K] = abT^T . O^T on the f32 matrix cores (v_mfma_f32_16x16x4_f32: the ascending-s fmaf chain of K2, same
        // bits): units of one 16-token block x two 16-tag blocks (shared A fragments from the staged products, independent
        // accumulators), B fragments from the matrix-core image of the output matrix (OTm, K2: staged in LDS by LDS-DMA when
        // it fits the transition table's area, else from L2), two state groups ahead.  Every score-phase wavefront takes units
        // (wavefront w sits on SIMD w % 4, HW_ID).  Rounds 2-5 ran this phase on the dynamic programme's wavefronts only: K = 75
        // = five, two of them on SIMD 0, which so carried five of a 64-token sequence's twelve units -- 16.7 k cycles for 672
        // matrix-core instructions whose own time is 5.4 k over four SIMDs (profiles/r06_base_probe_viterbi_k75_*).
        {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const int lr = lane & 15, lk = lane >> 4;
            // column blocks whose output rows are zero by construction are not computed: their scores are +0.0 (the fmaf chain of
            // finite products and zeros, then + 0.0f).  K = 130 = 128 labels + START / STOP: 8 blocks instead of 9, and the 16
            // (token block, column-block pair) units fall on the eight wavefronts in two even rounds instead of 20 in three
            const int kz = (p.kz > 0 && p.kz < K) ? p.kz : K;
            const int c16 = p.c16, ntb = (n + 15) >> 4, ncb = (kz + 15) >> 4, nblk = ntb * ncb;
            for (int i = tid; i < n * (Kp - 16 * ncb); i += ns) {             // (nothing when every block is computed: 16 ncb >= Kp)
                const int w_ = Kp - 16 * ncb, tok = i / w_, col = 16 * ncb + i - tok * w_;
                scl[(long long)tok * Kp + col] = 0.0f;
            }
            const int ngw = nsw;                         // every score-phase wavefront takes units: wavefront w sits on SIMD w % 4, so
                                                         // units w, w + ngw, ... load the four matrix cores evenly
            const int clamp_col = K - 3;                 // model_decompose.py:353
            // Both operands in LDS (the usual case): units of ONE 16-token block x ONE 16-tag block, drawn from four queues -- one
            // per SIMD (HW_ID[5:4]; unit u belongs to queue u % 4, a wavefront takes its SIMD's next unit with an LDS atomic).  Every
            // matrix core gets a quarter of the units whatever wavefronts the hardware placed on its SIMD, and a SIMD's wavefronts
            // share its queue.  (Rounds 2-5, and round 6's first form, dealt units to wavefront w by w's index on the assumption
            // "wavefront w sits on SIMD w % 4": wavefront 0's unit was done 5.8 k cycles after the products, the last one 9.5 k --
            // profiles/r06_probe_viterbi_k75_*.)  The reads of a state group are explicit instructions, issued a group ahead and
            // retired by a counted wait (five per group: four A entries, one B fragment; LDS returns in order): left to the compiler
            // the loop waited lgkmcnt(0) in front of every eight matrix-core instructions and read the odd groups' operands one at
            // a time, each behind a full wait -- ~8 exposed LDS round trips per pair of groups (from the ISA).
            auto product_lds = [&]() {
                const int simd = (int)(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3);      // HW_ID[5:4]
                const int nunits = ntb * ncb;
                // a wavefront drains its own SIMD's queue, then the others' (nothing guarantees that every SIMD hosts one of the
                // workgroup's wavefronts -- a three-wavefront body inside the one-launch kernel does not: round 6's switch matrix)
                for (int qi = 0; qi < 4;) {
                    const int q = (simd + qi) & 3;
                    int unit = 0;
                    if (lane == 0) unit = atomicAdd(&vh_queue[q], 4);
                    unit = __builtin_amdgcn_readfirstlane(unit);
                    if (unit >= nunits) { qi++; continue; }
                    const int tb = unit / ncb, cb = unit - tb * ncb;
                    // this lane's entries of a state group: A = its token row at columns 16 g + 4 e + lk (e = 0..3: 16 bytes apart --
                    // offset immediates; the row is padded with zeros to whole groups, so nothing is clamped), B = its 16 bytes of the
                    // image's piece g.  ONE address per operand, advanced once per group: left as index arithmetic the loop spent 14
                    // vector instructions per group on addresses (scripts/probe/score_product.hip: 60 -> 42 cycles per matrix-core
                    // instruction and SIMD at sixteen wavefronts).
                    unsigned pa = (unsigned)(size_t)(ab + (tb * 16 + lr) * SPq + lk);
                    unsigned pb_ = (unsigned)(size_t)(trl + (size_t)cb * c16 * 256 + lane * 4);
                    auto issue = [&](float (&a)[4], f32x4 &bf) {
                        asm volatile("ds_read_b32 %0, %5\n\tds_read_b32 %1, %5 offset:16\n\tds_read_b32 %2, %5 offset:32\n\tds_read_b32 %3, %5 offset:48\n\t"
                                     "ds_read_b128 %4, %6"
                                     : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(bf) : "v"(pa), "v"(pb_) : "memory");
                        pa += 64u; pb_ += 1024u;
                    };
                    auto landed = [&](float (&a)[4], f32x4 &bf, auto cnt) {                            // all but the `cnt` youngest reads have landed
                        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(bf) : "n"(decltype(cnt)::value));
                    };
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    auto mfma4 = [&](const float (&a)[4], const f32x4 &bf) {                          // the ascending-s fmaf chain of K2
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], bf.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], bf.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], bf.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], bf.w, acc, 0, 0, 0);
                    };
                    float ae[4], ao[4];
                    f32x4 be, bo;
                    issue(ae, be);
#pragma unroll 1
                    for (int g = 0; g < c16; g += 2) {
                        issue(ao, bo);                                                                 // (past the last group: read, never used)
                        landed(ae, be, std::integral_constant<int, 5>{});
                        mfma4(ae, be);
                        issue(ae, be);
                        landed(ao, bo, std::integral_constant<int, 5>{});
                        if (g + 1 < c16) mfma4(ao, bo);
                    }
                    landed(ae, be, std::integral_constant<int, 0>{});                                  // (the trailing, unused group)
                    const float av[4] = {acc.x, acc.y, acc.z, acc.w};                                  // rows lk*4 + r of the token block, column lr
                    const int col = cb * 16 + lr;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int tok = tb * 16 + lk * 4 + r;
                        float v = av[r] + 0.0f;                                                        // -0.0 -> +0.0 like score_tile_kernel
                        if (col == clamp_col) v = fminf(v, p.threshold);
                        if (tok < n && col < Kp) scl[(long long)tok * Kp + col] = v;
                    }
                }
            };
            auto product = [&](auto otm) {               // otm: this lane's entry of the image, typed LDS or global pointer
            const int ncp = (ncb + 1) >> 1;              // a unit = one token block x TWO column blocks: shared A fragments,
#pragma unroll 1                                         // two independent accumulators
            for (int unit = wu; unit < ntb * ncp && wu < ngw; unit += ngw) {
                const int tb = unit / ncp, cb0 = 2 * (unit - tb * ncp), cb1 = cb0 + 1 < ncb ? cb0 + 1 : cb0;
                auto bp0 = otm + cb0 * c16 * 64, bp1 = otm + cb1 * c16 * 64;
                lds_cfloat *ap = (lds_cfloat *)ab + (tb * 16 + lr) * SPq;     // this lane's token row
                auto a_group = [&](int g, float (&a)[4]) {
                    const int gc = g < c16 ? g : c16 - 1;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int s = 16 * gc + 4 * e + lk;                   // this lane's k of k-step 4g + e
                        a[e] = ap[s < SP ? s : SP - 1];                       // (no product column there: B is zero, any finite A will do)
                    }
                };
                auto b_group = [&](decltype(bp0) bp, int g) -> f32x4 { const v4f t = bp[(g < c16 ? g : c16 - 1) * 64]; return f32x4{t.x, t.y, t.z, t.w}; };
                f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                auto mfma8 = [&](const float (&a)[4], const f32x4 &b0, const f32x4 &b1) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b1.w, acc1, 0, 0, 0);
                };
                f32x4 be0 = b_group(bp0, 0), be1 = b_group(bp1, 0), bo0 = b_group(bp0, 1), bo1 = b_group(bp1, 1);
                float ae[4], ao[4];
                a_group(0, ae);
#pragma unroll 1
                for (int g = 0; g < c16; g += 2) {
                    a_group(g + 1, ao);
                    mfma8(ae, be0, be1);
                    be0 = b_group(bp0, g + 2); be1 = b_group(bp1, g + 2);
                    a_group(g + 2, ae);
                    if (g + 1 < c16) mfma8(ao, bo0, bo1);
                    bo0 = b_group(bp0, g + 3); bo1 = b_group(bp1, g + 3);
                }
                auto store = [&](const f32x4 &acc, int cb) {                  // rows lk*4 + r of the token block, column lr
                    const float av[4] = {acc.x, acc.y, acc.z, acc.w};
                    const int col = cb * 16 + lr;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int tok = tb * 16 + lk * 4 + r;
                        float v = av[r] + 0.0f;                               // -0.0 -> +0.0 like score_tile_kernel
                        if (col == clamp_col) v = fminf(v, p.threshold);
                        if (tok < n && col < Kp) scl[(long long)tok * Kp + col] = v;
                    }
                };
                store(acc0, cb0);
                if (cb1 != cb0) store(acc1, cb1);
            }
            };
            if (otm_lds) product_lds();
            else product((glb_cv4f *)p.OTm + lane);                             // (the image does not fit the LDS: B fragments from L2)
        }
        if (probe) pw = (long long)__builtin_amdgcn_s_memtime();                // (wavefront 0's own units are done)
        __syncthreads();                                 // the products (aliasing hist) are free again
        if (probe) pb = (long long)__builtin_amdgcn_s_memtime();
    }
    if (tid >= nthreads) return;                         // the score phase's extra wavefronts leave (behind its last barrier)
    int my_tr = 0;
    {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)trl);
        for (int k = wu; k < tr_pieces; k += nwaves, my_tr++)
            lds_dma16((unsigned)k * 1024u + (unsigned)lane * 16u, reinterpret_cast<const char *>(p.trT), lds0 + (unsigned)k * 1024u);
    }
    if (PW > K)
        for (int i = tid; i < n * (PW - K); i += nthreads) hist[(i / (PW - K)) * PW + K + i % (PW - K)] = ninf;
    wait_vmcnt(my_tr);                                   // scores + this lane's transition entries landed
    wg_barrier_lds();                                    // (a __syncthreads would drain the table DMA)
    // scores of this lane's two tags (a tag beyond K reads a finite pad column and its transitions are -inf)
    const float *fcol = scl + (own0 ? j0 : 0);
    auto scores_at = [&](int t) {                        // f_t[j0], f_t[j0+1]; nothing is read into a half without a tag
        v2f f = *reinterpret_cast<const v2f *>(fcol + (size_t)t * Kp);
        f.y = own1 ? f.y : 0.0f;
        return f;
    };
    if (writer) {                                                             // crf.py:135
        const v2f f0 = scores_at(0);
        hist[j0] = f0.x + t_start.x;
        if (own1) hist[j0 + 1] = f0.y + t_start.y;
    }
    wg_barrier_lds();
    if (probe) pc1 = (long long)__builtin_amdgcn_s_memtime();
    // the tail wavefront's step is a short latency chain (LDS read, a few adds, six DPP levels, LDS write): at the default
    // priority the SIMD's older wavefronts starve it until their own issue-bound step is over and the chain then runs
    // behind them (K = 130: +330 cycles per step); first in line, it hides inside their step
    if (tail) __builtin_amdgcn_s_setprio(3);
    // One step: two adds (crf.py:123,145) and half a v_max3 per (source, tag).  The step is VALU-issue bound on the CU:
    // hoisting the partition-independent add (f + tr) behind the step's LDS write was measured and bought nothing (the
    // idle time around the write and the barrier is ~100 cycles, not the ~350 the first probe suggested), and
    // v_pk_add_f32 issues at half rate on gfx950, so the packed form saves registers and LDS reads, not issue slots.
    // Two loops, the full wavefronts' specialised on its number of leftover slots: with the tail / leftover / ablation tests
    // inside one loop a step spent ~250 cycles on a dozen scalar branches (a taken branch refills the instruction buffer).
    const unsigned pin_lane = (unsigned)(size_t)(hist + g * 4);               // this lane's 16 bytes of a 32-source block
    const unsigned f_lane = (unsigned)(size_t)fcol;
    unsigned px_lane[4];
#pragma unroll
    for (int xk = 0; xk < 4; xk++) px_lane[xk] = (unsigned)(size_t)(hist + ixs[xk]);
    auto publish = [&](int t, const v2f &best) {
        if (writer) *reinterpret_cast<v2f *>(hist + (size_t)t * PW + j0) = best;   // (j0 + 1 == K: -inf into the pad)
        wg_barrier_lds();
    };
    auto full_steps = [&](auto xs_c) {
        constexpr int XSC = decltype(xs_c)::value;       // leftover source slots per lane (0..4)
        v2f fnext = n > 1 ? scores_at(1) : v2f{0.f, 0.f};
        for (int t = 1; t < n; t++) {
            const v2f f = fnext;
            auto f_tr = [&](int sl) { return f + trs[sl]; };
            v2f best = v2f{ninf, ninf};
            // every LDS read of the step up front, oldest first: the scores of the NEXT step, the IB4 blocks, the leftovers
            // (left to the compiler they went through one recycled register quad: three exposed LDS round trips per step)
            const unsigned row = (unsigned)((t - 1) * PW) * 4u;
            const unsigned frow = (unsigned)((t + 1 < n ? t + 1 : t) * Kp) * 4u;
            v4f p4[IB4 > 0 ? IB4 : 1];
            float px[XSC > 0 ? XSC : 1];
            asm volatile("ds_read_b64 %0, %1" : "=v"(fnext) : "v"(f_lane + frow));
            if constexpr (IB4 > 0) lds_read16_at<0>(p4[0], pin_lane + row);
            if constexpr (IB4 > 1) lds_read16_at<128>(p4[1], pin_lane + row);
            if constexpr (IB4 > 2) lds_read16_at<256>(p4[2], pin_lane + row);
            if constexpr (IB4 > 3) lds_read16_at<384>(p4[3], pin_lane + row);
            if constexpr (IB4 > 4) lds_read16_at<512>(p4[4], pin_lane + row);
            if constexpr (IB4 > 5) lds_read16_at<640>(p4[5], pin_lane + row);
#pragma unroll
            for (int xk = 0; xk < XSC; xk++) asm volatile("ds_read_b32 %0, %1" : "=v"(px[xk]) : "v"(px_lane[xk] + row));
            // (the partition of ONE source added to both tags of the lane: the packed add reads the same half of the register pair for
            //  both results -- op_sel -- where the compiler built {p, p} pairs with a v_mov per block; the first block starts `best`
            //  from its own values instead of -inf: round 6, late -- the step is VALU-issue bound over the compute unit's nine wavefronts)
            auto add_lo = [](v2f a, v2f pr) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(pr)); return r; };
            auto add_hi = [](v2f a, v2f pr) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(pr)); return r; };
            auto block = [&](int k4) {
                const v2f xy = v2f{p4[k4].x, p4[k4].y}, zw = v2f{p4[k4].z, p4[k4].w};
                const v2f v0 = add_lo(f_tr(k4 * 4 + 0), xy), v1 = add_hi(f_tr(k4 * 4 + 1), xy);
                const v2f v2 = add_lo(f_tr(k4 * 4 + 2), zw), v3 = add_hi(f_tr(k4 * 4 + 3), zw);
                if (k4 == 0) {                           // (max over the same values: the order of a max does not change its result)
                    best.x = fmaxf(fmaxf(v0.x, v1.x), v2.x); best.y = fmaxf(fmaxf(v0.y, v1.y), v2.y);
                    best.x = fmaxf(best.x, v3.x); best.y = fmaxf(best.y, v3.y);
                } else {
                    best.x = fmaxf(fmaxf(best.x, v0.x), v1.x); best.y = fmaxf(fmaxf(best.y, v0.y), v1.y);
                    best.x = fmaxf(fmaxf(best.x, v2.x), v3.x); best.y = fmaxf(fmaxf(best.y, v2.y), v3.y);
                }
            };
            if constexpr (IB4 > 0) { lds_wait_for<IB4 + XSC - 1>(p4[0]); block(0); }
            if constexpr (IB4 > 1) { lds_wait_for<IB4 + XSC - 2>(p4[1]); block(1); }
            if constexpr (IB4 > 2) { lds_wait_for<IB4 + XSC - 3>(p4[2]); block(2); }
            if constexpr (IB4 > 3) { lds_wait_for<IB4 + XSC - 4>(p4[3]); block(3); }
            if constexpr (IB4 > 4) { lds_wait_for<IB4 + XSC - 5>(p4[4]); block(4); }
            if constexpr (IB4 > 5) { lds_wait_for<IB4 + XSC - 6>(p4[5]); block(5); }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fnext));
#pragma unroll
            for (int xk = 0; xk < XSC; xk++) {
                asm volatile("" : "+v"(px[xk]));         // (behind the wait)
                const v2f v = f_tr(IB + xk) + v2f{px[xk], px[xk]};
                best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y);
            }
            fnext.y = own1 ? fnext.y : 0.0f;             // (scores_at's rule: nothing is read into a half without a tag)
            // the eight lanes of the group: xor 1, xor 2 inside the quad, then the mirrored quad of the half row
            asm volatile("s_nop 1\n\t"
                         "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                         "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1"
                         : "+v"(best.x), "+v"(best.y));
            publish(t, best);
        }
    };
    auto tail_steps = [&]() {
        v2f fnext = n > 1 ? scores_at(1) : v2f{0.f, 0.f};
        for (int t = 1; t < n; t++) {
            const float *pin = hist + (size_t)(t - 1) * PW;
            const v2f f = fnext;
            fnext = scores_at(t + 1 < n ? t + 1 : t);    // (the coming step's scores: fetched beside this step's partitions)
            auto f_tr = [&](int sl) { return f + trs[sl]; };
            v2f best = v2f{ninf, ninf};
#pragma unroll
            for (int s4 = 0; s4 < NSL; s4 += 4) {        // four slots at a time (slots beyond nst: -inf transitions)
                if (s4 >= nst) continue;
                float ps[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = g + GT * (s4 + u); ps[u] = pin[i < PW ? i : PW - 1]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const v2f v = f_tr(s4 + u) + v2f{ps[u], ps[u]};
                    best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y);
                }
            }
            // max scan over the GT lanes of the group: its last lane ends up with the group's maximum
#define FARNN_SCAN2(CTRL) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %1, %1, %1 " CTRL "\n\ts_nop 1" : "+v"(best.x), "+v"(best.y))
            FARNN_SCAN2("row_shr:1 row_mask:0xf bank_mask:0xf");
            FARNN_SCAN2("row_shr:2 row_mask:0xf bank_mask:0xf");
            FARNN_SCAN2("row_shr:4 row_mask:0xf bank_mask:0xf");
            if (GT >= 16) FARNN_SCAN2("row_shr:8 row_mask:0xf bank_mask:0xf");
            if (GT >= 32) FARNN_SCAN2("row_bcast:15 row_mask:0xa bank_mask:0xf");
            if (GT >= 64) FARNN_SCAN2("row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef FARNN_SCAN2
            publish(t, best);
        }
    };
    if (tail) tail_steps();
    else switch (XS) {
        case 0: full_steps(std::integral_constant<int, 0>{}); break;
        case 1: full_steps(std::integral_constant<int, 1>{}); break;
        case 2: full_steps(std::integral_constant<int, 2>{}); break;
        case 3: full_steps(std::integral_constant<int, 3>{}); break;
        default: full_steps(std::integral_constant<int, 4>{}); break;
    }
    if (tail) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the transition table is in LDS
    __syncthreads();
    if (probe) pc2 = (long long)__builtin_amdgcn_s_memtime();
    if (w == 0 && n > 0) {
        // One wavefront walks the path, and a lone wavefront issues one instruction every ~4-5 cycles: a step is bound by its
        // instruction COUNT plus one LDS round trip (r02: eight 4-byte reads, a six-level DPP max, three ballots -- 590 cycles;
        // rounds 3-5: four consecutive candidates per lane, 16-byte reads, no reduction -- 437).
        // Round 6: candidate i of a step sits on lane i % 64 of slice i / 64 (NU <= 4 slices of a row).  The comparison of a slice
        // against the step's maximum IS its ballot (v_cmp_eq writes the 64-lane mask to a scalar pair), and the first arg-max of
        // the row (torch.max's rule, crf.py:147-149) is the lowest set bit of the first non-empty slice: s_ff1 on scalars -- no
        // per-lane "first hit of my four", no readlane, half the vector instructions.  There is NO reduction: the maximum of a
        // step's candidates IS part_t[ptr], which the forward pass took over exactly these values (a max returns one of its
        // operands' bits) -- two broadcast reads fetch it and the score f_t[ptr] beside the table's row.
        constexpr int NU = (32 * IB4 + 32 + 63) / 64;                         // PW <= 32 IB4 + 32
        unsigned hq_a[NU], tq_a[NU];                                          // byte addresses of this lane's entries of a row
        unsigned long long valid[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int i = lane + 64 * u;
            const bool ok = i < PW;                                           // (entries in [K, PW) are -inf through the partitions' pads)
            valid[u] = __ballot(ok);
            hq_a[u] = (unsigned)(size_t)(hist + (ok ? i : PW - 1));
            tq_a[u] = (unsigned)(size_t)(trl + (ok ? i : PW - 1));
        }
        auto first_equal = [&](const float (&c)[NU], float m) {               // first index (torch.max's rule); 0 if none (NaN)
            int r = 0;
#pragma unroll
            for (int u = NU - 1; u >= 0; u--) {
                const unsigned long long hit = __ballot(c[u] == m) & valid[u];
                r = hit ? 64 * u + (int)__builtin_ctzll(hit) : r;
            }
            return __builtin_amdgcn_readfirstlane(r);
        };
        auto lds_f = [](unsigned a) { return *(lds_cfloat *)(size_t)a; };
        float prv[NU], c[NU];
        float mx = ninf;
#pragma unroll
        for (int u = 0; u < NU; u++) {                                        // crf.py:168-169: part_{n-1}[i] + tr[i][STOP]  (0 + x = x)
            c[u] = (0.0f + lds_f(tq_a[u] + 4u * (unsigned)(STOP * Kp))) + lds_f(hq_a[u] + 4u * (unsigned)((n - 1) * PW));
            mx = fmaxf(mx, c[u]);
        }
        int ptr = first_equal(c, wave_max_dpp(mx));
#pragma unroll
        for (int u = 0; u < NU; u++) prv[u] = lds_f(hq_a[u] + 4u * (unsigned)((n > 1 ? n - 2 : 0) * PW));   // part_{t-1} of the first step
        const unsigned h0 = (unsigned)(size_t)hist;
        const unsigned sc_off = (unsigned)(size_t)scl - h0;
        int mytag = 0;                                   // lane t % 64 keeps the tag of position t until its block of 64 is flushed
        auto keep_tag = [&](int t) {                     // position t's tag (model_decompose.py:356) into lane t % 64: a v_writelane through M0
            const int tag = (ptr == K - 3) ? p.o_idx : ptr;
            const int tl = __builtin_amdgcn_readfirstlane(t & 63), tg = __builtin_amdgcn_readfirstlane(tag);
            unsigned keep;                               // (one SGPR per instruction: the lane select rides in M0)
            asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                         : "+v"(mytag), "=&s"(keep) : "s"(tg), "s"(tl));
        };
        // Blocks of 64 positions, last block first.  The loop over a block's positions has NO test inside (round 6: the flush test, the
        // t == 0 test and the loop's own layout were three taken branches per position, each a refill of the instruction buffer --
        // ~60 of a step's ~400 cycles): position 0 is peeled off (it has no back-pointer to find), a block is flushed behind its loop.
        // (round 6, late: one lone wavefront issues an instruction per ~5 cycles, so a step costs its instruction COUNT.  The NU slices of a
        //  row are read off ONE address with immediate offsets -- lane + 64 u lies 256 u bytes further; a lane past the row's end reads
        //  whatever follows, and `valid` drops its candidate -- instead of NU clamped addresses per row; and two steps per trip hand the
        //  row fetched ahead to each other in place, instead of NU register copies per step.)
        const unsigned hq0 = (unsigned)(size_t)(hist + lane), tq0 = (unsigned)(size_t)(trl + lane);
        auto read_slices = [&](float (&dst)[NU], unsigned a) {
            asm volatile("ds_read_b32 %0, %1" : "=v"(dst[0]) : "v"(a) : "memory");
            if constexpr (NU > 1) asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(dst[1]) : "v"(a) : "memory");
            if constexpr (NU > 2) asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(dst[2]) : "v"(a) : "memory");
            if constexpr (NU > 3) asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(dst[3]) : "v"(a) : "memory");
        };
        auto step = [&](int t, const float (&pin)[NU], float (&pout)[NU]) {
            // the back-pointer of step t at tag ptr (crf.py:147-149), recomputed: same f32 expression, same values.  The step's LDS
            // reads are ISSUED first; the tag's bookkeeping runs under their round trip -- in front of them it sat on the chain
            // ptr -> addresses -> reads -> ptr (round 4: -70 cycles per step)
            const int tp = __builtin_amdgcn_readfirstlane(t > 1 ? t - 2 : 0);
            const unsigned o_pre = 4u * (unsigned)(tp * PW), o_tr = 4u * (unsigned)(ptr * Kp);
            const unsigned a_m = h0 + 4u * (unsigned)(t * PW + ptr), a_f = a_m + sc_off;         // (PW == Kp)
            float tr[NU], m, f;
            // the row fetched ahead first (its latency hides behind the others' -- LDS reads return in order), then what the step
            // waits for; volatile statements keep this order
            read_slices(pout, hq0 + o_pre);
            read_slices(tr, tq0 + o_tr);
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=&v"(m), "=&v"(f) : "v"(a_m), "v"(a_f) : "memory");
            keep_tag(t);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m), "+v"(f) : : "memory");
#pragma unroll
            for (int u = 0; u < NU; u++) asm volatile("" : "+v"(pout[u]), "+v"(tr[u]));              // (behind the wait)
#pragma unroll
            for (int u = 0; u < NU; u++) c[u] = (f + tr[u]) + pin[u];         // (feat + trans) + partition: crf.py:123,145
            ptr = first_equal(c, m);
        };
        float pnx[NU];
        // Blocks of 64 positions, last block first.  The loop over a block's positions has NO test inside (round 6: the flush test, the
        // t == 0 test and the loop's own layout were three taken branches per position, each a refill of the instruction buffer --
        // ~60 of a step's ~400 cycles): position 0 is peeled off (it has no back-pointer to find), a block is flushed behind its loop.
        for (int t_hi = n - 1; t_hi >= 0; t_hi = (t_hi & ~63) - 1) {
            const int t_lo = t_hi & ~63, t_stop = t_lo > 0 ? t_lo : 1;
            int t = t_hi;
#pragma unroll 1
            for (; t > t_stop; t -= 2) { step(t, prv, pnx); step(t - 1, pnx, prv); }
            if (t == t_stop) {
                step(t, prv, pnx);
#pragma unroll
                for (int u = 0; u < NU; u++) prv[u] = pnx[u];
            }
            if (t_lo == 0) keep_tag(0);
            if (t_lo + lane <= t_hi) {                   // the block's 64 positions at a time, coalesced
                if (p.tags) p.tags[(long long)b * p.L + t_lo + lane] = mytag;
                if (p.flat) p.flat[foff + t_lo + lane] = mytag;
            }
        }
    }
    if (probe && tid == 0) {
        pc3 = (long long)__builtin_amdgcn_s_memtime();
        printf("viterbi wg %d (%d positions, %d threads): set-up + scores %lld cycles (opening loads landed at %lld, products / label-map scores staged at %lld, wavefront 0's first rows landed or its units done at %lld, scores done at %lld), forward pass %lld (%lld per step), backtrace %lld (%lld per step)\n",
               b, n, nthreads, pc1 - pc0, pf0 - pc0, pa - pc0, (pw ? pw : pa) - pc0, pb - pc0, pc2 - pc1, (pc2 - pc1) / (n > 1 ? n - 1 : 1), pc3 - pc2, (pc3 - pc2) / n);
    }
    if (p.tags)
        for (int i = n + tid; i < p.L; i += nthreads) p.tags[(long long)b * p.L + i] = -1;   // pads (LOCAL and FULL)
}

// blockDim.x = the score phase's threads (viterbi_hist_score_threads: up to sixteen wavefronts for the fused form); the dynamic
// programme runs on the first viterbi_hist_threads(K) of them
template <int IB4, bool FUSED>
__global__ void __launch_bounds__(1024)
viterbi_hist_kernel(const ScoreParams p) {
    extern __shared__ __align__(16) float smem[];
    const int dp_threads = viterbi_dp_threads(p.K);                         // = viterbi_hist_threads(K)
    viterbi_hist_body<IB4, FUSED>(p, smem, (int)threadIdx.x, dp_threads, (int)blockIdx.x, nullptr, nullptr, false, nullptr, nullptr,
                                  (int)blockDim.x);
}

// viterbi_hist_kernel: contiguous float4 blocks per lane (K = 32*IB4 + leftovers) and its thread count
inline int viterbi_hist_ib4(int K) { return K / 32; }
inline int viterbi_hist_threads(int K) { return viterbi_dp_threads(K); }
// the stand-alone kernel's block: the fused form computes its scores on sixteen wavefronts (four per SIMD: every matrix core, more
// state rows in flight), of which all but the dynamic programme's leave behind the scores
inline int viterbi_hist_score_threads(int K, bool fused) { const int t = viterbi_hist_threads(K); return fused && t < 1024 ? 1024 : t; }
inline size_t viterbi_hist_lds_bytes(int K, int Kp, int SP, int L, bool fused) {
    return viterbi_hist_floats(Kp, SP, L, fused) * 4 + ((size_t)L * Kp * 4 + 1023) / 1024 * 1024 +
           (size_t)viterbi_table_pieces(K, Kp, SP, L, fused) * 1024;
}

}  // namespace farnn
