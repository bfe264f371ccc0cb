// K1d -- the compute wavefronts of chain_regs_body with the block split BY DESTINATION (round 5; S <= 72, sum semiring).
//
// Reference loops: model_onehot.py:372-403.  Per step  out[j] = nl( o[j] * sum_i in[i] * Tf[x][i][j] )  (forward) and
// out[i] = nl( sum_j Tf[x][i][j] * (o[j] * in[j]) )  (backward).
//
// K1r (chain_regs.hip.h) splits a block by SOURCE rows: a lane multiplies its four state entries into four rows of the block,
// every wavefront ends up with partial sums for ALL outputs, and a step needs a partial-sum store, a flag, five partial reads, an
// add tree and four ds_bpermute to get the new state back to the lanes that multiply with it: twelve LDS instructions on the
// step's dependent chain (970 cycles per step; a wavefront pays 35-65 cycles per LDS instruction it issues).
//
// Here a wavefront owns whole OUTPUT entries.  The forward chain reads the TRANSPOSED copy of the block the library keeps for the
// backward chain (layout.hip.h: Mb[x][j][i] = Tf[x][i][j]) and the backward chain the plain one, so in both directions
//     out[r] = sum_i Blk[r][i] * in[i]
// is a dot product along a contiguous row.  Wavefront w owns rows w RW .. w RW + RW - 1 (RW = ceil(S / 6) <= 12), four lanes per
// row; lane (r, q) holds the 16-byte chunks q, q + 4, .. q + 16 of its row (five global_load_dwordx4 per step, straight into a
// register ring D steps deep -- a quad reads 64 contiguous bytes per instruction) and multiplies them with the same chunks of the
// state: 20 FMAs into four accumulators, three adds, the quad joined on the DPP network, scale / non-linearity.  Nothing is
// partial: the row's lane 0 stores the new entry, and the only cross-wavefront traffic of a step is
//     ONE ds_write_b32   the new entry into the exchange row (the wavefront's step flag rides in lane 63 of the same instruction)
//   + ONE ds_write_b32   the entry into `hist` (the whole sequence's states, what the writer / scorer wavefronts and the tiles read)
//   + ONE batch of reads: the six step flags, then this lane's five chunks of the exchange row (the LDS serves a wavefront's
//     operations in order, so chunks read behind flags that say "written" are the written ones; the batch again when not)
// -- eight LDS instructions instead of twelve, no partial sums, no add tree, no ds_bpermute.
//
// The exchange row: xch[2][RD_XS] floats, double-buffered by the step's parity, RD_XS = 80 >= 5 x 16 with zeros behind S (a lane's
// chunks beyond the row multiply an exact zero with whatever finite bytes the block holds there).  Buffer t & 1 holds the state
// step t multiplies with (for the backward chain: scaled by o, model_onehot.py:393).  No hazard on the two buffers: a wavefront
// overwrites buffer t & 1 in step t + 1, after it has seen every partner's step-t flag -- and a partner stores that flag behind
// its own reads of buffer t & 1 (the store's data depends on them).
// Flags: wavefront w's word at flags[4 w] (buffer 0 of chain_regs_body's flag area: regs_flag_newest / regs_rows_reached work
// unchanged) = 1 after the set-up, t + 2 once its entries of state row t + 1 are in `hist` AND in the exchange row.
#pragma once
#include "common.hip.h"
#include "chain_regs_params.hip.h"

namespace farnn {

// What a compute wavefront carries from its priming (before the workgroup's set-up barrier) into the step loop
template <int D, int LPR>
struct DestRing {
    static constexpr int NC = 20 / LPR;
    v4f r[D][NC];                // the register ring: D steps x NC chunks
    unsigned voff[NC];           // this lane's chunk offsets inside a block
    int tkw_lo, tkw_hi;          // byte offsets of 64 steps' blocks (lane l: step window + l)
};

// this lane's place in the split: LPR lanes per output row (see regs_compute_dest)
template <int LPR>
__device__ __forceinline__ void dest_lane(const int lane, int &rj, int &rs, bool &has_row) {
    if constexpr (LPR == 4) { rj = lane >> 2; rs = lane & 3; has_row = true; }
    else { const int r16 = lane & 15, bq = r16 / 5; rs = r16 - 5 * bq; rj = 3 * (lane >> 4) + bq; has_row = bq < 3; }
}

// (cache-policy bits of the ring's loads: an experiment switch of the build, empty in the shipped library -- DESIGN.md, K1d)
#ifndef FARNN_RD_CPOL
#define FARNN_RD_CPOL ""
#endif
#define FARNN_RD_ISSUE_(ring_, d, base_)                                                       \
    do {                                                                                       \
        const char *bp_ = (base_);                                                             \
        if constexpr (NC == 5)                                                                 \
            asm volatile("s_nop 4\n\t"                                                         \
                         "global_load_dwordx4 %0, %5, %10" FARNN_RD_CPOL "\n\t"                                 \
                         "global_load_dwordx4 %1, %6, %10" FARNN_RD_CPOL "\n\t"                                 \
                         "global_load_dwordx4 %2, %7, %10" FARNN_RD_CPOL "\n\t"                                 \
                         "global_load_dwordx4 %3, %8, %10" FARNN_RD_CPOL "\n\t"                                 \
                         "global_load_dwordx4 %4, %9, %10" FARNN_RD_CPOL                                     \
                         : "=&v"(ring_.r[d][0]), "=&v"(ring_.r[d][1]), "=&v"(ring_.r[d][2]), "=&v"(ring_.r[d][3]), "=&v"(ring_.r[d][NC - 1]) \
                         : "v"(ring_.voff[0]), "v"(ring_.voff[1]), "v"(ring_.voff[2]), "v"(ring_.voff[3]), "v"(ring_.voff[NC - 1]), "s"(bp_)); \
        else                                                                                   \
            asm volatile("s_nop 4\n\t"                                                         \
                         "global_load_dwordx4 %0, %4, %8" FARNN_RD_CPOL "\n\t"                                  \
                         "global_load_dwordx4 %1, %5, %8" FARNN_RD_CPOL "\n\t"                                  \
                         "global_load_dwordx4 %2, %6, %8" FARNN_RD_CPOL "\n\t"                                  \
                         "global_load_dwordx4 %3, %7, %8" FARNN_RD_CPOL                                      \
                         : "=&v"(ring_.r[d][0]), "=&v"(ring_.r[d][1]), "=&v"(ring_.r[d][2]), "=&v"(ring_.r[d][3]) \
                         : "v"(ring_.voff[0]), "v"(ring_.voff[1]), "v"(ring_.voff[2]), "v"(ring_.voff[3]), "s"(bp_)); \
    } while (0)

// PRIMING, before the workgroup's set-up barrier (round 5, the review's "hide the set-up under the first block loads"): the block
// offsets of the first 64 steps straight from the token ids in global memory (the set-up writes them to LDS for the later windows
// and for nobody's first loads), and the first D steps' pieces requested -- an L2 / Infinity-Cache round trip at the launch's start,
// when all 512 workgroups ask at once, that used to begin only behind the barrier.
template <int D, int LPR>
__device__ __forceinline__ void regs_dest_prime(const RegsParams &p, const int dir, const int w, const int lane, const int nsteps,
                                                const int len, const int b, DestRing<D, LPR> &ring) {
    constexpr int NC = DestRing<D, LPR>::NC;
    const int S = p.S, SP = p.SP, CPR = p.CPR;
    const int RW = (S + RG_NWC - 1) / RG_NWC;
    int rj, rs;
    bool has_row;
    dest_lane<LPR>(lane, rj, rs, has_row);
    const int my_row = w * RW + rj;
    const bool my_valid = has_row && rj < RW && my_row < S;
#pragma unroll
    for (int i = 0; i < NC; i++) {
        const int ci = rs + LPR * i;
        // a chunk beyond the row (or a lane without a row) loads the lane's first chunk again -- the same line, no traffic;
        // its state chunk is zeros
        ring.voff[i] = ((unsigned)(my_valid ? my_row : 0) * (unsigned)SP + (unsigned)((my_valid && ci < CPR) ? ci : rs < CPR ? rs : 0) * 4u) * 4u;
    }
    {
        const int t = lane < nsteps ? lane : nsteps - 1;
        const int idx = (dir == 0) ? t : (t < len ? len - 1 - t : t);
        const long long o_ = (long long)clamp_tok(p.x[(long long)b * p.L + idx], p.V) * p.blk * 4;
        ring.tkw_lo = (int)(unsigned)o_; ring.tkw_hi = (int)(unsigned)(o_ >> 32);
    }
    const char *Mbase = reinterpret_cast<const char *>(dir == 0 ? p.Mb : p.Mf);      // rows = outputs
#pragma unroll
    for (int d = 0; d < D; d++) {
#pragma unroll
        for (int u = 0; u < NC; u++) ring.r[d][u] = v4f{0.f, 0.f, 0.f, 0.f};
        if (d < nsteps) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane(ring.tkw_lo, d), hi = (unsigned)__builtin_amdgcn_readlane(ring.tkw_hi, d);
            FARNN_RD_ISSUE_(ring, d, Mbase + (((long long)hi << 32) | lo));
        }
    }
}

template <bool NLX, int D, int LPR>
__device__ __forceinline__ void regs_compute_dest(const RegsParams &p, const int dir, const int w, const int lane_in, const int nsteps,
                                                  const long long *tokoff, float *flags, const float *ol, float *hist, float *xd,
                                                  const bool probe, const int b, DestRing<D, LPR> &ring) {
    static_assert(D == 2 || D == 4, "an even ring depth that divides 64 (the exchange row's parity and the address window)");
    static_assert(LPR == 4 || LPR == 5, "four or five lanes per output row");
    // LPR lanes per output row, NC chunks of the row per lane and step (LPR x NC = 20 chunks = RD_XS floats >= the padded row):
    //   LPR = 4: lane = 4 r + q, the quad joined by two DPP adds; five chunks -- 48 of 64 lanes carry 5 loads each (69 % of the
    //            load instructions' lanes fetch bytes of the block);
    //   LPR = 5: three rows per DPP row of sixteen lanes (lane = 16 a + 5 b + q, lane 15 of each DPP row idle), the five lanes joined
    //            by three DPP adds (row_shr 1, 2 and 4), four chunks -- 60 of 64 lanes carry 4 loads each (88 %): what the texture
    //            addresser handles per step is what bounds two workgroups on a compute unit, and this form asks for a fifth less.
    constexpr int NC = 20 / LPR;                         // 16-byte chunks per lane and step
    static_assert(LPR * NC * 4 == RD_XS, "the exchange row holds every lane's chunks");
    int lane = lane_in;
    const int S = p.S, SP = p.SP, CPR = p.CPR;
    const int RW = (S + RG_NWC - 1) / RG_NWC;            // outputs per wavefront (<= 12)
    int rj, rs;
    bool has_row;
    dest_lane<LPR>(lane, rj, rs, has_row);
    const int my_row = w * RW + rj;
    const bool my_valid = has_row && rj < RW && my_row < S;
    const bool my_writer = my_valid && rs == (LPR == 4 ? 0 : 4);      // the lane the row's sum ends up in
    const bool is_flane = lane == 63;                    // (never a row's lane: RW <= 12 rows take lanes 0 .. 47 / the lanes != 15 mod 16)
    (void)CPR;
    const char *Mbase = reinterpret_cast<const char *>(dir == 0 ? p.Mb : p.Mf);      // rows = outputs
    const float my_o = my_valid ? ol[my_row] : 1.0f;
    const float c_pre = dir == 0 ? my_o : 1.0f, c_post = dir == 0 ? 1.0f : my_o;     // (:377-386) / (:393-402)
    float *xbuf = xd;                                    // [2][RD_XS]
    float *dumpa = xd + 2 * RD_XS, *dumpb = dumpa + 64;
    // where this lane's results go: `hist` row t + 1 (advancing) and the exchange row of parity (t + 1) & 1; lanes without an
    // output store to dump slots, lane 63 stores the wavefront's flag with the exchange store
    float *hptr = my_writer ? hist + SP + my_row : dumpa + lane;
    const int hstep = my_writer ? SP : 0;
    float *xw0 = my_writer ? xbuf + my_row : (is_flane ? flags + 4 * w : dumpb + lane);              // into buffer 0
    float *xw1 = my_writer ? xbuf + RD_XS + my_row : (is_flane ? flags + 4 * w : dumpb + lane);      // into buffer 1
    const float *xr = xbuf + 4 * rs;                     // this lane's chunks: xr + 4 LPR i (+ RD_XS for the odd buffer)
    const int nl_mode = p.nl;
    const bool nl_relu = nl_mode == FARNN_NL_RELU;
    const int *pflag = reinterpret_cast<const int *>(flags) + 4 * (lane < RG_NWC ? lane : 0);

    v4f (&r)[D][NC] = ring.r;
    int &tkw_lo = ring.tkw_lo, &tkw_hi = ring.tkw_hi;
#define FARNN_RD_WINDOW(t_)                                                                    \
    do {                                                                                       \
        const int ti_ = (t_) + lane;                                                           \
        const long long o_ = tokoff[ti_ < nsteps ? ti_ : nsteps - 1];                          \
        tkw_lo = (int)(unsigned)o_; tkw_hi = (int)(unsigned)(o_ >> 32);                        \
    } while (0)
#define FARNN_RD_BASE(t_, lo_, hi_)                                                            \
    do {                                                                                       \
        const int li_ = (t_) & 63;                                                             \
        lo_ = (unsigned)__builtin_amdgcn_readlane(tkw_lo, li_);                                \
        hi_ = (unsigned)__builtin_amdgcn_readlane(tkw_hi, li_);                                \
    } while (0)
#define FARNN_RD_ISSUE(d, lo_, hi_) FARNN_RD_ISSUE_(ring, d, Mbase + (((long long)(hi_) << 32) | (lo_)))
    // ONE wait statement per step (chain_regs.hip.h says why): the D - 1 younger steps' pieces may stay outstanding, the last
    // D - 1 steps of a sequence drain
#define FARNN_RD_WAITSTR                                                                       \
    "s_cmp_ge_i32 %[rem], %[dm1]\n\t"                                                          \
    "s_cbranch_scc1 1f\n\t"                                                                    \
    "s_waitcnt vmcnt(0)\n\t"                                                                   \
    "s_branch 2f\n"                                                                            \
    "1:\n\t"                                                                                   \
    "s_waitcnt vmcnt(%[cnt])\n"                                                                \
    "2:"
#define FARNN_RD_WAIT(d, rem_)                                                                 \
    do {                                                                                       \
        if constexpr (NC == 5)                                                                 \
            asm volatile(FARNN_RD_WAITSTR : "+v"(r[d][0]), "+v"(r[d][1]), "+v"(r[d][2]), "+v"(r[d][3]), "+v"(r[d][NC - 1]) \
                         : [rem] "s"(rem_), [dm1] "n"(D - 1), [cnt] "n"((D - 1) * NC) : "scc"); \
        else                                                                                   \
            asm volatile(FARNN_RD_WAITSTR : "+v"(r[d][0]), "+v"(r[d][1]), "+v"(r[d][2]), "+v"(r[d][3]) \
                         : [rem] "s"(rem_), [dm1] "n"(D - 1), [cnt] "n"((D - 1) * NC) : "scc"); \
    } while (0)

    unsigned nlo = 0, nhi = 0;              // (the ring is primed: regs_dest_prime, before the set-up barrier)
#if defined(FARNN_PROBES)
    long long ph[4] = {0, 0, 0, 0}, pt = 0;
#define FARNN_RD_PHASE(i) do { if (probe && w == 0 && (p.dbg & 256)) { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); ph[i] += n_ - pt; pt = n_; } } while (0)
    if (probe && w == 0) pt = (long long)__builtin_amdgcn_s_memtime();
#else
#define FARNN_RD_PHASE(i) do { } while (0)
    (void)probe; (void)b;
#endif
    if (D < nsteps) FARNN_RD_BASE(D, nlo, nhi);            // step 0's look-ahead (the window of steps 0 .. 63 is loaded)
    for (int t0 = 0; t0 < nsteps; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int t = t0 + d;
            if (t >= nsteps) break;
            constexpr int dummy = 0; (void)dummy;
            const int roff = (d & 1) * RD_XS;                // the exchange buffer this step reads (t and d have the same parity)
            // the partners' flags FIRST, this lane's five state chunks behind them in the same batch
            v4f st[NC];
            {
                int fl = lds_flag_get(pflag);
#pragma unroll
                for (int i = 0; i < NC; i++) st[i] = *reinterpret_cast<const v4f *>(xr + roff + 4 * LPR * i);
                asm volatile("" ::: "memory");
                if (__ballot(fl < t + 1) != 0ull) {
                    // a partner is late: poll the flags alone (one 4-byte read per round -- re-reading the chunks with every poll
                    // would keep the LDS busy with 16-byte reads of stale data, at the expense of the other workgroup on the
                    // compute unit), then the chunks once more
                    do { fl = lds_flag_get(pflag); } while (__ballot(fl < t + 1) != 0ull);
#pragma unroll
                    for (int i = 0; i < NC; i++) st[i] = *reinterpret_cast<const v4f *>(xr + roff + 4 * LPR * i);
                    asm volatile("" ::: "memory");
                }
            }
            FARNN_RD_PHASE(0);                               // the partners' entries of this step's state
            FARNN_RD_WAIT(d, nsteps - 1 - t);                // steps issued after this one: min(D - 1, nsteps - 1 - t)
            FARNN_RD_PHASE(1);                               // this step's block pieces
            // The four accumulators as TWO register pairs, multiplied pair by pair: v_pk_fma_f32 on the halves of the 16-byte chunks as
            // they lie in the registers.  (Written as four scalar fmaf the compiler packed them too -- but as (x, z) / (y, w), and
            // shuffled the operands of every chunk into that pairing: 28 v_mov_b32 per step beside 8 v_pk_fma, seen in the ISA.
            // The compute wavefronts of a compute unit share four SIMDs: the step is their VALU instruction count.)  Same
            // products into the same accumulators in the same order: bit-identical.
            v2f a01 = v2f{0.f, 0.f}, a23 = v2f{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NC; i++) {
                a01 = __builtin_elementwise_fma(st[i].xy, r[d][i].xy, a01);
                a23 = __builtin_elementwise_fma(st[i].zw, r[d][i].zw, a23);
            }
            float sa = a01.x + a01.y, sb = a23.x + a23.y;
            asm volatile("" : "+v"(sa), "+v"(sb));            // (two scalar adds: packed into one v_pk_add they cost three v_mov)
            float s = sa + sb;
            if constexpr (LPR == 4) s = quad_sum(s);
            else {
                // the five lanes of a row: lane q = 4 ends up with x4 + x3 + x2 + x1 (+ x0): row_shr 1, row_shr 2, row_shr 4 of the
                // ORIGINAL value (the other lanes' sums mix rows and are never used)
                const float s1 = s + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x111, 0xf, 0xf, true));
                const float s2 = s1 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x112, 0xf, 0xf, true));
                s = s2 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x114, 0xf, 0xf, true));
            }
            const float pre = s * c_pre;
            float hn;
            if (NLX) hn = apply_nl(pre, nl_mode);            // tanh, relu-tanh, sigmoid
            else     hn = nl_relu ? fmaxf(pre, 0.0f) : pre;  // none / relu: no branch in the step
            const float hx = is_flane ? __int_as_float(t + 2) : hn * c_post;      // what the next step multiplies with; lane 63: the flag
            asm volatile("" ::: "memory");
            *hptr = hn;                                      // row t + 1 of `hist` (or a dump slot)
            hptr += hstep;
            if (d & 1) *xw0 = hx; else *xw1 = hx;            // the exchange row of parity (t + 1) & 1, the flag behind it in the same store
            asm volatile("" ::: "memory");
            // the slot's registers are dead: the block of step t + D
            if (t + D < nsteps) FARNN_RD_ISSUE(d, nlo, nhi);
            // where the block of step (t + 1) + D is, while the partners finish theirs (chain_regs.hip.h)
            if (d == D - 1 && ((t + 1 + D) & 63) == 0 && t + 1 + D < nsteps) FARNN_RD_WINDOW(t + 1 + D);
            FARNN_RD_BASE(t + 1 + D, nlo, nhi);
            asm volatile("" : "+s"(nlo), "+s"(nhi));
            FARNN_RD_PHASE(2);                               // FMAs, reduce, stores, next loads issued
        }
    }
    // (the last step stored nsteps + 1: writer / scorer count the rows of `hist` by these flags)
#if defined(FARNN_PROBES)
    if (probe && w == 0 && lane == 0 && (p.dbg & 256))
        printf("seq %d dir %d chain phases (destination split), cycles per step: partner wait + state reads %lld, block wait %lld, fma + reduce + stores + issue %lld\n",
               b, dir, ph[0] / nsteps, ph[1] / nsteps, ph[2] / nsteps);
#endif
#undef FARNN_RD_PHASE
#undef FARNN_RD_WAIT
#undef FARNN_RD_WAITSTR
#undef FARNN_RD_ISSUE
#undef FARNN_RD_BASE
#undef FARNN_RD_WINDOW
}

}  // namespace farnn
