// K1r, WIDE form -- the compute wavefronts' part of chain_regs_body for automata of 73..128 states (the reference's SNIPS-BIO and
// ATIS-ZH-BIO automata have 104: RE.py:56-60; the loops are model_onehot.py:372-403 whatever S is).
//
// The narrow form (chain_regs.hip.h, S <= 72) gives a lane at most four rows of 16 bytes per step and keeps four steps in
// flight in 64 VGPRs; two such workgroups share a compute unit.  At S = 104 a row is 26 chunks of 16 bytes, a wavefront holds
// two row groups (52 of 64 lanes), the six compute wavefronts twelve: RPG = 9 rows per lane and step, 36 VGPRs per step in
// flight.  So the wide form runs ONE workgroup per compute unit with the 256-VGPR budget (__launch_bounds__(512, 2)) and a ring
// of D = 4 steps x RQ rows (RQ = 8 / 9 / 11 instantiated: S <= 96 / 108 / 128), and differs from the narrow step in three places:
//   * the row reduce has TWO lanes per row (a wavefront finishes 2 RPG <= 22 rows; four lanes per row would need 88 lanes):
//     six partial sums per lane, one DPP add.  The partial-sum vectors lie PS floats apart with PS = 16 mod 32, so the two
//     lanes of a row -- and the 32 rows a wavefront reads with one instruction -- fall on different banks;
//   * the new state entries reach the lanes that multiply with them next through a 128-byte exchange area of the wavefront
//     (one ds_write_b32 by the finishing lanes, <= three broadcast ds_read_b128 by everybody; the LDS serves a wavefront's
//     operations in order, so the reads see the write): nine to eleven ds_bpermute would cost more than the step's FMAs;
//   * a lane's row slots past RPG (RQ is the instantiated upper bound) re-read its last row -- the same line, no traffic --
//     and multiply it with an exact zero.
// Everything around it (set-up, writer and scorer wavefronts, the hand-off, the end-of-chain tiles) is chain_regs_body's.
#pragma once
#include "common.hip.h"
#include "chain_regs_params.hip.h"

namespace farnn {

// the ring's wait: ONE statement that names every register of the step "+v" (chain_regs.hip.h explains why one)
#define FARNN_RGW_OPS4(d, b) "+v"(r[d][b]), "+v"(r[d][(b) + 1]), "+v"(r[d][(b) + 2]), "+v"(r[d][(b) + 3])
#define FARNN_RGW_WAITSTR                                                                      \
    "s_cmp_ge_i32 %[rem], %[dm1]\n\t"                                                          \
    "s_cbranch_scc1 1f\n\t"                                                                    \
    "s_waitcnt vmcnt(0)\n\t"                                                                   \
    "s_branch 2f\n"                                                                            \
    "1:\n\t"                                                                                   \
    "s_waitcnt vmcnt(%[cnt])\n"                                                                \
    "2:"

template <bool MAXSR, bool NLX, int RQ, int D>
__device__ __forceinline__ void regs_compute_wide(const RegsParams &p, const int dir, const int w, const int lane_in, const int nsteps,
                                                  const long long *tokoff, float *part, const float *ol, float *hist, float *xch) {
    static_assert(RQ == 8 || RQ == 9 || RQ == 11, "the ring's wait statements are written out for 8, 9 and 11 rows per lane");
    static_assert(D == 2 || D == 4, "an even ring depth that divides 64 (the two partial-sum buffers and the address window)");
    constexpr int PSTR = rgw_part_stride(RQ);
    constexpr int G = RGW_G, NP = RGW_NP;
    constexpr int NQ4 = (RQ + 3) / 4;                    // 16-byte reads of the exchanged state entries
    constexpr int NQI = NP / 2;                          // partial sums per reducing lane (two lanes per row)
    int lane = lane_in;
    const int S = p.S, SP = p.SP, PS = p.PS, RPG = p.RPG, CPR = p.CPR;
    int g = lane / CPR;
    const int c = lane - g * CPR;
    const bool active = g < G;
    if (!active) g = 0;                                   // idle lanes shadow group 0 (same lines, results unused)
    const int gid = w * G + g, row0 = gid * RPG;
    unsigned voff[RQ];
    bool okrow[RQ];
#pragma unroll
    for (int u = 0; u < RQ; u++) {
        const int uu = u < RPG ? u : RPG - 1;             // slots past RPG: the lane's last row again
        voff[u] = ((unsigned)(row0 + uu) * (unsigned)SP + (unsigned)c * 4u) * 4u;
        okrow[u] = u < RPG && row0 + u < S;
    }
    const char *Mbase = reinterpret_cast<const char *>(dir == 0 ? p.Mf : p.Mb);
    const int rows_w = G * RPG;
    const int rj = lane >> 1, rs = lane & 1;              // reduce: two lanes per row
    const int my_row = w * rows_w + rj;
    const bool my_valid = rj < rows_w && my_row < S;
    const bool my_writer = my_valid && rs == 0;
    const float my_o = my_valid ? ol[my_row] : 1.0f;
    const float c_pre = dir == 0 ? my_o : 1.0f, c_post = dir == 0 ? 1.0f : my_o;
    const float ninf = -INFINITY;
    const float *ident = part + PSTR - 1;                 // 0.0f / -inf in both buffers (chain_regs_body's set-up)
    float *dump = part + 2 * PSTR;                        // [64][4]
    const float *qptr[NQI];
#pragma unroll
    for (int i = 0; i < NQI; i++) qptr[i] = my_valid ? part + (rs + 2 * i) * PS + my_row : ident;
    const int nact = G * CPR;
    const bool has_flane = nact < 64;
    const bool is_flane = lane == nact;
    float *fslot = part + NP * PS + 4 * w;                // this wavefront's step flag (buffer 0)
    float *wptr = active ? part + gid * PS + c * 4 : (is_flane ? fslot : part + NP * PS + 4 * RG_NWC + (lane - nact) * 4);
    float *hptr = my_writer ? hist + SP + my_row : dump + lane;
    const int hstep = my_writer ? SP : 0;
    const int gq = rj / RPG;
    float *xw = my_writer ? xch + w * RGW_XCH + gq * 12 + (rj - gq * RPG) : dump + 64 + lane;
    const float *xr = xch + w * RGW_XCH + g * 12;
    float hs[4 * NQ4];
#pragma unroll
    for (int u = 0; u < 4 * NQ4; u++) hs[u] = (u < RQ && okrow[u < RQ ? u : 0]) ? hist[row0 + u] * (dir == 1 ? ol[row0 + u] : 1.0f) : 0.0f;
    const int nl_mode = p.nl;
    const bool nl_relu = nl_mode == FARNN_NL_RELU;
    const int *pflag = reinterpret_cast<const int *>(part + NP * PS) + 4 * (lane < RG_NWC ? lane : 0);   // the partners' flags (buffer 0)

    v4f r[D][RQ];
#define FARNN_RGW_WINDOW(t_)                                                                   \
    do {                                                                                       \
        const int ti_ = (t_) + lane;                                                           \
        const long long o_ = tokoff[ti_ < nsteps ? ti_ : nsteps - 1];                          \
        tkw_lo = (int)(unsigned)o_; tkw_hi = (int)(unsigned)(o_ >> 32);                        \
    } while (0)
#define FARNN_RGW_BASE(t_, lo_, hi_)                                                           \
    do {                                                                                       \
        const int li_ = (t_) & 63;                                                             \
        lo_ = (unsigned)__builtin_amdgcn_readlane(tkw_lo, li_);                                \
        hi_ = (unsigned)__builtin_amdgcn_readlane(tkw_hi, li_);                                \
    } while (0)
#define FARNN_RGW_LD4(d, b, pre_)                                                              \
    asm volatile(pre_ "global_load_dwordx4 %0, %4, %8\n\t"                                     \
                 "global_load_dwordx4 %1, %5, %8\n\t"                                          \
                 "global_load_dwordx4 %2, %6, %8\n\t"                                          \
                 "global_load_dwordx4 %3, %7, %8"                                              \
                 : "=&v"(r[d][b]), "=&v"(r[d][(b) + 1]), "=&v"(r[d][(b) + 2]), "=&v"(r[d][(b) + 3]) \
                 : "v"(voff[b]), "v"(voff[(b) + 1]), "v"(voff[(b) + 2]), "v"(voff[(b) + 3]), "s"(bp_))
#define FARNN_RGW_LD1(d, b)                                                                    \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r[d][b]) : "v"(voff[b]), "s"(bp_))
#define FARNN_RGW_ISSUE(d, lo_, hi_)                                                           \
    do {                                                                                       \
        const char *bp_ = Mbase + (((long long)(hi_) << 32) | (lo_));                          \
        FARNN_RGW_LD4(d, 0, "s_nop 4\n\t");                                                    \
        FARNN_RGW_LD4(d, 4, "");                                                               \
        if constexpr (RQ >= 9) FARNN_RGW_LD1(d, 8);                                            \
        if constexpr (RQ >= 10) FARNN_RGW_LD1(d, RQ >= 10 ? 9 : 0);                            \
        if constexpr (RQ >= 11) FARNN_RGW_LD1(d, RQ >= 11 ? 10 : 0);                           \
    } while (0)
    // steady state: the D - 1 younger steps' pieces may stay outstanding; the last D - 1 steps of a sequence drain
#define FARNN_RGW_WAIT(d, rem_)                                                                \
    do {                                                                                       \
        if constexpr (RQ == 8)                                                                 \
            asm volatile(FARNN_RGW_WAITSTR : FARNN_RGW_OPS4(d, 0), FARNN_RGW_OPS4(d, 4)        \
                         : [rem] "s"(rem_), [dm1] "n"(D - 1), [cnt] "n"((D - 1) * RQ) : "scc"); \
        else if constexpr (RQ == 9)                                                            \
            asm volatile(FARNN_RGW_WAITSTR : FARNN_RGW_OPS4(d, 0), FARNN_RGW_OPS4(d, 4), "+v"(r[d][RQ >= 9 ? 8 : 0]) \
                         : [rem] "s"(rem_), [dm1] "n"(D - 1), [cnt] "n"((D - 1) * RQ) : "scc"); \
        else                                                                                   \
            asm volatile(FARNN_RGW_WAITSTR : FARNN_RGW_OPS4(d, 0), FARNN_RGW_OPS4(d, 4), "+v"(r[d][RQ >= 9 ? 8 : 0]), \
                         "+v"(r[d][RQ >= 10 ? 9 : 0]), "+v"(r[d][RQ >= 11 ? 10 : 0])           \
                         : [rem] "s"(rem_), [dm1] "n"(D - 1), [cnt] "n"((D - 1) * RQ) : "scc"); \
    } while (0)

    unsigned nlo = 0, nhi = 0;
    int tkw_lo, tkw_hi;
    FARNN_RGW_WINDOW(0);
#pragma unroll
    for (int d = 0; d < D; d++) {
#pragma unroll
        for (int u = 0; u < RQ; u++) r[d][u] = v4f{0.f, 0.f, 0.f, 0.f};
        if (d < nsteps) {
            FARNN_RGW_BASE(d, nlo, nhi);
            FARNN_RGW_ISSUE(d, nlo, nhi);
        }
    }
    if (D < nsteps) FARNN_RGW_BASE(D, nlo, nhi);           // step 0's look-ahead (the window of steps 0 .. 63 is loaded)
    for (int t0 = 0; t0 < nsteps; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int t = t0 + d;
            if (t >= nsteps) break;
            FARNN_RGW_WAIT(d, nsteps - 1 - t);
            v4f acc = MAXSR ? v4f{ninf, ninf, ninf, ninf} : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < RQ; u++) {
                if (MAXSR) {
                    acc.x = fmaxf(acc.x, okrow[u] ? hs[u] * r[d][u].x : ninf);
                    acc.y = fmaxf(acc.y, okrow[u] ? hs[u] * r[d][u].y : ninf);
                    acc.z = fmaxf(acc.z, okrow[u] ? hs[u] * r[d][u].z : ninf);
                    acc.w = fmaxf(acc.w, okrow[u] ? hs[u] * r[d][u].w : ninf);
                } else {
                    acc.x = fmaf(hs[u], r[d][u].x, acc.x);
                    acc.y = fmaf(hs[u], r[d][u].y, acc.y);
                    acc.z = fmaf(hs[u], r[d][u].z, acc.z);
                    acc.w = fmaf(hs[u], r[d][u].w, acc.w);
                }
            }
            const int boff = (d & 1) * PSTR;                     // this step's partial-sum buffer (t and d have the same parity)
            if (has_flane) acc.x = is_flane ? __int_as_float(t + 1) : acc.x;      // the flag rides in the first idle lane's slot
            asm volatile("" ::: "memory");
            *reinterpret_cast<v4f *>(wptr + boff) = acc;
            if (!has_flane && lane == 0) lds_flag_set(reinterpret_cast<int *>(fslot + boff), t + 1);   // behind the partial sums in LDS order
            asm volatile("" : "+v"(acc));
            if (t + D < nsteps) FARNN_RGW_ISSUE(d, nlo, nhi);
            // the partners' flags first, this lane's six partial sums behind them in the same batch (chain_regs.hip.h)
            float pv[NQI];
            for (;;) {
                const int fl = lds_flag_get(pflag + boff);
#pragma unroll
                for (int i = 0; i < NQI; i++) pv[i] = qptr[i][boff];
                asm volatile("" ::: "memory");
                if (__ballot(fl < t + 1) == 0ull) break;
            }
            static_assert(NQI == 6, "the reduction tree below is written out for six partial sums per lane");
            float s;
            if (MAXSR) {
                s = fmaxf(fmaxf(fmaxf(pv[0], pv[1]), fmaxf(pv[2], pv[3])), fmaxf(pv[4], pv[5]));
                s = fmaxf(s, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xf, 0xf, true)));
            } else {
                s = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + (pv[4] + pv[5]);
                s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
            }
            const float pre = s * c_pre;                         // (:377-386) / (:393-402): o before the non-linearity forward, after it backward
            float hn;
            if (NLX) hn = apply_nl(pre, nl_mode);
            else     hn = nl_relu ? fmaxf(pre, 0.0f) : pre;
            const float hx = hn * c_post;                        // what the next step multiplies with (rows without a state never reach xch)
            *hptr = hn;                                          // row t + 1 of `hist` (or the dump slot)
            hptr += hstep;
            *xw = hx;
            asm volatile("" ::: "memory");
            v4f hq[NQ4];
#pragma unroll
            for (int q = 0; q < NQ4; q++) hq[q] = *reinterpret_cast<const v4f *>(xr + 4 * q);
            // where the block of step (t + 1) + D is, while the exchange is in flight (chain_regs.hip.h)
            if (d == D - 1 && ((t + 1 + D) & 63) == 0 && t + 1 + D < nsteps) FARNN_RGW_WINDOW(t + 1 + D);
            FARNN_RGW_BASE(t + 1 + D, nlo, nhi);
            asm volatile("" : "+s"(nlo), "+s"(nhi));
#pragma unroll
            for (int q = 0; q < NQ4; q++) { hs[4 * q] = hq[q].x; hs[4 * q + 1] = hq[q].y; hs[4 * q + 2] = hq[q].z; hs[4 * q + 3] = hq[q].w; }
        }
    }
    if (lane == 0) lds_flag_set(reinterpret_cast<int *>(fslot), nsteps + 1);   // (the last state row is in `hist`)
#undef FARNN_RGW_WAIT
#undef FARNN_RGW_ISSUE
#undef FARNN_RGW_LD1
#undef FARNN_RGW_LD4
#undef FARNN_RGW_BASE
#undef FARNN_RGW_WINDOW
}
#undef FARNN_RGW_WAITSTR
#undef FARNN_RGW_OPS4

}  // namespace farnn
