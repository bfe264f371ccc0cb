// K2s -- the label scores of a token when the output matrix is a LABEL MAP: every state carries at most one label, with
// weight exactly 1 (what the loader writes: fsa_to_tensor.py:585-586 `output_mat[slot, to_state] = 1`; fix_inedge_node makes every
// state of an i-FST have one incoming label, wfa_convert.py:61-147).  Then
//     score[c] = sum_s O[c][s] a[s] b[s]  =  the sum of a[s] b[s] over the states labelled c            (model_onehot.py:346-349)
// is S multiplies and S adds per token instead of the S x K products of the matrix form (score_decode.hip.h / beside.hip.h run
// those on the f32 matrix cores: [16 x S].[S x K] per tile of 16 tokens, two thirds of the time a workgroup spends behind its
// chain).  ONE wavefront scores a token in ~70 instructions:
//   * the states, sorted by (label, state), lie along the lanes -- two registers for up to 128 states; a lane multiplies its
//     state's two entries (own direction's row and the other's, both in LDS);
//   * a segmented inclusive scan over the lanes adds up each label's run: six v_fmac_f32 with a DPP operand per register
//     (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31), x[j] += x[j - d] * cf[j] with cf = 1 where lane j - d holds the same label and
//     0 where it does not -- the coefficients are per-lane constants of the model (one packed word per state, two VGPRs);
//   * the last lane of a run holds the label's score.  Arg-max decode (model_onehot.py:162-180): clamp the `oo` label, wave
//     maximum, the FIRST lane that holds it (the lanes are sorted by label: that is torch.max's first index), against the score
//     +0 of the first label without any state.  CRF decode: the runs' scores are scattered into the position's emission row.
// Arithmetic: every product a[s] b[s] is rounded as in the matrix form; a label's products are added pairwise (a tree) instead of
// one after the other.  On integer-valued states (0/1 automata with `none` / `relu`) both orders are exact, so tags and scores are
// bit-identical to the matrix form's; with tanh-type non-linearities the two differ by rounding (1e-7 relative; the 1e-4 bar).
// The matrix form stays for output matrices that are not label maps (trained C_output_mat), for the priority layer and whenever
// the caller asks for the score tensor.
#pragma once
#include "common.hip.h"

namespace farnn {

constexpr int LM_MAXS = 128;         // states a label map reaches (two registers of 64 lanes)

// One 32-bit word per position j of the sorted states (positions 0..63: register 0, 64..127: register 1):
//   bits  0- 7  the state at position j (pads: 0 -- a valid address; every coefficient of a pad is zero)
//   bits  8-16  its label
//   bits 17-22  scan step d (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31) may add its source lane: same label
//   bit  23     position j >= 64 continues the run that position 63 ends
//   bit  24     j ends a run: the label's score lands here
constexpr int LM_LB_SHIFT = 8, LM_CF_SHIFT = 17, LM_CC_BIT = 23, LM_TL_BIT = 24;

struct LabelMap {
    const unsigned *tab;     // [128] packed words
    int on;                  // 0: the output matrix is not a label map (tab unused)
    int nq;                  // registers in use: 1 (<= 64 labelled states) or 2
    int e0;                  // the first label without any state (its score is +0), -1: none
    float z0;                // that label's clamped score
    int clampcol;            // the column whose score is capped at the threshold (K - 1; K - 3 under a CRF)
    int clamp_empty;         // 1: that column owns no state -- its score is the constant min(0, threshold) (model_decompose.py:353 clamps
                             //    the column whatever feeds it)
    float threshold;
};

struct LabelMapRegs {
    int st0, st1, lb0, lb1;
    float c0[6], c1[6], cc1, tl0, tl1, th0, th1;
};

// this lane's two packed words (a wavefront keeps them from the kernel's start: two VGPRs) ...
__device__ __forceinline__ void lm_load_packed(const LabelMap &lm, int lane, unsigned &pk0, unsigned &pk1) {
    pk0 = lm.tab[lane];
    pk1 = lm.tab[64 + lane];
}
// ... and what a token's scan needs of them, unpacked when the scoring starts (~50 instructions)
__device__ __forceinline__ void lm_unpack(const LabelMap &lm, unsigned pk0, unsigned pk1, LabelMapRegs &r) {
    r.st0 = (int)(pk0 & 0xffu); r.st1 = (int)(pk1 & 0xffu);
    r.lb0 = (int)((pk0 >> LM_LB_SHIFT) & 0x1ffu); r.lb1 = (int)((pk1 >> LM_LB_SHIFT) & 0x1ffu);
#pragma unroll
    for (int d = 0; d < 6; d++) {
        r.c0[d] = ((pk0 >> (LM_CF_SHIFT + d)) & 1u) ? 1.0f : 0.0f;
        r.c1[d] = ((pk1 >> (LM_CF_SHIFT + d)) & 1u) ? 1.0f : 0.0f;
    }
    r.cc1 = ((pk1 >> LM_CC_BIT) & 1u) ? 1.0f : 0.0f;
    r.tl0 = ((pk0 >> LM_TL_BIT) & 1u) ? 0.0f : -INFINITY;
    r.tl1 = ((pk1 >> LM_TL_BIT) & 1u) ? 0.0f : -INFINITY;
    r.th0 = r.lb0 == lm.clampcol ? lm.threshold : INFINITY;
    r.th1 = r.lb1 == lm.clampcol ? lm.threshold : INFINITY;
}
__device__ __forceinline__ void lm_load(const LabelMap &lm, int lane, LabelMapRegs &r) {
    unsigned pk0, pk1;
    lm_load_packed(lm, lane, pk0, pk1);
    lm_unpack(lm, pk0, pk1, r);
}

// torch.min(score, threshold) as the reference's clamp computes it (model_onehot.py:166, model_decompose.py:353): a NaN score stays a NaN
// (fminf would turn it into the threshold; clamp_oo_column_kernel keeps it too)
__device__ __forceinline__ float lm_clamp(float y, float th) { return (y >= th) ? th : y; }

// The candidates of one token from its products x0 / x1 = a[s] * b[s] of this lane's states (x1: 0 when one register is in use):
// y0 / y1 = the clamped score of label lb0 / lb1 at the lanes that end a run, -inf elsewhere.
__device__ __forceinline__ void lm_scan_scores(const LabelMapRegs &r, float x0, float x1, float &y0, float &y1) {
    // x[j] += x[j - d] * cf_d[j]: the segmented scan, both registers interleaved (a DPP read of a VGPR needs two wait states
    // behind the instruction that wrote it: the other register's step and one s_nop)
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_fmac_f32_dpp %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 0\n\t"
                 "v_fmac_f32_dpp %0, %0, %3 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_fmac_f32_dpp %1, %1, %9 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 0\n\t"
                 "v_fmac_f32_dpp %0, %0, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_fmac_f32_dpp %1, %1, %10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 0\n\t"
                 "v_fmac_f32_dpp %0, %0, %5 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_fmac_f32_dpp %1, %1, %11 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 0\n\t"
                 "v_fmac_f32_dpp %0, %0, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %1, %12 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_fmac_f32_dpp %0, %0, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %1, %13 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(x0), "+v"(x1)
                 : "v"(r.c0[0]), "v"(r.c0[1]), "v"(r.c0[2]), "v"(r.c0[3]), "v"(r.c0[4]), "v"(r.c0[5]),
                   "v"(r.c1[0]), "v"(r.c1[1]), "v"(r.c1[2]), "v"(r.c1[3]), "v"(r.c1[4]), "v"(r.c1[5]));
    // a run that crosses from register 0 into register 1: what lane 63 holds is added to every lane of the continuing run
    const float carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x0), 63));
    x1 = fmaf(carry, r.cc1, x1);
    // + (+0.0f) turns a -0 into the +0 torch compares equal to 0; + (-inf) removes the lanes that end no run
    y0 = lm_clamp(x0 + r.tl0, r.th0);
    y1 = lm_clamp(x1 + r.tl1, r.th1);
}

// The same for TWO tokens at once: the four registers' scan steps interleave, so no DPP read waits for the instruction before it
// (two wait states) and a lone wavefront's issue slots carry two tokens' dependency chains instead of one and s_nops.
__device__ __forceinline__ void lm_scan_scores2(const LabelMapRegs &r, float xa0, float xa1, float xb0, float xb1,
                                                float &ya0, float &ya1, float &yb0, float &yb1) {
#define FARNN_LM_STEP(CTRL, C0, C1)                                                            \
    "v_fmac_f32_dpp %0, %0, " C0 " " CTRL "\n\t"                                               \
    "v_fmac_f32_dpp %1, %1, " C1 " " CTRL "\n\t"                                               \
    "v_fmac_f32_dpp %2, %2, " C0 " " CTRL "\n\t"                                               \
    "v_fmac_f32_dpp %3, %3, " C1 " " CTRL "\n\t"
    asm volatile("s_nop 1\n\t"
                 FARNN_LM_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0", "%4", "%10")
                 FARNN_LM_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0", "%5", "%11")
                 FARNN_LM_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0", "%6", "%12")
                 FARNN_LM_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0", "%7", "%13")
                 FARNN_LM_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf", "%8", "%14")
                 FARNN_LM_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf", "%9", "%15")
                 "s_nop 1"
                 : "+v"(xa0), "+v"(xa1), "+v"(xb0), "+v"(xb1)
                 : "v"(r.c0[0]), "v"(r.c0[1]), "v"(r.c0[2]), "v"(r.c0[3]), "v"(r.c0[4]), "v"(r.c0[5]),
                   "v"(r.c1[0]), "v"(r.c1[1]), "v"(r.c1[2]), "v"(r.c1[3]), "v"(r.c1[4]), "v"(r.c1[5]));
#undef FARNN_LM_STEP
    const float ca = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xa0), 63));
    const float cb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xb0), 63));
    xa1 = fmaf(ca, r.cc1, xa1);
    xb1 = fmaf(cb, r.cc1, xb1);
    ya0 = lm_clamp(xa0 + r.tl0, r.th0); ya1 = lm_clamp(xa1 + r.tl1, r.th1);
    yb0 = lm_clamp(xb0 + r.tl0, r.th0); yb1 = lm_clamp(xb1 + r.tl1, r.th1);
}

// own / oth: the two directions' state rows of the token (LDS)
__device__ __forceinline__ void lm_token_scores(const LabelMap &lm, const LabelMapRegs &r, const float *own, const float *oth,
                                                float &y0, float &y1) {
    const float x0 = own[r.st0] * oth[r.st0];
    float x1 = 0.0f;
    if (lm.nq > 1) x1 = own[r.st1] * oth[r.st1];     // (a select, not a multiply by 0: an infinite state must not become a NaN)
    lm_scan_scores(r, x0, x1, y0, y1);
}

// CRF decode: the token's emission row [Kp] (zeros, then every label's clamped score at its column: model_decompose.py:351-353).
// One wavefront; `row` in LDS.  The zeros and the scores are stores of the same wavefront: the LDS keeps their order.
__device__ __forceinline__ void lm_store_emissions(const LabelMap &lm, const LabelMapRegs &r, float y0, float y1, float *row, int Kp, int lane) {
    // (a clamp column that owns no state: min(0, threshold), not the 0 of a column nothing scores)
    const float zc = fminf(0.0f, lm.threshold);
    for (int c = lane; c < Kp; c += WAVE) row[c] = (lm.clamp_empty && c == lm.clampcol) ? zc : 0.0f;
    asm volatile("" ::: "memory");
    if (r.tl0 == 0.0f) row[r.lb0] = y0;
    if (lm.nq > 1 && r.tl1 == 0.0f) row[r.lb1] = y1;
}

// local_decode (model_onehot.py:162-180) of one token from its candidates (y0 / y1 of lm_scan_scores) and their wave maximum m:
// the first index of the maximum of the clamped scores, K-1 -> o_idx.  Wave-uniform result.
__device__ __forceinline__ int lm_tag_from_candidates(const LabelMap &lm, const LabelMapRegs &r, const float y0, const float y1, float m,
                                                      const int K, const int o_idx) {
    if (lm.e0 >= 0) m = fmaxf(m, lm.z0);
    const unsigned long long b0 = __ballot(y0 == m), b1 = __ballot(y1 == m);
    int idx = 0x7fffffff;
    // the lanes are sorted by label: the first lane that holds the maximum has the smallest label
    if (b0) idx = __builtin_amdgcn_readlane(r.lb0, (int)__builtin_ctzll(b0));
    else if (b1) idx = __builtin_amdgcn_readlane(r.lb1, (int)__builtin_ctzll(b1));
    if (lm.e0 >= 0 && lm.z0 == m && lm.e0 < idx) idx = lm.e0;
    if (idx >= K) idx = 0;                               // nothing compares equal (all NaN): 0, like the matrix form
    return idx == K - 1 ? o_idx : idx;
}

// the wave maxima of two values at once (wave_max_dpp's ladder, the two chains interleaved)
__device__ __forceinline__ void wave_max_dpp2(float &a, float &b) {
#define FARNN_MAX2(CTRL) "v_max_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
    asm volatile("s_nop 1\n\t"
                 FARNN_MAX2("row_shr:1 row_mask:0xf bank_mask:0xf")
                 FARNN_MAX2("row_shr:2 row_mask:0xf bank_mask:0xf")
                 FARNN_MAX2("row_shr:4 row_mask:0xf bank_mask:0xf")
                 FARNN_MAX2("row_shr:8 row_mask:0xf bank_mask:0xf")
                 FARNN_MAX2("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 FARNN_MAX2("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(a), "+v"(b));
#undef FARNN_MAX2
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
}

__device__ __forceinline__ int lm_tag_from_products(const LabelMap &lm, const LabelMapRegs &r, const float x0, const float x1,
                                                    const int K, const int o_idx) {
    float y0, y1;
    lm_scan_scores(r, x0, x1, y0, y1);
    float m = wave_max_dpp(fmaxf(y0, y1));
    if (lm.e0 >= 0) m = fmaxf(m, lm.z0);
    const unsigned long long b0 = __ballot(y0 == m), b1 = __ballot(y1 == m);
    int idx = 0x7fffffff;
    // the lanes are sorted by label: the first lane that holds the maximum has the smallest label
    if (b0) idx = __builtin_amdgcn_readlane(r.lb0, (int)__builtin_ctzll(b0));
    else if (b1) idx = __builtin_amdgcn_readlane(r.lb1, (int)__builtin_ctzll(b1));
    if (lm.e0 >= 0 && lm.z0 == m && lm.e0 < idx) idx = lm.e0;
    if (idx >= K) idx = 0;                               // nothing compares equal (all NaN): 0, like the matrix form
    return idx == K - 1 ? o_idx : idx;
}

}  // namespace farnn
