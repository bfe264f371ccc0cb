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
//     0 where it does not -- the coefficients are per-lane constants of the model, loaded once;
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
enum { LM_ST = 0, LM_LB = 1, LM_CF = 2, LM_CC = 8, LM_TL = 9, LM_TH = 10, LM_ROWS = 11 };

struct LabelMap {
    const unsigned *tab;     // [LM_ROWS][128] 32-bit words per position j (positions 0..63: register 0, 64..127: register 1):
                             //   LM_ST   state at position j (pads: 0 -- a valid address; their coefficients are all zero)
                             //   LM_LB   its label (pads: K + j, distinct and above every label)
                             //   LM_CF+d 1.0f if the scan's step d may add its source lane (same label), else 0.0f
                             //   LM_CC   1.0f if position j >= 64 continues the run that position 63 ends
                             //   LM_TL   +0.0f if j ends a run (a label's score lands here), else -inf
                             //   LM_TH   the clamp of j's label: the threshold for the clamped column, else +inf
    int on;                  // 0: the output matrix is not a label map (tab unused)
    int nq;                  // registers in use: 1 (<= 64 labelled states) or 2
    int e0;                  // the first label without any state (its score is +0), -1: none
    float z0;                // that label's clamped score
};

struct LabelMapRegs {
    int st0, st1, lb0, lb1;
    float c0[6], c1[6], cc1, tl0, tl1, th0, th1;
};

__device__ __forceinline__ void lm_load(const LabelMap &lm, int lane, LabelMapRegs &r) {
    const unsigned *t = lm.tab + lane;
    r.st0 = (int)t[LM_ST * 128]; r.st1 = (int)t[LM_ST * 128 + 64];
    r.lb0 = (int)t[LM_LB * 128]; r.lb1 = (int)t[LM_LB * 128 + 64];
#pragma unroll
    for (int d = 0; d < 6; d++) {
        r.c0[d] = __uint_as_float(t[(LM_CF + d) * 128]);
        r.c1[d] = __uint_as_float(t[(LM_CF + d) * 128 + 64]);
    }
    r.cc1 = __uint_as_float(t[LM_CC * 128 + 64]);
    r.tl0 = __uint_as_float(t[LM_TL * 128]); r.tl1 = __uint_as_float(t[LM_TL * 128 + 64]);
    r.th0 = __uint_as_float(t[LM_TH * 128]); r.th1 = __uint_as_float(t[LM_TH * 128 + 64]);
}

// The candidates of one token: y0 / y1 = the clamped score of label lb0 / lb1 at the lanes that end a run, -inf elsewhere.
// own / oth: the two directions' state rows of the token (LDS).
__device__ __forceinline__ void lm_token_scores(const LabelMap &lm, const LabelMapRegs &r, const float *own, const float *oth,
                                                float &y0, float &y1) {
    float x0 = own[r.st0] * oth[r.st0];
    float x1 = 0.0f;
    if (lm.nq > 1) x1 = own[r.st1] * oth[r.st1];
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
    y0 = fminf(x0 + r.tl0, r.th0);
    y1 = fminf(x1 + r.tl1, r.th1);
}

// local_decode (model_onehot.py:162-180) of one token: the first index of the maximum of the clamped scores, K-1 -> o_idx.
// Wave-uniform result.
__device__ __forceinline__ int lm_token_tag(const LabelMap &lm, const LabelMapRegs &r, const float *own, const float *oth,
                                            const int K, const int o_idx) {
    float y0, y1;
    lm_token_scores(lm, r, own, oth, y0, y1);
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
