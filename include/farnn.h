/*
 * farnn.h -- C-ABI of the MI355X-native FA-RNN forward tagging path (libfarnn_hip.so).
 *
 * The reference (jeffchy/RE2NN-SEQ) is pure PyTorch and has no FFI of its own; the seam this
 * library replaces is the nn.Module method contract consumed by the reference's callers
 * (SURVEY.md section 8b).  Each entry point cites the reference interface it stands in for
 * (paths relative to the reference checkout).  The reference-side binding a maintainer would
 * add (a ctypes stub inside the model classes) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch / C++ types in any signature;
 *   - every function returns FARNN_OK (0) or a negative errno-style code, never throws;
 *     farnn_last_error() returns a human-readable message for the calling thread;
 *   - weights are given as HOST pointers (float32, C-contiguous) unless desc.weights_on_device
 *     is set; create() copies them into device-resident, re-laid-out storage that the handle
 *     owns until farnn_destroy();
 *   - x / lengths / tags / scores of farnn_tag() are DEVICE pointers (e.g. tensor.data_ptr()
 *     of PyTorch-ROCm tensors used purely as containers); the library never retains them
 *     after the call's work has been enqueued on `stream`;
 *   - one caller thread per handle (the reference is single-threaded Python); one workspace per handle: a call on another
 *     stream than the previous call's is ordered behind that call's work by the library.  The ONE thing the handle keeps of a
 *     call is its stream handle, until the next call: the first time a handle is called on a different stream, the previous
 *     call's stream must still exist (the library records its ordering event there); from then on every call leaves the
 *     handle's own event behind itself and earlier streams are never touched again.  A call that is being captured into a
 *     HIP graph is not ordered against other streams.
 *
 * Environment switches (read ONCE per handle, when farnn_*_create / farnn_train_create builds it; no call on the tagging path
 * touches the environment).  Every one selects between code paths that the test suite holds to the same results.  Those marked
 * [A/B build only] select forms the production library does not carry (its dispatch never picks them: 54 kernels): they exist in
 * the A/B build (csrc/build.py --probes -> libfarnn_hip_probes.so, loaded through FARNN_LIB; farnn_ab_build() says which one is
 * loaded), and farnn_*_create of the production library answers them with FARNN_EINVAL and a message that says so (like every
 * switch they are resolved when the handle is created, never at tag time); switches of earlier rounds that no longer exist
 * (FARNN_CV_WIDE, FARNN_DECOMP_OLD) are refused the same way by both builds:
 *   FARNN_NOFUSE=1          the multi-launch forms (recurrence kernel, then score / Viterbi kernel) instead of one launch per step
 *                           (also: the compact form's two launches instead of compact_tag_kernel)
 *   FARNN_FUSE=1            onehot i-FST, S <= 72, label-map scores: ONE launch per step (scores + decode beside the recurrence) instead of
 *                           the default there, the recurrence kernel + the label-map score launch (faster at every measured shape)
 *   FARNN_NOREGS=1          the LDS-ring recurrence kernel where the register-fed one (S <= 128) would run
 *   FARNN_NODEST=1          [A/B build only] S <= 72, sum semiring: round 3's compute wavefronts (a block split by SOURCE rows,
 *                           partial sums reduced across the wavefronts) instead of the destination-split ones (chain_dest.hip.h)
 *   FARNN_NOLABELMAP=1      scores on the matrix cores even when the output matrix is a label map (one state, one label, weight 1)
 *   FARNN_CV_ONE=1          [A/B build only] a CRF on the onehot i-FST, S <= 108: recurrence + scores + Viterbi in ONE launch
 *                           (chain_viterbi_kernel).  Two launches are faster at every measured shape
 *   FARNN_WIDE_UNPAIRED=1   72 < S <= 108: one workgroup per compute unit for the wide recurrence (default: two, label-map scores)
 *   FARNN_CV_STASH=1        [A/B build only] the one-launch CRF kernel with the state rows through the stash instead of LDS
 *   FARNN_VITERBI_BP=1 / FARNN_VITERBI_UNFUSED=1   the stored-back-pointer Viterbi kernel / scores through HBM in front of it
 *   FARNN_PREP=1, FARNN_NOSORT=1                   the separate batch-prep kernel / the batch's own launch order
 *   FARNN_DECOMP_NOREGS=1, FARNN_ROWS_NOREGS=1     the decomposed recurrence's LDS-fed kernels instead of the register forms
 *   FARNN_ROWS_LPR4=1|2     gated decomposed models (farnn = 2, S <= 160): 1 = four lanes per row instead of eight (round 3's forms),
 *                           2 = eight lanes per row but P2 swept from LDS (without the all-in-registers / mixed eight-lane forms)
 *   FARNN_ROWS_NOROUNDS=1   the decomposed recurrence's register forms with one workgroup per sequence and direction (round 5) instead of
 *                           one per compute unit walking its sequences with the weights kept in registers
 *   FARNN_TRAIN_NOLDS=1|2, FARNN_TRAIN_NSEQ=2|4    training chains with the matrices read through L2 / sequences per workgroup
 * Diagnostic switches (ablations, geometry overrides: FARNN_DBG, FARNN_KS, FARNN_RPG, FARNN_NLD, FARNN_FUSE_SPIN, FARNN_SOLO_MARGIN,
 * ...) exist only in the profiling build of the library (csrc/build.py --probes); the production library ignores them.
 */
#ifndef FARNN_H
#define FARNN_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: FARNN_MODE_RE, farnn_tag_host_submit/_wait, farnn_flatten_host, the compact / folded creators, farnn_set_compact
 * (round 2); calls on different streams of one handle are ordered by the library (round 3).  1: round 1. */
#define FARNN_ABI_VERSION 2

/* ---- return codes ------------------------------------------------------------------- */
#define FARNN_OK         0
#define FARNN_EINVAL   (-22)   /* bad argument / unsupported combination               */
#define FARNN_ENOMEM   (-12)   /* device or host allocation failed                     */
#define FARNN_ENODEV   (-19)   /* no usable HIP device                                 */
#define FARNN_EIO       (-5)   /* a HIP runtime call failed (see farnn_last_error)     */
#define FARNN_ERANGE   (-34)   /* size beyond what the kernels support                 */

/* ---- enumerations ------------------------------------------------------------------- */
/* --update_nonlinear (reference main.py:93, model_onehot.py:379-386) */
#define FARNN_NL_NONE      0
#define FARNN_NL_RELU      1
#define FARNN_NL_TANH      2
#define FARNN_NL_RELUTANH  3
#define FARNN_NL_SIGMOID   4   /* only --additional_nonlinear uses it (model_decompose.py:232) */
/* --train_mode (reference main.py:33, utils.py:192-199) */
#define FARNN_SEMIRING_SUM 0
#define FARNN_SEMIRING_MAX 1
/* farnn_tag() modes */
#define FARNN_MODE_LOCAL   0   /* forward_local: valid positions only (model_onehot.py:131-146) */
#define FARNN_MODE_FULL    1   /* forward_score: all L positions incl. pads, scores unclamped (:351-428) */
#define FARNN_MODE_RE      2   /* forward_RE (:148-160): FULL, and the `oo` column (C-1) of `scores` comes back capped at the
                                  threshold (:153-154); onehot models only */

typedef struct farnn_model farnn_model;   /* opaque handle */

/* ---- onehot i-FST: FARNN_S_O_I_S (reference model_onehot.py:310-428), --independent 2 ---- */
typedef struct farnn_onehot_ifst_desc {
    int32_t V, S, C;            /* vocabulary (incl. pad row), states, label columns (len(s2i)+1) */
    const float *T;             /* [V,S,S] language_tensor   (:326-328)                     */
    const float *W;             /* [S,S]   wildcard_mat      (:329-331)                     */
    const float *O;             /* [C,S]   output_mat        (:332-333)                     */
    const float *h0;            /* [S]     start_vector      (:322-323)                     */
    const float *hT;            /* [S]     final_vector      (:324-325)                     */
    const float *P;             /* [C,C] expanded priority matrix (priority.py:6-18) or NULL */
    int32_t nl;                 /* FARNN_NL_*                                                */
    int32_t semiring;           /* FARNN_SEMIRING_*                                          */
    float   threshold;          /* --threshold: clamp of the `oo` column (:166-167)          */
    int32_t o_idx;              /* s2i['o']: what the `oo` column decodes to (:169)          */
    int32_t use_crf;            /* 0: threshold+argmax; 1: Viterbi over C+2 tags */
    const float *crf_trans;     /* [(C+2),(C+2)] CRF.transitions (crf.py:39-46) or NULL=defaults */
    int32_t weights_on_device;  /* 1: T/W/O/h0/hT/P/crf_trans are device pointers            */
} farnn_onehot_ifst_desc;

int farnn_onehot_ifst_create(const farnn_onehot_ifst_desc *desc, int device, farnn_model **out);

/* ---- onehot models built on the device from the automaton's edge list -----------------
 * The reference's loaders (wfa/fsa_to_tensor.py:398-615) write dense float64 tensors on the host
 * ([V,S,S]: 21 GB at BASELINE configs[4]; [V,C,S,S]: 2.5 GB at ATIS size).  These entries take the edges
 * the loader would have written and scatter them straight into HBM.  Per edge e, with w = word[e],
 * f = from[e], t = to[e], l = label[e] (label column of the edge; destination-state label in the i-FST):
 *     i-FST   (:546-615)  w >= 0: T[w,f,t] = v     w == -1: W[f,t] = v        l >= 0: O[l,t] = 1
 *     FST 4-D (:398-474)  w >= 0: T4[w,l,f,t] = v  w == -1: W4[l,f,t] = v
 *     indep=1 (:477-543)  w >= 0: T[w,f,t] = v     w == -1: W[f,t] = v        l >= 0: Oten[l,f,t] = 1
 * w < -1 marks an entry that only carries its label (a class edge, '&' / '%', none of whose words is in
 * the vocabulary).  Assignment, not accumulation, like the reference (duplicates are idempotent); class
 * edges are expanded to one entry per word by the caller.  The edge arrays are host pointers (they are
 * small).  The dense tensor fields of `base` are ignored; `base.weights_on_device` covers h0/hT/P/crf_trans. */
typedef struct farnn_edge_list {
    int64_t n_edges;
    const int32_t *word;           /* [n_edges] word id, -1 wildcard edge, < -1 label only         */
    const int32_t *from, *to;      /* [n_edges] state indices                                      */
    const int32_t *label;          /* [n_edges] label column, or -1 (NULL: no labels)              */
    const float *val;              /* [n_edges] edge weight, or NULL (= 1)                         */
} farnn_edge_list;

int farnn_onehot_ifst_create_from_edges(const farnn_onehot_ifst_desc *base, const farnn_edge_list *edges,
                                        int device, farnn_model **out);

/* ---- compact form of the onehot i-FST ----------------------------------------------------------
 * With --rand_constant 0 (all main.py allows for --method onehot, :175-176) every entry of T and W is 0 or 1 and a
 * word's S x S block holds a handful of edges; the loader still writes dense float64 tensors
 * (wfa/fsa_to_tensor.py:546-615).  The compact form keeps a block as S rows of S bits (forward: sources of every
 * destination; backward: destinations of every source) and the recurrence walks only the non-zero entries of the
 * current state: 2*S*ceil(S/64)*8 bytes per token instead of 2*S*S*4, same results (bit-identical for `none` / `relu`,
 * within rounding order for tanh).  Sum semiring, S <= 512, L <= 1024.  Contract numbers stay fp32-dense; bench.py
 * reports this path separately (`compact`).
 *   farnn_onehot_ifst_create / _create_from_edges build the bitmaps beside the dense blocks whenever the weights are
 *   0/1 (farnn_has_compact); farnn_set_compact(m, 1) makes farnn_tag use them, (m, 0) goes back to the dense blocks.
 *   farnn_onehot_ifst_create_compact builds ONLY the bitmaps, straight from the edge list (no dense tensor on the host
 *   or in HBM: BASELINE's largest config is 0.66 GB instead of 2 x 21 GB); edge weights must be 1 (or val == NULL). */
int farnn_has_compact(const farnn_model *m);
int farnn_set_compact(farnn_model *m, int32_t enable);
int farnn_onehot_ifst_create_compact(const farnn_onehot_ifst_desc *base, const farnn_edge_list *edges, int device,
                                     farnn_model **out);

/* ---- onehot FST 4-D: FARNN_S_O (reference model_onehot.py:8-129), --independent 0 ---- */
typedef struct farnn_onehot_fst4_desc {
    int32_t V, S, C;
    const float *T4;            /* [V,C,S,S] language_tensor (:34)  */
    const float *W4;            /* [C,S,S]   wildcard_tensor (:35)  */
    const float *h0, *hT;       /* [S]                              */
    const float *P;             /* [C,C] or NULL                    */
    int32_t semiring;           /* relu is unconditional (:93-94)   */
    float   threshold;
    int32_t o_idx;
    int32_t weights_on_device;
} farnn_onehot_fst4_desc;

int farnn_onehot_fst4_create(const farnn_onehot_fst4_desc *desc, int device, farnn_model **out);

/* ---- onehot independent=1: FARNN_S_O_I (reference model_onehot.py:184-306) ---- */
typedef struct farnn_onehot_ind1_desc {
    int32_t V, S, C;
    const float *T;             /* [V,S,S]                           */
    const float *W;             /* [S,S]                             */
    const float *Oten;          /* [C,S,S] output_tensor (:212-213)  */
    const float *h0, *hT;
    const float *P;
    int32_t semiring;
    int32_t mask_by_output;     /* the `args.independent == 2` branch (:259-262) */
    float   threshold;
    int32_t o_idx;
    int32_t weights_on_device;
} farnn_onehot_ind1_desc;

int farnn_onehot_ind1_create(const farnn_onehot_ind1_desc *desc, int device, farnn_model **out);

/* the same two models from the edge list (see farnn_edge_list above) */
int farnn_onehot_fst4_create_from_edges(const farnn_onehot_fst4_desc *base, const farnn_edge_list *edges,
                                        int device, farnn_model **out);
int farnn_onehot_ind1_create_from_edges(const farnn_onehot_ind1_desc *base, const farnn_edge_list *edges,
                                        int device, farnn_model **out);

/* ---- decomposed i-FST: FARNN_S_D_W_I_S (reference model_decompose_single.py:12-304) ---- */
typedef struct farnn_decomp_ifst_desc {
    int32_t V, S, R, K;         /* S incl. additional_states; K = score columns (C, or C+2 with CRF) */
    const float *Vgen;          /* [V,R] generalized word table: V_embed*beta + nl_add(E@G)*(1-beta),
                                   model_decompose.py:222-241, precomputed once (weights frozen)   */
    const float *S1, *S2;       /* [S,R]  (model_decompose_single.py:70-71)                        */
    const float *W;             /* [S,S]  wildcard_mat (:84-86)                                     */
    const float *Cout;          /* [K,S]  C_output_mat (:81-82)                                     */
    const float *h0, *hT;       /* [S]                                                              */
    const float *P;             /* [K,K] or NULL                                                    */
    int32_t farnn;              /* 0 plain, 1 update gate, 2 update+reset gates (:143-154)          */
    const float *Wss1, *Wrs1, *bs1;   /* [S,S], [R,S], [S]   (farnn >= 1)                           */
    const float *Wss2, *Wrs2, *bs2;   /* [S,S], [R,S], [S]   (farnn == 2)                           */
    float   sigmoid_exponent;   /* gate_activation scale (model_decompose.py:97-102)                */
    int32_t nl;
    int32_t semiring;
    float   threshold;
    int32_t o_idx;
    int32_t use_crf;
    const float *crf_trans;     /* [K,K] when use_crf                                               */
    int32_t weights_on_device;
} farnn_decomp_ifst_desc;

int farnn_decomp_ifst_create(const farnn_decomp_ifst_desc *desc, int device, farnn_model **out);

/* The same model with the word table -- and --normalize_automata -- computed ON THE DEVICE.  The
 * reference recomputes get_generalized_v_embed_vec twice per time step (model_decompose_single.py:140-141):
 *     Vgen[w] = V_embed[w] * beta + nl_add(E[w] @ G) * (1 - beta)                 (model_decompose.py:222-241)
 * and its loader scales V_embed, S1, S2 by the averaged column norms on the host (init_params.py:285-297):
 *     factor = cbrt(avg(V) avg(S1) avg(S2)),  M <- M * factor / avg(M),  avg(M)[c] = ||M[:, c]|| / rows  (utils.py:202-225).
 * Here desc->Vgen is ignored: V_embed / E / G / beta go to the device once, the scaling (all four modes) and the fold
 * run there, and nothing of size V x R comes back to the host.  G = pinv(E) @ V_embed (model_decompose_single.py:73-76)
 * is given for the UN-normalised V_embed; its columns take V_embed's scale (the bridge is linear in them). */
#define FARNN_NORM_NONE     0
#define FARNN_NORM_L1       1      /* --normalize_automata l1: numpy's matrix 1-norm (largest column sum) / size          */
#define FARNN_NORM_L2       2      /* --normalize_automata l2: the spectral norm / size (Gram matrix on the device, its    */
                                   /*   largest eigenvalue by a Jacobi sweep over R x R doubles on the host)               */
#define FARNN_NORM_L1_RANK  3      /* --normalize_automata l1-rank */
#define FARNN_NORM_L2_RANK  4      /* --normalize_automata l2-rank (main.py:55 default) */
typedef struct farnn_vgen_fold {
    const float *V_embed;       /* [V,R]  (pad row included)                                                  */
    const float *E;             /* [V,D]  embedding.weight                                                    */
    const float *G;             /* [D,R]  embed_r_generalized                                                 */
    const float *beta;          /* [R]    beta_vec                                                            */
    int32_t D;
    int32_t add_nl;             /* FARNN_NL_*: --additional_nonlinear                                         */
    int32_t normalize;          /* FARNN_NORM_*: applied to V_embed (and G), desc->S1, desc->S2 before the fold */
    int32_t on_device;          /* 1: V_embed / E / G / beta are device pointers                              */
} farnn_vgen_fold;
int farnn_decomp_ifst_create_folded(const farnn_decomp_ifst_desc *desc, const farnn_vgen_fold *fold, int device,
                                    farnn_model **out);

/* ---- decomposed independent=1: FARNN_S_D_W_I (reference model_decompose_independent.py:11-300) ---- */
typedef struct farnn_decomp_ind1_desc {
    int32_t V, S, R, RO, K;     /* S incl. additional_states; RO = rank of the output factors        */
    const float *Vgen;          /* [V,R]  generalized word table (model_decompose.py:222-241)        */
    const float *S1, *S2;       /* [S,R]                                                              */
    const float *W;             /* [S,S]  wildcard_mat                                                */
    const float *Cout;          /* [K,RO] C_output (:82-83)                                           */
    const float *S1o, *S2o;     /* [S,RO] S1_output / S2_output (:85-89)                              */
    const float *Wo;            /* [S,S]  wildcard_output, added to the output sum unless CE1; NULL   */
    const float *h0, *hT;       /* [S]                                                                */
    const float *P;             /* [K,K] or NULL                                                      */
    int32_t farnn;
    const float *Wss1, *Wrs1, *bs1;
    const float *Wss2, *Wrs2, *bs2;
    float   sigmoid_exponent;
    int32_t nl;
    int32_t semiring;
    float   threshold;
    int32_t o_idx;
    int32_t use_crf;
    const float *crf_trans;     /* [K,K] when use_crf                                                 */
    int32_t weights_on_device;
} farnn_decomp_ind1_desc;

int farnn_decomp_ind1_create(const farnn_decomp_ind1_desc *desc, int device, farnn_model **out);

/* ---- decomposed independent=0: FARNN_S_D_W (reference model_decompose.py:10-459) ------------------ */
typedef struct farnn_decomp_fst_desc {
    int32_t V, S, R, RW, K;     /* S incl. additional_states; RW = rank of the wildcard factors       */
    const float *Vgen;          /* [V,R]  generalized word table (model_decompose.py:222-241)        */
    const float *C;             /* [K,R]  C_embed (:125); the recurrence sees Vgen * sum_c C (:253)   */
    const float *S1, *S2;       /* [S,R]                                                              */
    const float *Cw;            /* [K,RW] C_wildcard (:122-123)                                       */
    const float *S1w, *S2w;     /* [S,RW] S1_wildcard / S2_wildcard (:127-131)                        */
    const float *WW;            /* [S,S]  wildcard_wildcard (:133-135)                                */
    const float *h0, *hT;       /* [S]                                                                */
    const float *P;             /* [K,K] or NULL                                                      */
    int32_t farnn;
    const float *Wss1, *Wrs1, *bs1;
    const float *Wss2, *Wrs2, *bs2;
    float   sigmoid_exponent;
    int32_t nl;
    int32_t semiring;
    float   threshold;
    int32_t o_idx;
    int32_t use_crf;
    const float *crf_trans;     /* [K,K] when use_crf                                                 */
    int32_t weights_on_device;
} farnn_decomp_fst_desc;

int farnn_decomp_fst_create(const farnn_decomp_fst_desc *desc, int device, farnn_model **out);

/* ---- the hot path ------------------------------------------------------------------- */
/*
 * farnn_tag: model.forward_local / forward_RE / forward_score of the reference
 *            (model_onehot.py:131-160, model_decompose_single.py:207-304), minus the loss.
 *
 *   x        int64 [B,L]   token ids, pad id at positions >= lengths[b]        (device)
 *   lengths  int64 [B]     1..L                                                (device)
 *   mode     FARNN_MODE_LOCAL: only positions < lengths[b] are tagged; FARNN_MODE_FULL: all L
 *            positions run through the recurrence exactly like the reference's padded loop.
 *   tags     int32 [B,L] or NULL. LOCAL: positions >= lengths[b] are set to -1.       (device)
 *   flat_tags int64 [sum(lengths)] or NULL: the reference's flattened prediction order
 *            (utils.py:153-164): for b in batch, positions 0..lengths[b]-1.            (device)
 *   scores   float32 [B,L,K] or NULL: per-token label scores BEFORE the threshold clamp
 *            (K = farnn_num_columns()).  LOCAL: rows at pad positions are zero-filled. (device)
 *   stream   hipStream_t (as void*), NULL = the default stream.  Work is enqueued, not awaited.
 */
int farnn_tag(farnn_model *m, const int64_t *x, const int64_t *lengths, int32_t B, int32_t L,
              int32_t mode, int32_t *tags, int64_t *flat_tags, float *scores, void *stream);

/* Pre-size the handle's device workspace so that farnn_tag() for up to (B,L) allocates nothing
 * (required before capturing farnn_tag() into a hipGraph). */
int farnn_reserve(farnn_model *m, int32_t B, int32_t L);

void farnn_destroy(farnn_model *m);

/* ---- the same call with HOST buffers: what the reference's eval loop hands over (val.py:17-31) -----------------------
 * model.forward_local(x, label, lengths, train=False) is called with CPU tensors and its flat predictions are read on
 * the CPU.  farnn_tag_host_submit stages [x | lengths] through a pinned buffer the handle owns, enqueues ONE H2D copy,
 * farnn_tag(FARNN_MODE_LOCAL) and the D2H copy of the flat predictions on a stream of the handle, and returns a ticket at
 * once; farnn_tag_host_wait blocks until that batch is done and copies its predictions out.  Up to FARNN_HOST_SLOTS
 * batches may be in flight (the host prepares batch i+1 while batch i runs); tickets are waited for in submission order.
 * x_host / len_host may be reused as soon as submit returns.  PCIe-inclusive path: bench.py reports it as
 * `host_inclusive`, never as the headline value.
 *   x_host    int64 [B,L]   (host)          len_host  int64 [B]  (host)
 *   ticket    out: names THIS submit (slot | generation << 8) for farnn_tag_host_wait; a ticket that was already waited for, or
 *             whose slot has since been reclaimed for a later submit, is refused with FARNN_EINVAL and consumes nothing;
 *             n_flat out (or NULL): sum(clamp(len, 0, L))
 *   flat_out  int64 [n_flat] (host)         n_out (or NULL): how many were written
 * farnn_flatten_host: utils.flatten (utils.py:153-164) of a host int64 [B,L] array (the flat gold labels the same
 * call returns beside the predictions); returns the element count, -1 on a null argument. */
#define FARNN_HOST_SLOTS 4
int farnn_tag_host_submit(farnn_model *m, const int64_t *x_host, const int64_t *len_host, int32_t B, int32_t L,
                          int32_t *ticket, int64_t *n_flat);
int farnn_tag_host_wait(farnn_model *m, int32_t ticket, int64_t *flat_out, int64_t *n_out);
int64_t farnn_flatten_host(const int64_t *a, const int64_t *len_host, int32_t B, int32_t L, int64_t *out);

/* ---- training step of the decomposed i-FST -----------------------------------
 * Replaces FARNN_S_D_W_I_S.forward_local(train=True) + loss.backward()
 * (model_decompose_single.py:207-304, train_decompose.py:186-190) for farnn = 0/1/2, the sum semiring and the
 * CE1 loss: cross-entropy (mean over the valid tokens) of the scores, or with use_crf the CRF negative
 * log-likelihood, and the gradient with respect to every tensor the recurrence and the scoring read.  The generalized word table Vgen
 * (model_decompose.py:222-241) is an input; the caller differentiates it from dVgen.
 * All pointers are DEVICE pointers; matrices are row-major and unpadded. */
typedef struct farnn_train_ctx farnn_train_ctx;

typedef struct {
    int32_t V, S, R, K;         /* vocabulary rows of Vgen, states (incl. additional states), rank, score columns */
    int32_t nl;                 /* FARNN_NL_* (update_nonlinear)                                               */
    float   threshold;          /* decode clamp of column K-1 (model_decompose.py:365)                         */
    int32_t o_idx;              /* label written for column K-1 (:367) / K-3 with the CRF (:356)                 */
    int32_t farnn;              /* 0 plain recurrence, 1 update gate, 2 update + reset gate (:143-154,:193-198) */
    float   sigmoid_exponent;   /* k of the gate activation sigmoid(k x) (model_decompose.py:97-103)              */
    int32_t use_crf;            /* 1: loss = CRF.neg_log_likelihood_loss (baselines/crf.py:250-260, a sum over the
                                   batch) on the scores, K = labels + 2 (START, STOP); decode = Viterbi (:351-356) */
} farnn_train_dims;

typedef struct {
    const float *Vgen;          /* [V][R]   */
    const float *S1, *S2;       /* [S][R]   */
    const float *W;             /* [S][S] wildcard_mat   */
    const float *C;             /* [K][S] C_output_mat   */
    const float *h0, *hT;       /* [S]      */
    const float *P;             /* [K][K] priority matrix or NULL (args.use_priority = 0) */
    const float *crf_trans;     /* [K][K] crf.transitions (use_crf = 1), else NULL        */
    const float *Wss1, *Wrs1, *bs1;   /* [S][S], [R][S], [S] update gate (farnn >= 1), else NULL */
    const float *Wss2, *Wrs2, *bs2;   /* reset gate (farnn = 2), else NULL                       */
} farnn_train_weights;

typedef struct {
    float *loss;                /* [1]                                                  */
    float *dVgen;               /* [V][R]  (rows of words that do not occur stay zero)  */
    float *dS1, *dS2;           /* [S][R]  */
    float *dW;                  /* [S][S]  */
    float *dC;                  /* [K][S]  */
    float *dh0, *dhT;           /* [S]     */
    int32_t *tags;              /* [B][L] decoded labels of this forward pass, -1 at pad positions */
    float *dtrans;              /* [K][K] gradient of crf.transitions (use_crf = 1), else NULL     */
    float *dWss1, *dWrs1, *dbs1;      /* gate gradients (farnn >= 1), else NULL */
    float *dWss2, *dWrs2, *dbs2;      /* (farnn = 2), else NULL                 */
} farnn_train_outputs;

int  farnn_train_create(const farnn_train_dims *dims, int device, farnn_train_ctx **out);
void farnn_train_destroy(farnn_train_ctx *ctx);
/* One step on the given stream: zeroes the outputs, runs both chains with the state stash, the loss, the
 * back-propagation through time and the parameter-gradient reductions.  x, lengths, labels: int64
 * [B][L], [B], [B][L] (labels outside 0..K-1 are treated as 0; they only matter at valid positions). */
int  farnn_decomp_ifst_train_step(farnn_train_ctx *ctx, const farnn_train_weights *w, const int64_t *x,
                                  const int64_t *lengths, const int64_t *labels, int32_t B, int32_t L,
                                  int64_t valid_tokens, const farnn_train_outputs *out, void *stream);
/* accumulated HIP-event time of the step's launches since the last call (ms) and the number of steps timed;
 * profiling is enabled with farnn_train_set_profiling(ctx, 1) */
int  farnn_train_set_profiling(farnn_train_ctx *ctx, int32_t enable);
int  farnn_train_time(farnn_train_ctx *ctx, double *total_ms, int64_t *steps);

/* ---- introspection / measurement ----------------------------------------------------- */
int  farnn_abi_version(void);
/* 1: the A/B (profiling) build of the library (csrc/build.py --probes): it also carries the forms the production build left behind --
 * FARNN_CV_ONE / FARNN_CV_STASH, the one-launch CRF step -- and the in-kernel probes (FARNN_DBG); 0: the production build. */
int  farnn_ab_build(void);
int  farnn_device_count(void);
const char *farnn_last_error(void);
int  farnn_num_columns(const farnn_model *m);              /* K of the scores tensor */
/* algorithmic HBM bytes of one farnn_tag() call with `valid_tokens` tagged tokens
 * (SURVEY.md 8d / DESIGN.md: per-token figure x tokens), for the roofline line of bench.py */
double farnn_algorithmic_bytes(const farnn_model *m, int64_t valid_tokens);
/* the share of that figure read/written by kernel `which` (0 = recurrence chain, 1 = score+decode) */
double farnn_kernel_algorithmic_bytes(const farnn_model *m, int32_t which, int64_t valid_tokens);
/* per-kernel timing with HIP events recorded on the launch stream.  enable=N>0 starts collecting
 * on every N-th farnn_tag() call (N=1: every call; event records cost a few microseconds of
 * launch latency each, so a stride keeps the timed region honest); enable=0 stops.
 * farnn_kernel_time() synchronises, then reports the accumulated milliseconds and the number of
 * timed launches of kernel `which` (0 = recurrence chain, 1 = score+decode, 2 = prep). */
int  farnn_set_profiling(farnn_model *m, int32_t enable);
int  farnn_kernel_time(farnn_model *m, int32_t which, double *total_ms, int64_t *launches);
const char *farnn_kernel_name(const farnn_model *m, int32_t which);

#ifdef __cplusplus
}
#endif
#endif /* FARNN_H */
