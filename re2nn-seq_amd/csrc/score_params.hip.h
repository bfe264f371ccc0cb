// Parameters of the score + decode stage (K2), shared by the stand-alone score kernels (score_decode.hip.h) and by the
// recurrence kernel that runs the stage as it goes (chain_regs.hip.h).
#pragma once
#include "common.hip.h"
#include "label_map.hip.h"

namespace farnn {

struct ScoreParams {
    const float *A, *Bk;    // stash [B][L+1][SP]
    const float *OT;        // [S][Kc] transposed output matrix, columns >= K zero (alloc padded to 1 KiB)
    const float *P;         // [K][Kc] priority matrix or nullptr
    const float *trT;       // [K][Kp] TRANSPOSED CRF transitions trT[j][i] = tr[i][j], or nullptr
    const int64_t *len;     // [B]
    const int64_t *offs;    // [B+1] exclusive prefix of lengths (flat output) or nullptr: with `flat` set the
                            //        kernel then sums the lengths before its sequence itself (B <= 1024)
    int32_t *tags;          // [B][L] or nullptr
    int64_t *flat;          // [sum len] or nullptr
    float *scores;          // [B][L][K] or nullptr (unclamped, what forward_score returns)
    float *crf_scores;      // [B][L][Kp] workspace: clamped scores for the Viterbi kernel
    const float *OTm;       // matrix-core image of OT (ot_to_mfma_kernel), c16 = ceil(S/16) state groups
    int c16;
    int B, L, S, SP, K, Kp, Kc, kch;
    int full, use_crf, o_idx;
    float threshold;
    int dbg;                // diagnostic ablation mask (FARNN_DBG bits 16/32/64); 0 in production
    int kz;                 // output columns >= kz are exact zero rows of the output matrix BY CONSTRUCTION (the onehot models' CRF
                            // extension: START / STOP, model_decompose_single.py:78-79), 0 = unknown: every column is computed
    LabelMap lm;            // lm.on: the output matrix is a label map (one state, one label, weight 1): label_map.hip.h
};

constexpr int RG_TT = 16;            // tokens per score tile of the stage when it runs beside a recurrence (beside.hip.h)
constexpr int RG_NOB = 2;            // tiles whose rows of the other direction are parked in LDS ahead of the chain's end

// What the score + decode stage needs when it runs BESIDE a recurrence kernel (beside.hip.h): the stashes, the hand-off words of
// the two workgroups of a sequence, and the stage's own parameters.
struct BesideParams {
    const float *A, *Bk;         // stash [B][L+1][SP]
    int B, L, SP, CPR;           // CPR = SP / 4
    unsigned long long *prog;    // [2][B] {epoch, rows stored} per (direction, sequence)
    unsigned long long *arr;     // [B]    {epoch, 1 << 31 | mask of the tiles it scores} of the workgroup that arrived last
    unsigned long long *done;    // sequences whose SECOND workgroup has arrived, over all launches since the words were zeroed: a launch's
                                 // epoch is done / B + 1, read from device memory by every workgroup at its start (beside.hip.h,
                                 // bs_launch_epoch) -- nothing per launch comes from the host, so a captured launch replays correctly
    unsigned epoch;              // (filled in by the kernel from `done`)
    int spin;                    // polls a finished workgroup spends on the tiles of its own half before it leaves them to the other
    int dbg;                     // FARNN_DBG probe mask: read by the profiling build (-DFARNN_PROBES) only
    ScoreParams sp;
};

}  // namespace farnn
