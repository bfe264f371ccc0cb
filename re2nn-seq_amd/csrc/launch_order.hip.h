// K0 -- the launch order of the recurrence kernels, computed where it is used.
#pragma once
#include "common.hip.h"

namespace farnn {

// In-kernel replacement of the batch-prep launch order for B <= 1024 (saves a ~5.5 us kernel in front of
// every call): the workgroup in launch slot `slot` picks the sequence whose length rank (descending,
// ties by index) is slot's rank under the fold of batch_prep_*_kernel (slot < B/2: rank = slot, else
// rank = B-1 - (slot - B/2), so that workgroups i and i + B/2 pair a long with a short sequence).
// Counting select: histogram of the lengths in LDS, a descending scan by one wavefront finds the
// length class and the index inside it, a ballot/popcount pass finds the sequence.  Deterministic
// (no dependence on atomic order), every workgroup of a launch sees the same permutation.
// `scratch` needs L + 1 + 16 ints of LDS that nothing else uses yet.
__device__ __forceinline__ int folded_rank(int slot, int B) {
    const int half = B / 2;
    return slot < half ? slot : (B - 1) - (slot - half);
}

__device__ __forceinline__ int select_by_length_rank(const int64_t *len, int B, int L, int rank, int *scratch,
                                                     int tid, int nthreads) {
    const int lane = tid & 63, w = tid >> 6, nwaves = nthreads >> 6;
    int *hist = scratch;                 // [L + 1]
    int *misc = scratch + L + 1;         // [0] length class, [1] index inside the class, [2] result, [4..] per-wave counts
    for (int i = tid; i <= L; i += nthreads) hist[i] = 0;
    __syncthreads();
    for (int k = tid; k < B; k += nthreads) {
        int v = (int)len[k];
        v = v < 0 ? 0 : (v > L ? L : v);
        atomicAdd(&hist[v], 1);
    }
    __syncthreads();
    if (w == 0) {                        // descending scan: lane i owns the lengths L - i, L - i - 64, ...
        int above = 0;                   // sequences strictly longer than the chunk being scanned
        for (int base = L; base >= 0; base -= 64) {
            const int l = base - lane;
            const int c = l >= 0 ? hist[l] : 0;
            int inc = c;                 // inclusive prefix over the lanes (longer lengths first)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            const int before = above + inc - c;
            if (c > 0 && rank >= before && rank < before + c) { misc[0] = l; misc[1] = rank - before; }
            above += __shfl(inc, 63, 64);
            if (above > rank) break;     // uniform: the class has been found
        }
    }
    __syncthreads();
    const int cls = misc[0];
    int rem = misc[1];
    for (int q0 = 0; q0 < B; q0 += nthreads) {      // the rem-th sequence (by index) of that length
        const int k = q0 + tid;
        int v = k < B ? (int)len[k] : -1;
        if (k < B) v = v < 0 ? 0 : (v > L ? L : v);
        const bool flag = v == cls;
        const unsigned long long m = __ballot(flag);
        if (lane == 0) misc[4 + w] = __popcll(m);
        __syncthreads();
        int pre = 0, tot = 0;
        for (int ww = 0; ww < nwaves; ww++) { const int c = misc[4 + ww]; pre += ww < w ? c : 0; tot += c; }
        if (rem < tot) {
            const int local = rem - pre;
            if (flag && local >= 0 && __popcll(m & ((1ull << lane) - 1ull)) == local) misc[2] = k;
            __syncthreads();
            break;
        }
        rem -= tot;
        __syncthreads();
    }
    const int b = misc[2];
    __syncthreads();                     // the scratch is reused by the caller
    return __builtin_amdgcn_readfirstlane(b);
}

}  // namespace farnn
