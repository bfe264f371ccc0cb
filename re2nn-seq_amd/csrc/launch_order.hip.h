// K0 -- the launch order of the recurrence kernels, computed where it is used.
#pragma once
#include "common.hip.h"

namespace farnn {

// In-kernel replacement of the batch-prep launch order for B <= 1024 (saves a ~5.5 us kernel in front of
// every call): the workgroup in launch slot `slot` picks the sequence whose length rank (descending,
// ties by index) is slot's rank under the fold of batch_prep_*_kernel (slot < B/2: rank = slot, else
// rank = B-1 - (slot - B/2), so that workgroups i and i + B/2 pair a long with a short sequence).
// Deterministic (a pure function of the lengths): every workgroup of a launch sees the same permutation.
// `scratch`: one int of LDS that nothing else uses yet.
__device__ __forceinline__ int folded_rank(int slot, int B) {
    const int half = B / 2;
    return slot < half ? slot : (B - 1) - (slot - half);
}

// Round 3: ONE wavefront does the whole selection in registers -- no LDS histogram, no atomics, two workgroup barriers instead of
// seven.  Lane l holds the lengths of sequences l, l + 64, ...; a binary search over the length finds the class of the rank
// (counts = popcounts of ballots), a ballot per register finds the sequence inside it.  (The histogram form cost every
// workgroup ~4.5 k cycles, 1.9 us of a 52 us full-length launch: FARNN_NOSORT A/B.)
template <int NV>
__device__ __forceinline__ int select_by_length_rank_wave(const int64_t *len, int B, int L, int rank, int lane, int *len_out = nullptr) {
    int v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int k = lane + 64 * i;
        int x = k < B ? (int)len[k] : -1;                // (-1: no sequence; below every class)
        if (k < B) x = x < 0 ? 0 : (x > L ? L : x);
        v[i] = x;
    }
    auto count_ge = [&](int t) {
        int c = 0;
#pragma unroll
        for (int i = 0; i < NV; i++) c += __popcll(__ballot(v[i] >= t));
        return c;
    };
    int lo = 0, hi = L;                                  // count_ge(lo) > rank holds (count_ge(0) = B > rank); the class is the largest such length
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (count_ge(mid) > rank) lo = mid; else hi = mid - 1;
    }
    const int cls = lo;
    if (len_out) *len_out = cls;                         // the selected sequence's (clamped) length: its class
    int rem = rank - count_ge(cls + 1);                  // index inside the class (sequences in index order: register-major, then lane)
    int res = 0;
    bool found = false;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const unsigned long long m = __ballot(v[i] == cls);
        const int c = __popcll(m);
        if (!found && rem < c) {
            const bool hit = ((m >> lane) & 1ull) && __popcll(m & ((1ull << lane) - 1ull)) == rem;
            res = 64 * i + (int)__builtin_ctzll(__ballot(hit) | (1ull << 63));
            found = true;
        } else if (!found) rem -= c;
    }
    return res;
}

// len_out (optional): the selected sequence's clamped length comes back with it -- the class the search found -- so the caller
// need not load len[b] behind the selection (a dependent global round trip at the launch's start, when every workgroup loads).
// LDS_ONLY: the two workgroup barriers order LDS traffic only (wg_barrier_lds) -- a __syncthreads() also drains the vector-memory
// counter, i.e. every wavefront would wait here for whatever global loads it has in flight (decomp_rows_kernel: the ~0.5 MB of
// register-resident weights it issues first -- round 5's "selection 10 k cycles" was that drain, not the selection).
template <bool LDS_ONLY = false>
__device__ __forceinline__ int select_by_length_rank(const int64_t *len, int B, int L, int rank, int *scratch,
                                                     int tid, int /*nthreads*/, int *len_out = nullptr) {
    if ((tid >> 6) == 0) {
        const int lane = tid & 63;
        int res, cls = 0;
        if (B <= 256) res = select_by_length_rank_wave<4>(len, B, L, rank, lane, &cls);
        else if (B <= 512) res = select_by_length_rank_wave<8>(len, B, L, rank, lane, &cls);
        else res = select_by_length_rank_wave<16>(len, B, L, rank, lane, &cls);       // B <= 1024
        if (lane == 0) { scratch[0] = res; scratch[1] = cls; }
    }
    if constexpr (LDS_ONLY) wg_barrier_lds(); else __syncthreads();
    const int b = scratch[0], l = scratch[1];
    if constexpr (LDS_ONLY) wg_barrier_lds(); else __syncthreads();      // the scratch is reused by the caller
    if (len_out) *len_out = __builtin_amdgcn_readfirstlane(l);
    return __builtin_amdgcn_readfirstlane(b);
}

}  // namespace farnn
