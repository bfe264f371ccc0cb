/*
 * CPU ORACLE (C port) of the onehot i-FST tagging path -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, per-sequence restatement of FARNN_S_O_I_S.forward_score + local_decode
 * (reference src_seq/farnn/model_onehot.py:351-428, :162-180) over the VALID positions of every
 * sequence, with T+W hoisted out of the call ("fair" CPU variant of BASELINE.md section 3).
 * It exists for two things only: (1) a second, independent checker beside the numpy oracle
 * (tests/test_oracle_c.py pins it to the same reference fixtures) and (2) the `cpu_baseline`
 * leg of bench.py (kind "port"), parallelised with OpenMP over the 2B (sequence, direction) chains and
 * then over the B*L score rows, so that a host with more threads than sequences is still used.
 * Nothing under re2nn-seq_amd/ links or loads this file.
 *
 * Build: make -C oracle     (gcc -O3 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static float nlf(float v, int nl) {            /* model_onehot.py:379-386 */
    switch (nl) {
        case 1: return v > 0.f ? v : 0.f;
        case 2: return tanhf(v);
        case 3: return tanhf(v > 0.f ? v : 0.f);
        default: return v;
    }
}

/* Tf = T + W premixed [V,S,S]; O [C,S]; x [B,L]; len [B]; tags [B,L] (-1 at pads);
 * scores [B,L,C] or NULL (zero at pads). semiring: 0 sum, 1 max. Returns threads used.
 * reps > 1 (bench.py's cpu_baseline only): the same batch `reps` times inside ONE parallel region -- the thread team stays hot
 * between passes (libgomp's barriers spin), so the rate is the steady state of a serving loop on all host cores and not the
 * fork / wake-up latency of a 256-thread team per 2 ms batch (which is where the round-3 port stopped scaling, at 16 threads). */
int oracle_onehot_ifst_tag_reps(const float *Tf, const float *O, const float *h0, const float *hT,
                                int V, int S, int C, const int64_t *x, const int64_t *len, int B, int L,
                                int nl, int semiring, float threshold, int o_idx, int32_t *tags,
                                float *scores, int nthreads, int reps) {
    (void)V;
    /* the 2B chains longest first (counting sort by length): with dynamic scheduling no thread is handed a 64-step chain last */
    int *order = (int *)malloc(sizeof(int) * 2 * (size_t)B);
    {
        int *cnt = (int *)calloc((size_t)L + 2, sizeof(int));
        for (int b = 0; b < B; b++) { int n = (int)len[b]; n = n < 0 ? 0 : (n > L ? L : n); cnt[L - n + 1] += 2; }
        for (int i = 1; i <= L + 1; i++) cnt[i] += cnt[i - 1];
        for (int b = 0; b < B; b++) {
            int n = (int)len[b]; n = n < 0 ? 0 : (n > L ? L : n);
            order[cnt[L - n]++] = 2 * b; order[cnt[L - n]++] = 2 * b + 1;
        }
        free(cnt);
    }
    float *o = (float *)calloc((size_t)S, sizeof(float));
    for (int c = 0; c < C; c++)                        /* :368 (CE1) */
        for (int s = 0; s < S; s++) o[s] += O[(size_t)c * S + s];
    int used = 1;
    /* every forward / backward state of every sequence: a[b][k] after k tokens, bt[b][k] before token k */
    const size_t stride = (size_t)(L + 1) * S;
    float *A = (float *)malloc(sizeof(float) * stride * (size_t)B);
    float *BT = (float *)malloc(sizeof(float) * stride * (size_t)B);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel
#endif
    {
        float *tmp = (float *)malloc(sizeof(float) * (size_t)S);
        float *sc = (float *)malloc(sizeof(float) * (size_t)C);
        for (int rep_ = 0; rep_ < (reps > 1 ? reps : 1); rep_++) {
        /* phase 1: the 2B independent chains (sequence x direction), longest first would be the GPU's order;
         * dynamic scheduling balances the ragged lengths here */
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int it_ = 0; it_ < 2 * B; it_++) {
            const int item = order[it_];
            const int b = item >> 1, dir = item & 1;
            const int n = (int)len[b];
            const int64_t *xb = x + (size_t)b * L;
            float *a = A + stride * b, *bt = BT + stride * b;
            if (dir == 0) {
                memcpy(a, h0, sizeof(float) * S);
                for (int t = 0; t < n; t++) {               /* :374-387 */
                    const float *M = Tf + (size_t)xb[t] * S * S;
                    const float *ai = a + (size_t)t * S;
                    float *ao = a + (size_t)(t + 1) * S;
                    for (int j = 0; j < S; j++) ao[j] = semiring ? -INFINITY : 0.f;
                    for (int s = 0; s < S; s++) {
                        const float hs = ai[s];
                        const float *row = M + (size_t)s * S;
                        if (semiring) { for (int j = 0; j < S; j++) { float p = hs * row[j]; if (p > ao[j]) ao[j] = p; } }
                        else          { for (int j = 0; j < S; j++) ao[j] += hs * row[j]; }
                    }
                    for (int j = 0; j < S; j++) ao[j] = nlf(ao[j] * o[j], nl);
                }
            } else {
                memcpy(bt + (size_t)n * S, hT, sizeof(float) * S);
                for (int t = n - 1; t >= 0; t--) {          /* :390-403 */
                    const float *M = Tf + (size_t)xb[t] * S * S;
                    const float *bi = bt + (size_t)(t + 1) * S;
                    float *bo = bt + (size_t)t * S;
                    for (int j = 0; j < S; j++) tmp[j] = bi[j] * o[j];
                    for (int s = 0; s < S; s++) {
                        const float *row = M + (size_t)s * S;
                        float acc = semiring ? -INFINITY : 0.f;
                        if (semiring) { for (int j = 0; j < S; j++) { float p = tmp[j] * row[j]; if (p > acc) acc = p; } }
                        else          { for (int j = 0; j < S; j++) acc += row[j] * tmp[j]; }
                        bo[s] = nlf(acc, nl);
                    }
                }
            }
        }
        /* phase 2 (after the implicit barrier): scores + decode, one (sequence, position) row at a time */
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
        for (int bt_i = 0; bt_i < B * L; bt_i++) {
            const int b = bt_i / L, t = bt_i - b * L;
            const int n = (int)len[b];
            const float *a = A + stride * b, *bt = BT + stride * b;
            int32_t *tg = tags ? tags + (size_t)b * L + t : NULL;
            float *so = scores ? scores + ((size_t)b * L + t) * C : NULL;
            if (t >= n) {
                if (tg) *tg = -1;
                if (so) memset(so, 0, sizeof(float) * C);
                continue;
            }
            const float *at = a + (size_t)(t + 1) * S, *bb = bt + (size_t)(t + 1) * S;
            for (int s = 0; s < S; s++) tmp[s] = at[s] * bb[s];           /* :347 */
            int best = 0; float bv = -INFINITY;
            for (int c = 0; c < C; c++) {
                const float *orow = O + (size_t)c * S;
                float acc = 0.f;
                for (int s = 0; s < S; s++) acc += orow[s] * tmp[s];       /* :348 */
                sc[c] = acc;
                float v = (c == C - 1 && acc > threshold) ? threshold : acc;   /* :167 */
                if (v > bv) { bv = v; best = c; }                          /* first max (:168) */
            }
            if (so) memcpy(so, sc, sizeof(float) * C);
            if (tg) *tg = (best == C - 1) ? o_idx : best;                  /* :169 */
        }
        }   /* reps (the omp for's implicit barrier separates a pass's scoring from the next pass's chains) */
        free(tmp); free(sc);
    }
    free(A); free(BT);
    free(o); free(order);
    return used;
}

int oracle_onehot_ifst_tag(const float *Tf, const float *O, const float *h0, const float *hT,
                           int V, int S, int C, const int64_t *x, const int64_t *len, int B, int L,
                           int nl, int semiring, float threshold, int o_idx, int32_t *tags,
                           float *scores, int nthreads) {
    return oracle_onehot_ifst_tag_reps(Tf, O, h0, hT, V, S, C, x, len, B, L, nl, semiring, threshold, o_idx, tags, scores, nthreads, 1);
}

/* The same tagging as a THROUGHPUT loop over whole sequences (bench.py's cpu_baseline, round 5): `reps` passes of the batch are
 * reps x B independent items -- one sequence each: its forward chain, its backward chain, its score rows and their decode, with the
 * thread's own two state histories -- dealt to the threads dynamically, longest sequences first within a pass, with NO barrier
 * anywhere.  The two-phase form above stops scaling at 16-32 threads (two team barriers per 2 ms pass, a 64-step chain as the
 * critical path of every pass); a serving loop on all host cores is bound by neither.  Same arithmetic, same tags (tests/test_oracle_c.py). */
int oracle_onehot_ifst_tag_stream(const float *Tf, const float *O, const float *h0, const float *hT,
                                  int V, int S, int C, const int64_t *x, const int64_t *len, int B, int L,
                                  int nl, int semiring, float threshold, int o_idx, int32_t *tags,
                                  float *scores, int nthreads, int reps) {
    (void)V;
    if (reps < 1) reps = 1;
    int *order = (int *)malloc(sizeof(int) * (size_t)B);          /* sequences, longest first */
    {
        int *cnt = (int *)calloc((size_t)L + 2, sizeof(int));
        for (int b = 0; b < B; b++) { int n = (int)len[b]; n = n < 0 ? 0 : (n > L ? L : n); cnt[L - n + 1]++; }
        for (int i = 1; i <= L + 1; i++) cnt[i] += cnt[i - 1];
        for (int b = 0; b < B; b++) { int n = (int)len[b]; n = n < 0 ? 0 : (n > L ? L : n); order[cnt[L - n]++] = b; }
        free(cnt);
    }
    float *o = (float *)calloc((size_t)S, sizeof(float));
    for (int c = 0; c < C; c++)                        /* :368 (CE1) */
        for (int s = 0; s < S; s++) o[s] += O[(size_t)c * S + s];
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel
#endif
    {
        const size_t stride = (size_t)(L + 1) * S;
        float *a = (float *)malloc(sizeof(float) * stride), *bt = (float *)malloc(sizeof(float) * stride);
        float *tmp = (float *)malloc(sizeof(float) * (size_t)S), *sc = (float *)malloc(sizeof(float) * (size_t)C);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1) nowait
#endif
        for (long long it_ = 0; it_ < (long long)reps * B; it_++) {
            const int rep_ = (int)(it_ / B), b = order[it_ % B];
            int n = (int)len[b]; n = n < 0 ? 0 : (n > L ? L : n);
            const int64_t *xb = x + (size_t)b * L;
            memcpy(a, h0, sizeof(float) * S);
            for (int t = 0; t < n; t++) {               /* :374-387 */
                const float *M = Tf + (size_t)xb[t] * S * S;
                const float *ai = a + (size_t)t * S;
                float *ao = a + (size_t)(t + 1) * S;
                for (int j = 0; j < S; j++) ao[j] = semiring ? -INFINITY : 0.f;
                for (int s = 0; s < S; s++) {
                    const float hs = ai[s];
                    const float *row = M + (size_t)s * S;
                    if (semiring) { for (int j = 0; j < S; j++) { float p = hs * row[j]; if (p > ao[j]) ao[j] = p; } }
                    else          { for (int j = 0; j < S; j++) ao[j] += hs * row[j]; }
                }
                for (int j = 0; j < S; j++) ao[j] = nlf(ao[j] * o[j], nl);
            }
            memcpy(bt + (size_t)n * S, hT, sizeof(float) * S);
            for (int t = n - 1; t >= 0; t--) {          /* :390-403 */
                const float *M = Tf + (size_t)xb[t] * S * S;
                const float *bi = bt + (size_t)(t + 1) * S;
                float *bo = bt + (size_t)t * S;
                for (int j = 0; j < S; j++) tmp[j] = bi[j] * o[j];
                for (int s = 0; s < S; s++) {
                    const float *row = M + (size_t)s * S;
                    float acc = semiring ? -INFINITY : 0.f;
                    if (semiring) { for (int j = 0; j < S; j++) { float p = tmp[j] * row[j]; if (p > acc) acc = p; } }
                    else          { for (int j = 0; j < S; j++) acc += row[j] * tmp[j]; }
                    bo[s] = nlf(acc, nl);
                }
            }
            for (int t = 0; t < L; t++) {
                int32_t *tg = (tags && rep_ == 0) ? tags + (size_t)b * L + t : NULL;     /* (every pass computes; the first one stores) */
                float *so = (scores && rep_ == 0) ? scores + ((size_t)b * L + t) * C : NULL;
                if (t >= n) {
                    if (tg) *tg = -1;
                    if (so) memset(so, 0, sizeof(float) * C);
                    continue;
                }
                const float *at = a + (size_t)(t + 1) * S, *bb = bt + (size_t)(t + 1) * S;
                for (int s = 0; s < S; s++) tmp[s] = at[s] * bb[s];           /* :347 */
                int best = 0; float bv = -INFINITY;
                for (int c = 0; c < C; c++) {
                    const float *orow = O + (size_t)c * S;
                    float acc = 0.f;
                    for (int s = 0; s < S; s++) acc += orow[s] * tmp[s];       /* :348 */
                    sc[c] = acc;
                    float v = (c == C - 1 && acc > threshold) ? threshold : acc;   /* :167 */
                    if (v > bv) { bv = v; best = c; }                          /* first max (:168) */
                }
                if (so) memcpy(so, sc, sizeof(float) * C);
                const int tagv = (best == C - 1) ? o_idx : best;               /* :169 */
                if (tg) *tg = tagv;
                else if (tagv == 0x7fffffff) tmp[0] = 0.f;                     /* (the decode of a later pass is not dead code) */
            }
        }
        free(a); free(bt); free(tmp); free(sc);
    }
    free(o); free(order);
    return used;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
