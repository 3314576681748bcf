/*
 * knn_oracle.c -- float64 brute-force top-k.  TEST INFRASTRUCTURE ONLY (oracle for
 * radad_knn_search; never linked into libradad_hip.so).
 *
 * Restates what faiss IndexFlatL2 / IndexFlatIP compute for RADAD's
 * VectorDatabase.search_batch (vector_database.py:159-182): squared-L2 ascending or inner product
 * descending, int64 ids in insertion order, unfilled slots id -1.  faiss itself is not in
 * /root/reference (requirements.txt:7,10 pull it from PyPI), so this is a restatement of its
 * published semantics -- "parity unpinned" by the reference; ties are broken by the lower id.
 *
 * Inputs are float32 (what the store holds); accumulation is float64, L2 uses the direct
 * sum((q-y)^2) form so there is no cancellation.
 *
 * build: gcc -O3 -fopenmp -shared -fPIC -o _build/libknn_oracle.so knn_oracle.c
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* better(a,b): does (da,ia) rank strictly before (db,ib)?  key is already "smaller is better". */
static inline int better(double da, int64_t ia, double db, int64_t ib) {
    return da < db || (da == db && ia < ib);
}

/* metric: 0 = L2 squared (ascending), 1 = inner product (descending).
 * db [n, d], q [nq, d] float32 row-major; out_dist float64 [nq,k]; out_idx int64 [nq,k].
 * id_base is added to the reported ids. */
int knn_oracle_f64(const float* db, int64_t n, const float* q, int64_t nq, int d, int k, int metric,
                   int64_t id_base, double* out_dist, int64_t* out_idx) {
    if (k <= 0 || d <= 0 || n < 0 || nq < 0) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t qi = 0; qi < nq; ++qi) {
        const float* qq = q + qi * (int64_t)d;
        double* bd = out_dist + qi * (int64_t)k;   /* keys while scanning, converted at the end */
        int64_t* bi = out_idx + qi * (int64_t)k;
        int filled = 0;
        for (int64_t r = 0; r < n; ++r) {
            const float* y = db + r * (int64_t)d;
            double acc = 0.0;
            if (metric == 0) {
                for (int c = 0; c < d; ++c) { double t = (double)qq[c] - (double)y[c]; acc += t * t; }
            } else {
                for (int c = 0; c < d; ++c) acc += (double)qq[c] * (double)y[c];
                acc = -acc;
            }
            if (filled == k && !better(acc, r, bd[k - 1], bi[k - 1])) continue;
            int pos = filled < k ? filled : k - 1;
            while (pos > 0 && better(acc, r, bd[pos - 1], bi[pos - 1])) {
                bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos;
            }
            bd[pos] = acc; bi[pos] = r;
            if (filled < k) ++filled;
        }
        for (int j = 0; j < k; ++j) {
            if (j < filled) { bd[j] = metric == 0 ? bd[j] : -bd[j]; bi[j] += id_base; }
            else { bd[j] = metric == 0 ? INFINITY : -INFINITY; bi[j] = -1; }
        }
    }
    return 0;
}
