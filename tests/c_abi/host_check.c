/*
 * A host program in plain C against include/rag_hip.h: what a non-Python caller
 * (a cgo / JNI / C++ host) does with the library.  Builds a small corpus, checks
 * rag_index_search against a brute-force loop, exercises segments, soft delete,
 * reconstruct, the IVF entry points, score fusion, MMR, the several-devices index, the keyword
 * scorer and the error convention.
 * Exit code 0 = every check passed; prints the first failure otherwise.
 *
 *   gcc -O2 -std=c11 -I include tests/c_abi/host_check.c -o tests/c_abi/host_check \
 *       -L rag_amd -lrag_hip -Wl,-rpath,'$ORIGIN/../../rag_amd' -lm
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rag_hip.h"

#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                 \
            fprintf(stderr, " [%s]\n", rag_last_error()); \
            return 1;                                     \
        }                                                 \
    } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float frand(void) { /* xorshift, uniform in [-1, 1) */
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (float)((rng_state >> 40) / 8388608.0 - 1.0);
}

/* exact top-k by insertion: score descending, lower row first */
static void brute(const float* X, int n, int d, const float* q, int lo, int hi, const unsigned char* mask, float min_score,
                  int k, float* D, long long* I) {
    for (int j = 0; j < k; ++j) { D[j] = -FLT_MAX; I[j] = -1; }
    (void)n;
    for (int r = lo; r < hi; ++r) {
        if (mask && !mask[r]) continue;
        double acc = 0.0;
        for (int c = 0; c < d; ++c) acc += (double)X[(size_t)r * d + c] * q[c];
        float s = (float)acc;
        if (s < min_score) continue;
        int pos = k;
        while (pos > 0 && (I[pos - 1] < 0 || s > D[pos - 1])) --pos;
        if (pos == k) continue;
        for (int j = k - 1; j > pos; --j) { D[j] = D[j - 1]; I[j] = I[j - 1]; }
        D[pos] = s;
        I[pos] = r;
    }
}

static int same_hits(const float* D, const int64_t* I, const float* Dr, const long long* Ir, int k) {
    for (int j = 0; j < k; ++j) {
        if (I[j] != Ir[j]) {
            /* a different row is fine only when the scores tie within the fp32 tolerance */
            if (I[j] < 0 || Ir[j] < 0 || fabsf(D[j] - Dr[j]) > 1e-5f * fabsf(Dr[j]) + 1e-6f) return 0;
        } else if (I[j] >= 0 && fabsf(D[j] - Dr[j]) > 1e-5f * fabsf(Dr[j]) + 1e-6f) {
            return 0;
        }
    }
    return 1;
}

int main(void) {
    int ndev = 0;
    CHECK(rag_version() == RAG_ABI_VERSION, "ABI version");
    CHECK(rag_device_count(&ndev) == RAG_OK && ndev >= 1, "no device");

    const int n = 20000, d = 128, nq = 7, k = 10;
    float* X = malloc((size_t)n * d * sizeof(float));
    float* Q = malloc((size_t)nq * d * sizeof(float));
    CHECK(X && Q, "malloc");
    for (size_t i = 0; i < (size_t)n * d; ++i) X[i] = frand();
    for (size_t i = 0; i < (size_t)nq * d; ++i) Q[i] = frand();

    /* error convention: status code + message, nothing thrown, nothing written */
    rag_index_t* h = NULL;
    CHECK(rag_index_create(0, RAG_METRIC_IP, RAG_KIND_FLAT, 0, &h) == RAG_ERR_INVALID && h == NULL, "dim 0 accepted");
    CHECK(strlen(rag_last_error()) > 0, "no message");
    CHECK(rag_index_create(d, RAG_METRIC_IP, RAG_KIND_FLAT, 0, &h) == RAG_OK && h, "create");
    CHECK(rag_index_add(h, n / 2, X, NULL) == RAG_OK, "add 1");
    CHECK(rag_index_add(h, n - n / 2, X + (size_t)(n / 2) * d, NULL) == RAG_OK, "add 2");
    CHECK(rag_index_ntotal(h) == n && rag_index_dim(h) == d, "ntotal / dim");

    float D[7 * 10], Dr[10];
    int64_t I[7 * 10];
    long long Ir[10];
    CHECK(rag_index_search(h, nq, Q, 0, 0.f, 1, NULL, D, I) == RAG_ERR_INVALID, "k = 0 accepted");
    CHECK(rag_index_search(h, nq, Q, k, -FLT_MAX, 1, NULL, D, I) == RAG_OK, "search");
    for (int qi = 0; qi < nq; ++qi) {
        brute(X, n, d, Q + (size_t)qi * d, 0, n, NULL, -FLT_MAX, k, Dr, Ir);
        CHECK(same_hits(D + qi * k, I + qi * k, Dr, Ir, k), "query %d differs from the brute-force scan", qi);
    }
    /* one query, a threshold and a byte mask */
    unsigned char* mask = malloc(n);
    for (int r = 0; r < n; ++r) mask[r] = (unsigned char)(r % 3 != 0);
    CHECK(rag_index_search(h, 1, Q, k, 1.0f, 1, mask, D, I) == RAG_OK, "masked search");
    brute(X, n, d, Q, 0, n, mask, 1.0f, k, Dr, Ir);
    CHECK(same_hits(D, I, Dr, Ir, k), "masked query differs");
    /* the same bytes kept on the device as a filter object */
    rag_filter_t* flt = NULL;
    CHECK(rag_filter_create(0, n, mask, &flt) == RAG_OK && flt && rag_filter_rows(flt) == n && rag_filter_device_bytes(flt),
          "filter create");
    CHECK(rag_index_search_filtered(h, 1, Q, k, 1.0f, 1, flt, D, I) == RAG_OK, "filtered search");
    CHECK(same_hits(D, I, Dr, Ir, k), "filtered query differs");
    {
        rag_filter_t* small = NULL;
        CHECK(rag_filter_create(0, n - 1, mask, &small) == RAG_OK, "filter create (short)");
        CHECK(rag_index_search_filtered(h, 1, Q, k, 1.0f, 1, small, D, I) == RAG_ERR_INVALID, "filter of another row count accepted");
        CHECK(rag_filter_free(small) == RAG_OK, "filter free (short)");
        CHECK(rag_filter_create(0, 5, NULL, &small) == RAG_ERR_INVALID && small == NULL, "NULL filter bytes accepted");
    }
    CHECK(rag_filter_free(flt) == RAG_OK && rag_filter_free(NULL) == RAG_OK, "filter free");

    /* soft delete hides rows, restore brings them back */
    brute(X, n, d, Q, 0, n, NULL, -FLT_MAX, k, Dr, Ir);
    int64_t best = Ir[0];
    CHECK(rag_index_set_active(h, 1, &best, 0) == RAG_OK, "set_active 0");
    CHECK(rag_index_search(h, 1, Q, k, -FLT_MAX, 1, NULL, D, I) == RAG_OK && I[0] == Ir[1], "hidden row returned");
    CHECK(rag_index_set_active(h, 1, &best, 1) == RAG_OK, "set_active 1");
    CHECK(rag_index_search(h, 1, Q, k, -FLT_MAX, 1, NULL, D, I) == RAG_OK && I[0] == Ir[0], "restored row missing");

    /* segments: the k best of every (query, row range) pair in one pass */
    int64_t sb[3] = {0, 5000, 19990}, se[3] = {5000, 19990, 20000};
    float Ds[2 * 3 * 10];
    int64_t Is[2 * 3 * 10];
    CHECK(rag_index_search_segments(h, 2, Q, 3, sb, se, k, -FLT_MAX, NULL, Ds, Is) == RAG_OK, "segments");
    for (int qi = 0; qi < 2; ++qi)
        for (int s = 0; s < 3; ++s) {
            brute(X, n, d, Q + (size_t)qi * d, (int)sb[s], (int)se[s], NULL, -FLT_MAX, k, Dr, Ir);
            CHECK(same_hits(Ds + (qi * 3 + s) * k, Is + (qi * 3 + s) * k, Dr, Ir, k), "segment (%d, %d) differs", qi, s);
        }
    /* the same lists weighted, reduced to the best per row and ranked over all segments on the device:
     * against a host fold of the lists above in the reference's loop order (segment, query, rank) */
    {
        const double wq[2] = {0.4, 0.6};
        const int limit = 12;
        int64_t fr[12];
        double fs[12];
        int32_t fq[12], fn = -1;
        CHECK(rag_index_search_segments_fused(h, 2, Q, wq, 3, sb, se, k, -FLT_MAX, NULL, limit, fr, fs, fq, &fn) == RAG_OK,
              "fused segments");
        long long er[60];
        double es[60];
        int eq[60], ne = 0;
        for (int s = 0; s < 3; ++s)
            for (int qi = 0; qi < 2; ++qi)
                for (int j = 0; j < k; ++j) {
                    const int64_t r = Is[(qi * 3 + s) * k + j];
                    if (r < 0) continue;
                    const double w = (double)Ds[(qi * 3 + s) * k + j] * wq[qi];
                    int at = -1;
                    for (int e = 0; e < ne; ++e)
                        if (er[e] == r) at = e;
                    if (at < 0) { er[ne] = r; es[ne] = w; eq[ne] = qi; ++ne; }
                    else if (w > es[at]) { es[at] = w; eq[at] = qi; }
                }
        CHECK(fn == (ne < limit ? ne : limit), "fused count %d", (int)fn);
        unsigned char used[60] = {0};
        for (int t = 0; t < fn; ++t) {            /* stable selection: best score, earliest entry on ties */
            int best_e = -1;
            for (int e = 0; e < ne; ++e)
                if (!used[e] && (best_e < 0 || es[e] > es[best_e])) best_e = e;
            used[best_e] = 1;
            CHECK(fr[t] == er[best_e] && fs[t] == es[best_e] && fq[t] == eq[best_e], "fused entry %d differs", t);
        }
        CHECK(rag_index_search_segments_fused(h, 2, Q, wq, 3, sb, se, 600, -FLT_MAX, NULL, limit, fr, fs, fq, &fn) == RAG_ERR_INVALID,
              "nq * k above 1024 accepted");
    }
    se[2] = 20001;
    CHECK(rag_index_search_segments(h, 2, Q, 3, sb, se, k, -FLT_MAX, NULL, Ds, Is) == RAG_ERR_INVALID, "bad range accepted");

    /* reconstruct returns the stored rows bit for bit */
    int64_t rows[2] = {3, 19999};
    float back[2 * 128];
    CHECK(rag_index_reconstruct(h, 2, rows, back) == RAG_OK, "reconstruct");
    CHECK(memcmp(back, X + 3 * d, d * sizeof(float)) == 0 && memcmp(back + d, X + (size_t)19999 * d, d * sizeof(float)) == 0,
          "reconstructed rows differ");

    /* MMR over rows of the index: lambda 1 is the relevance order */
    int64_t cand[5];
    double rel[5] = {0.2, 1.0, 0.0, 0.7, 0.5};
    for (int j = 0; j < 5; ++j) cand[j] = I[j];
    int32_t pick[5], npick = 0;
    CHECK(rag_mmr_select_rows(h, 5, cand, rel, 1.0, 0.8, RAG_MMR_COSINE, 3, pick, &npick) == RAG_OK, "mmr rows");
    CHECK(npick == 3 && pick[0] == 1 && pick[1] == 3 && pick[2] == 4, "mmr order %d %d %d", pick[0], pick[1], pick[2]);
    /* pair similarities of the two reconstructed rows: unit rows, so the cosine diagonal is 1 and the matrix symmetric */
    double S2[4];
    CHECK(rag_similarity_matrix(2, d, back, 0, RAG_MMR_COSINE, 0, S2) == RAG_OK, "similarity matrix");
    CHECK(fabs(S2[0] - 1.0) < 1e-12 && fabs(S2[3] - 1.0) < 1e-12 && S2[1] == S2[2] && fabs(S2[1]) < 1.0, "similarity values %g %g", S2[0], S2[1]);
    CHECK(rag_similarity_matrix(2, d, back, 0, 7, 0, S2) == RAG_ERR_INVALID, "unknown similarity metric accepted");
    CHECK(rag_index_free(h) == RAG_OK, "free");

    /* L2 metric: squared distances, nearest first; a stored row is at distance 0 from itself; the threshold argument is
       the largest squared distance returned */
    rag_index_t* l2 = NULL;
    CHECK(rag_index_create(d, RAG_METRIC_L2, RAG_KIND_FLAT, 0, &l2) == RAG_OK, "create l2");
    CHECK(rag_index_add(l2, 5000, X, NULL) == RAG_OK, "add l2");
    float D2[4];
    int64_t I2[4];
    CHECK(rag_index_search(l2, 1, X + (size_t)77 * d, 4, -1.0f, 1, NULL, D2, I2) == RAG_OK, "search l2");
    CHECK(I2[0] == 77 && D2[0] == 0.0f && D2[1] > 0.0f && D2[1] <= D2[2] && D2[2] <= D2[3], "l2 order %lld %g %g", (long long)I2[0], D2[0], D2[1]);
    double want = 0.0;
    for (int j = 0; j < d; ++j) { const double f = (double)X[(size_t)77 * d + j] - (double)X[(size_t)I2[1] * d + j]; want += f * f; }
    CHECK(fabs(D2[1] - want) <= 1e-5 * want + 1e-6, "l2 distance %g vs %g", D2[1], want);
    CHECK(rag_index_search(l2, 1, X + (size_t)77 * d, 4, 0.0f, 1, NULL, D2, I2) == RAG_OK, "search l2 limit");
    CHECK(I2[0] == 77 && I2[1] == -1 && D2[1] == FLT_MAX, "l2 limit 0 keeps %lld %lld", (long long)I2[0], (long long)I2[1]);
    CHECK(rag_index_create(d, RAG_METRIC_L2, RAG_KIND_IVF, 0, &h) == RAG_ERR_UNSUPPORTED, "l2 ivf accepted");
    CHECK(rag_index_free(l2) == RAG_OK, "free l2");

    /* IVF: every list probed = the exhaustive answer */
    const int nlist = 16;
    CHECK(rag_index_create(d, RAG_METRIC_IP, RAG_KIND_IVF, 0, &h) == RAG_OK, "create ivf");
    CHECK(rag_index_train_ivf(h, nlist, X, 0, NULL) == RAG_OK, "train");       /* the first 16 rows are the centroids */
    CHECK(rag_index_add(h, n, X, NULL) == RAG_OK, "ivf add");
    CHECK(rag_index_search(h, nq, Q, k, -FLT_MAX, nlist, NULL, D, I) == RAG_OK, "ivf search");
    for (int qi = 0; qi < nq; ++qi) {
        brute(X, n, d, Q + (size_t)qi * d, 0, n, NULL, -FLT_MAX, k, Dr, Ir);
        CHECK(same_hits(D + qi * k, I + qi * k, Dr, Ir, k), "ivf query %d differs", qi);
    }
    CHECK(strncmp(rag_index_last_path(h), "ivf", 3) == 0, "path %s", rag_index_last_path(h));
    CHECK(rag_index_free(h) == RAG_OK, "free ivf");

    /* content / context fusion: 0.7 a + 0.3 b, threshold, stable order */
    double a[4] = {0.9, 0.5, 0.0, 0.5}, b[4] = {0.0, 0.8, 0.4, 0.8}, out[4];
    int32_t order[4], nout = 0;
    CHECK(rag_fuse_weighted(4, a, b, 0.7, 0.3, 1, 0.2, 3, order, out, &nout) == RAG_OK, "fuse");
    CHECK(nout == 3 && order[0] == 0 && order[1] == 1 && order[2] == 3, "fusion order");
    CHECK(out[0] == 0.9 * 0.7 + 0.0 * 0.3 && out[1] == 0.5 * 0.7 + 0.8 * 0.3 && out[2] == 0.0 * 0.7 + 0.4 * 0.3, "fusion scores");

    /* one index over several devices of this process (here: three shards on device 0) */
    {
        rag_multi_t* m = NULL;
        int devs[3] = {0, 0, 0};
        CHECK(rag_multi_create(d, RAG_METRIC_IP, RAG_KIND_FLAT, 0, devs, &m) == RAG_ERR_INVALID && m == NULL, "0 devices accepted");
        CHECK(rag_multi_create(d, RAG_METRIC_IP, RAG_KIND_FLAT, 3, devs, &m) == RAG_OK && m, "multi create");
        CHECK(rag_multi_add(m, 7001, X, NULL) == RAG_OK, "multi add 1");
        CHECK(rag_multi_add(m, n - 7001, X + (size_t)7001 * d, NULL) == RAG_OK, "multi add 2");
        CHECK(rag_multi_ntotal(m) == n && rag_multi_ndevices(m) == 3, "multi ntotal");
        CHECK(rag_multi_shard_rows(m, 0) + rag_multi_shard_rows(m, 1) + rag_multi_shard_rows(m, 2) == n, "shard rows");
        CHECK(rag_multi_search(m, nq, Q, k, -FLT_MAX, 1, NULL, D, I) == RAG_OK, "multi search");
        for (int qi = 0; qi < nq; ++qi) {
            brute(X, n, d, Q + (size_t)qi * d, 0, n, NULL, -FLT_MAX, k, Dr, Ir);
            CHECK(same_hits(D + qi * k, I + qi * k, Dr, Ir, k), "multi query %d differs from the brute-force scan", qi);
        }
        CHECK(rag_multi_search(m, 1, Q, k, 1.0f, 1, mask, D, I) == RAG_OK, "multi masked search");
        brute(X, n, d, Q, 0, n, mask, 1.0f, k, Dr, Ir);
        CHECK(same_hits(D, I, Dr, Ir, k), "multi masked query differs");
        brute(X, n, d, Q, 0, n, NULL, -FLT_MAX, k, Dr, Ir);
        best = Ir[0];
        CHECK(rag_multi_set_active(m, 1, &best, 0) == RAG_OK, "multi set_active");
        CHECK(rag_multi_search(m, 1, Q, k, -FLT_MAX, 1, NULL, D, I) == RAG_OK && I[0] == Ir[1], "multi hidden row returned");
        int64_t mrows[3] = {3, 7001, 19999};
        float mback[3 * 128];
        CHECK(rag_multi_reconstruct(m, 3, mrows, mback) == RAG_OK, "multi reconstruct");
        for (int j = 0; j < 3; ++j)
            CHECK(memcmp(mback + j * d, X + (size_t)mrows[j] * d, d * sizeof(float)) == 0, "multi reconstructed row %d differs", j);
        mrows[0] = n;
        CHECK(rag_multi_reconstruct(m, 1, mrows, mback) == RAG_ERR_INVALID, "row out of range accepted");
        CHECK(rag_multi_free(m) == RAG_OK, "multi free");
    }

    /* keyword side: BM25 over token ids, ranking */
    {
        rag_bm25_t* kw = NULL;
        CHECK(rag_bm25_create(0, &kw) == RAG_OK && kw, "bm25 create");
        /* three documents: [5 5 7 9 2], [1 1 1], [] */
        int64_t dlen[3] = {5, 3, 0};
        int32_t toks[8] = {5, 5, 7, 9, 2, 1, 1, 1};
        CHECK(rag_bm25_add(kw, 3, dlen, toks) == RAG_OK && rag_bm25_ndocs(kw) == 3, "bm25 add");
        int32_t terms[2] = {5, 7};
        double idf[2] = {1.2, 0.7}, raw[3];
        CHECK(rag_bm25_score(kw, 3, NULL, 2, terms, idf, 5.0, 1.5, 0.75, raw) == RAG_OK, "bm25 score");
        /* vector_db/search_manager.py:171-176 with f = 2 and 1, dl = avgdl = 5 */
        double len = 1.5 * (1 - 0.75 + 0.75 * (5.0 / 5.0));
        double want = 0.0;
        want += 1.2 * ((2 * (1.5 + 1)) / (2 + len));
        want += 0.7 * ((1 * (1.5 + 1)) / (1 + len));
        CHECK(raw[0] == want && raw[1] == 0.0 && raw[2] == 0.0, "bm25 scores %.17g vs %.17g", raw[0], want);
        int64_t krows[2];
        double knorm[2], kraw[2];
        int32_t kn = 0;
        CHECK(rag_bm25_rank(kw, 3, NULL, 2, terms, idf, 5.0, 1.5, 0.75, 1, 0.5, 2, krows, knorm, kraw, &kn) == RAG_OK, "bm25 rank");
        CHECK(kn == 1 && krows[0] == 0 && knorm[0] == 1.0 && kraw[0] == want && krows[1] == -1, "bm25 ranking");
        CHECK(rag_bm25_score(kw, 3, NULL, 2, terms, idf, 0.0, 1.5, 0.75, raw) == RAG_ERR_INVALID, "avgdl 0 accepted");
        rag_bm25_destroy(kw);
    }

    free(X);
    free(Q);
    free(mask);
    printf("c abi host check: ok\n");
    return 0;
}
