// Keyword side of hybrid search on the device: BM25 over a token-id corpus held in HBM,
// min-max normalisation, score filter and ranking.
//
// vector_db/search_manager.py:142-180 (CorpusStatsManager.calculate_bm25_score) tokenises the
// candidate text again for every query.  Here a document is tokenised once, when it is stored:
// its kept tokens (preprocess_query, :57-66) become vocabulary ids in one CSR array, a query
// brings only its term ids and idf values, and the device counts term frequencies by comparing
// ids.  The arithmetic is fp64 with the reference's operation order and no fused multiply-add,
// so raw scores, the min-max normalisation of :240-252 and the ranking of :1200-1213 are
// bit-identical to the Python floats.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <cmath>

#include <mutex>
#include <string>
#include <vector>

#include "index.h"
#include "sort_entries.h"

struct rag_bm25 {
    int device = 0;
    long long n_docs = 0, n_tokens = 0;
    long long cap_docs = 0, cap_tokens = 0;
    int32_t* tokens = nullptr;        // [n_tokens] vocabulary ids, document after document
    long long* doc_begin = nullptr;   // [n_docs + 1]
    rag::Workspace ws;                // staging of one call (host-pointer entry points)
    std::mutex mu;
};

namespace rag {

constexpr int BM25_MAX_TERMS = 64;
constexpr int BM25_MAX_RANK = 4096;

struct Bm25Args {
    const int32_t* tokens;
    const long long* doc_begin;
    long long n_docs;
    const long long* cand;     // [m] documents to score, NULL: documents 0..m-1
    long long m;
    const int32_t* term_ids;   // [T] (< 0: a term outside the vocabulary, matches nothing)
    const double* term_idf;    // [T]
    int T;
    double avgdl, k1, b;
    double* raw;               // [m]
};

// one wave per document, lane t = query term t.  64 tokens are loaded at a time (one per lane)
// and handed round with readlane, so every lane sees every token once.
__global__ __launch_bounds__(256) void bm25_score_kernel(Bm25Args p) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.m) return;
    const long long r = p.cand ? p.cand[i] : i;
    double score = 0.0;
    if (r >= 0 && r < p.n_docs) {
        const long long b0 = p.doc_begin[r];
        const long long dl = p.doc_begin[r + 1] - b0;
        const int my = lane < p.T ? p.term_ids[lane] : -1;
        int f = 0;
        for (long long j0 = 0; j0 < dl; j0 += 64) {
            const int tok = (j0 + lane < dl) ? p.tokens[b0 + j0 + lane] : -2;
            const int cnt = (dl - j0) < 64 ? (int)(dl - j0) : 64;
            for (int j = 0; j < cnt; ++j) f += (__shfl(tok, j, 64) == my && my >= 0) ? 1 : 0;
        }
        // idf * ((f * (k1 + 1)) / (f + k1 * (1 - b + b * (dl / avgdl)))), :171-176
        double part = 0.0;
        if (f > 0) {
            const double fd = (double)f;
            const double len = __dmul_rn(p.k1, __dadd_rn(__dsub_rn(1.0, p.b), __dmul_rn(p.b, __ddiv_rn((double)dl, p.avgdl))));
            part = __dmul_rn(p.term_idf[lane], __ddiv_rn(__dmul_rn(fd, __dadd_rn(p.k1, 1.0)), __dadd_rn(fd, len)));
        }
        // the terms that occur, summed in query order (:166-176)
        for (int t = 0; t < p.T; ++t) {
            const int ft = __shfl(f, t, 64);
            const double pt = __shfl(part, t, 64);
            if (ft > 0) score = __dadd_rn(score, pt);
        }
        if (0.0 > score) score = 0.0;          // max(score, 0.0), :178
    }
    if (lane == 0) p.raw[i] = score;
}

struct RankArgs {
    const double* raw;         // [m]
    const long long* cand;     // [m] or NULL
    int m;
    int use_min;
    double min_score;
    int limit;
    long long* out_rows;       // [limit] document rows, -1 beyond n_out
    double* out_norm;          // [limit]
    double* out_raw;           // [limit] or NULL
    double* norm_all;          // [m] or NULL: every candidate's normalised score
    int* n_out;
};

// normalize_scores(method="minmax") (:182-263), the min_score filter and the stable descending
// sort of :1200-1213, one workgroup
__global__ __launch_bounds__(1024) void bm25_rank_kernel(RankArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64& s_lo = *reinterpret_cast<u64*>(smem);
    u64& s_hi = *reinterpret_cast<u64*>(smem + 8);
    int& s_kept = *reinterpret_cast<int*>(smem + 16);
    int npad = 2;
    while (npad < p.m) npad <<= 1;
    SortEntry* e = reinterpret_cast<SortEntry*>(smem + 32);          // [npad]
    double* s_norm = reinterpret_cast<double*>(e + npad);            // [m]
    const int tid = threadIdx.x;
    if (tid == 0) { s_lo = ~0ull; s_hi = 0ull; s_kept = 0; }
    __syncthreads();
    for (int i = tid; i < p.m; i += 1024) {
        const u64 o = f64_ordered(p.raw[i]);
        atomicMin(&s_lo, o);
        atomicMax(&s_hi, o);
    }
    __syncthreads();
    // back from the ordered bits (f64_ordered maps -0.0 to +0.0: the arithmetic below does not tell them apart)
    const u64 lob = s_lo, hib = s_hi;
    const double lo = __longlong_as_double((long long)((lob >> 63) ? (lob & 0x7FFFFFFFFFFFFFFFull) : ~lob));
    const double hi = __longlong_as_double((long long)((hib >> 63) ? (hib & 0x7FFFFFFFFFFFFFFFull) : ~hib));
    for (int i = tid; i < npad; i += 1024) {
        SortEntry en;
        en.key = 0;
        en.idx = 0xFFFFFFFFu;
        en.pad = 0;
        if (i < p.m) {
            const double s = p.raw[i];
            double nv;
            if (p.m == 1) {
                // a single score goes through the sigmoid 1 / (1 + exp(-2 s)) (:195-197, :254-263)
                const double ex = exp(__dmul_rn(-2.0, s));
                nv = isinf(ex) ? (s < 0.0 ? 0.0 : 1.0) : __ddiv_rn(1.0, __dadd_rn(1.0, ex));
            } else if (hi == lo) {
                nv = hi == 0.0 ? 0.0 : 1.0;
            } else {
                nv = __ddiv_rn(__dsub_rn(s, lo), __dsub_rn(hi, lo));
            }
            s_norm[i] = nv;
            if (p.norm_all) p.norm_all[i] = nv;
            if ((!p.use_min || nv >= p.min_score) && nv == nv) {
                en.key = f64_ordered(nv);
                if (en.key == 0) en.key = 1;
                en.idx = (uint32_t)i;
                atomicAdd(&s_kept, 1);
            }
        }
        e[i] = en;
    }
    sort_entries(e, npad, tid, 1024);
    const int n = s_kept < p.limit ? s_kept : p.limit;
    for (int r = tid; r < p.limit; r += 1024) {
        if (r < n) {
            const int i = (int)e[r].idx;
            p.out_rows[r] = p.cand ? p.cand[i] : (long long)i;
            p.out_norm[r] = s_norm[i];
            if (p.out_raw) p.out_raw[r] = p.raw[i];
        } else {
            p.out_rows[r] = -1;
            p.out_norm[r] = 0.0;
            if (p.out_raw) p.out_raw[r] = 0.0;
        }
    }
    if (tid == 0) *p.n_out = n;
}

static size_t rank_lds(int m) {
    size_t npad = 2;
    while (npad < (size_t)m) npad <<= 1;
    return 32 + npad * sizeof(SortEntry) + (size_t)m * sizeof(double);
}

struct DeviceScope {
    int prev = -1, dev = -1;
    int enter(int device) {
        RAG_HIP_CHECK(hipGetDevice(&prev));
        dev = device;
        if (prev != dev) RAG_HIP_CHECK(hipSetDevice(dev));
        return RAG_OK;
    }
    ~DeviceScope() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

static int bm25_check(const rag_bm25* h, long long m, int T, const void* ids, const void* idf, double avgdl) {
    if (!h) return fail(RAG_ERR_INVALID, "keyword corpus is NULL");
    if (m < 0) return fail(RAG_ERR_INVALID, "m < 0");
    if (T < 1 || T > BM25_MAX_TERMS) return fail(RAG_ERR_INVALID, "1.." + std::to_string(BM25_MAX_TERMS) + " query terms");
    if (!ids || !idf) return fail(RAG_ERR_INVALID, "NULL buffer");
    // a zero average length raises ZeroDivisionError in the reference and sends it down its substring
    // fallback (:179-180), which needs the document text: host work
    if (!(avgdl > 0.0) && !(avgdl < 0.0)) return fail(RAG_ERR_INVALID, "avg_doc_length must be a non-zero number");
    return RAG_OK;
}

static int enqueue_score(const rag_bm25* h, long long m, const long long* cand_dev, int T, const int32_t* ids_dev,
                         const double* idf_dev, double avgdl, double k1, double b, double* raw_dev, hipStream_t s) {
    if (m == 0) return RAG_OK;
    Bm25Args a;
    a.tokens = h->tokens; a.doc_begin = h->doc_begin; a.n_docs = h->n_docs;
    a.cand = cand_dev; a.m = m; a.term_ids = ids_dev; a.term_idf = idf_dev; a.T = T;
    a.avgdl = avgdl; a.k1 = k1; a.b = b; a.raw = raw_dev;
    const long long blocks = (m + 3) / 4;
    if (blocks > 0x7FFFFFFFLL) return fail(RAG_ERR_INVALID, "too many candidates");
    hipLaunchKernelGGL(bm25_score_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    RAG_HIP_CHECK(hipGetLastError());
    return RAG_OK;
}

static int enqueue_rank(long long m, const long long* cand_dev, const double* raw_dev, int use_min, double min_score, int limit,
                        long long* out_rows, double* out_norm, double* out_raw, double* norm_all, int* n_out, hipStream_t s) {
    RankArgs r;
    r.raw = raw_dev; r.cand = cand_dev; r.m = (int)m; r.use_min = use_min; r.min_score = min_score; r.limit = limit;
    r.out_rows = out_rows; r.out_norm = out_norm; r.out_raw = out_raw; r.norm_all = norm_all; r.n_out = n_out;
    const size_t lds = rank_lds((int)m);
    if (lds > 48 * 1024)
        RAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bm25_rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(bm25_rank_kernel, dim3(1), dim3(1024), lds, s, r);
    RAG_HIP_CHECK(hipGetLastError());
    return RAG_OK;
}

static int rank_check(long long m, long long n_docs, bool all_docs, int limit, const void* rows, const void* norm, const void* n_out) {
    if (limit < 1 || limit > BM25_MAX_RANK) return fail(RAG_ERR_INVALID, "limit must be in [1, " + std::to_string(BM25_MAX_RANK) + "]");
    if (m > BM25_MAX_RANK) return fail(RAG_ERR_INVALID, "ranking handles at most " + std::to_string(BM25_MAX_RANK) + " candidates");
    if (all_docs && m > n_docs) return fail(RAG_ERR_INVALID, "m exceeds the number of documents");
    if (!rows || !norm || !n_out) return fail(RAG_ERR_INVALID, "NULL buffer");
    return RAG_OK;
}

// grow-only device arrays of the corpus
static int reserve(rag_bm25* h, long long docs, long long toks) {
    if (docs > h->cap_docs) {
        long long cap = std::max<long long>(docs, h->cap_docs * 2);
        long long* nb = nullptr;
        RAG_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&nb), (size_t)(cap + 1) * sizeof(long long)));
        if (h->doc_begin) {
            RAG_HIP_CHECK(hipMemcpy(nb, h->doc_begin, (size_t)(h->n_docs + 1) * sizeof(long long), hipMemcpyDeviceToDevice));
            (void)hipFree(h->doc_begin);
        } else {
            RAG_HIP_CHECK(hipMemset(nb, 0, sizeof(long long)));
        }
        h->doc_begin = nb;
        h->cap_docs = cap;
    }
    if (toks > h->cap_tokens) {
        long long cap = std::max<long long>(toks, h->cap_tokens * 2);
        int32_t* nt = nullptr;
        RAG_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&nt), (size_t)cap * sizeof(int32_t)));
        if (h->tokens) {
            RAG_HIP_CHECK(hipMemcpy(nt, h->tokens, (size_t)h->n_tokens * sizeof(int32_t), hipMemcpyDeviceToDevice));
            (void)hipFree(h->tokens);
        }
        h->tokens = nt;
        h->cap_tokens = cap;
    }
    return RAG_OK;
}

}  // namespace rag

using namespace rag;

extern "C" {

int rag_bm25_create(int device, rag_bm25_t** out) {
    if (!out) return fail(RAG_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int count = 0;
    RAG_HIP_CHECK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(RAG_ERR_INVALID, "no such device");
    rag_bm25* h = new rag_bm25();
    h->device = device;
    DeviceScope sc;
    if (int rc = sc.enter(device)) { delete h; return rc; }
    if (int rc = reserve(h, 1, 1)) { delete h; return rc; }
    *out = h;
    return RAG_OK;
}

void rag_bm25_destroy(rag_bm25_t* h) {
    if (!h) return;
    DeviceScope sc;
    if (sc.enter(h->device) == RAG_OK) {
        if (h->tokens) (void)hipFree(h->tokens);
        if (h->doc_begin) (void)hipFree(h->doc_begin);
        h->ws.release();
    }
    delete h;
}

int64_t rag_bm25_ndocs(const rag_bm25_t* h) { return h ? h->n_docs : 0; }

int rag_bm25_add(rag_bm25_t* h, int64_t n_new, const int64_t* doc_len, const int32_t* tokens) {
    if (!h) return fail(RAG_ERR_INVALID, "keyword corpus is NULL");
    if (n_new < 0) return fail(RAG_ERR_INVALID, "n_new < 0");
    if (n_new == 0) return RAG_OK;
    if (!doc_len) return fail(RAG_ERR_INVALID, "NULL buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    std::vector<long long> begin((size_t)n_new);
    long long total = h->n_tokens;
    for (int64_t i = 0; i < n_new; ++i) {
        if (doc_len[i] < 0) return fail(RAG_ERR_INVALID, "negative document length");
        total += doc_len[i];
        begin[(size_t)i] = total;              // end of document i = begin of i + 1
    }
    const long long add = total - h->n_tokens;
    if (add > 0 && !tokens) return fail(RAG_ERR_INVALID, "NULL buffer");
    DeviceScope sc;
    if (int rc = sc.enter(h->device)) return rc;
    if (int rc = reserve(h, h->n_docs + n_new, total)) return rc;
    if (add > 0)
        RAG_HIP_CHECK(hipMemcpy(h->tokens + h->n_tokens, tokens, (size_t)add * sizeof(int32_t), hipMemcpyHostToDevice));
    RAG_HIP_CHECK(hipMemcpy(h->doc_begin + h->n_docs + 1, begin.data(), (size_t)n_new * sizeof(long long), hipMemcpyHostToDevice));
    h->n_docs += n_new;
    h->n_tokens = total;
    return RAG_OK;
}

int rag_bm25_clear(rag_bm25_t* h) {
    if (!h) return fail(RAG_ERR_INVALID, "keyword corpus is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    h->n_docs = 0;
    h->n_tokens = 0;
    return RAG_OK;
}

int rag_bm25_score_device(rag_bm25_t* h, int64_t m, const int64_t* cand_dev, int n_terms, const int32_t* term_ids_dev,
                          const double* term_idf_dev, double avgdl, double k1, double b, double* raw_out_dev, void* stream) {
    if (int rc = bm25_check(h, m, n_terms, term_ids_dev, term_idf_dev, avgdl)) return rc;
    if (m > 0 && !raw_out_dev) return fail(RAG_ERR_INVALID, "NULL buffer");
    if (!cand_dev && m > h->n_docs) return fail(RAG_ERR_INVALID, "m exceeds the number of documents");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceScope sc;
    if (int rc = sc.enter(h->device)) return rc;
    return enqueue_score(h, m, reinterpret_cast<const long long*>(cand_dev), n_terms, term_ids_dev, term_idf_dev, avgdl, k1, b,
                         raw_out_dev, static_cast<hipStream_t>(stream));
}

int rag_bm25_rank_device(rag_bm25_t* h, int64_t m, const int64_t* cand_dev, int n_terms, const int32_t* term_ids_dev,
                         const double* term_idf_dev, double avgdl, double k1, double b, int use_min_score, double min_score,
                         int limit, double* raw_scratch_dev, int64_t* out_rows_dev, double* out_norm_dev, double* out_raw_dev,
                         int32_t* n_out_dev, void* stream) {
    if (int rc = bm25_check(h, m, n_terms, term_ids_dev, term_idf_dev, avgdl)) return rc;
    if (int rc = rank_check(m, h->n_docs, cand_dev == nullptr, limit, out_rows_dev, out_norm_dev, n_out_dev)) return rc;
    if (m > 0 && !raw_scratch_dev) return fail(RAG_ERR_INVALID, "NULL buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceScope sc;
    if (int rc = sc.enter(h->device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = enqueue_score(h, m, reinterpret_cast<const long long*>(cand_dev), n_terms, term_ids_dev, term_idf_dev, avgdl, k1, b,
                               raw_scratch_dev, s)) return rc;
    return enqueue_rank(m, reinterpret_cast<const long long*>(cand_dev), raw_scratch_dev, use_min_score, min_score, limit,
                        reinterpret_cast<long long*>(out_rows_dev), out_norm_dev, out_raw_dev, nullptr, n_out_dev, s);
}

// host-pointer forms: stage through the handle's workspace, one synchronisation at the end
static int host_call(rag_bm25_t* h, int64_t m, const int64_t* cand, int T, const int32_t* ids, const double* idf, double avgdl,
                     double k1, double b, bool rank, int use_min, double min_score, int limit, double* raw_out, double* norm_all,
                     int64_t* out_rows, double* out_norm, double* out_raw, int32_t* n_out) {
    if (int rc = bm25_check(h, m, T, ids, idf, avgdl)) return rc;
    if (rank) {
        if (int rc = rank_check(m, h->n_docs, cand == nullptr, limit, out_rows, out_norm, n_out)) return rc;
        *n_out = 0;
    } else if (m > 0 && !raw_out) {
        return fail(RAG_ERR_INVALID, "NULL buffer");
    }
    if (!cand && m > h->n_docs) return fail(RAG_ERR_INVALID, "m exceeds the number of documents");
    if (m == 0) {
        for (int r = 0; rank && r < limit; ++r) {
            out_rows[r] = -1;
            out_norm[r] = 0.0;
            if (out_raw) out_raw[r] = 0.0;
        }
        return RAG_OK;
    }
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceScope sc;
    if (int rc = sc.enter(h->device)) return rc;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t mb = al((size_t)m * 8), lb = al((size_t)(rank ? limit : 0) * 8);
    // layout: cand | raw | norm_all | term ids | idf | rows | norm | raw(top) | n_out
    const size_t total = 3 * mb + al((size_t)T * 4) + al((size_t)T * 8) + 3 * lb + 256;
    if (int rc = h->ws.ensure(total)) return rc;
    char* base = static_cast<char*>(h->ws.ptr);
    long long* dcand = reinterpret_cast<long long*>(base);
    double* draw = reinterpret_cast<double*>(base + mb);
    double* dnorm = reinterpret_cast<double*>(base + 2 * mb);
    int32_t* dids = reinterpret_cast<int32_t*>(base + 3 * mb);
    double* didf = reinterpret_cast<double*>(reinterpret_cast<char*>(dids) + al((size_t)T * 4));
    long long* drows = reinterpret_cast<long long*>(reinterpret_cast<char*>(didf) + al((size_t)T * 8));
    double* dtn = reinterpret_cast<double*>(reinterpret_cast<char*>(drows) + lb);
    double* dtr = reinterpret_cast<double*>(reinterpret_cast<char*>(dtn) + lb);
    int* dn = reinterpret_cast<int*>(reinterpret_cast<char*>(dtr) + lb);
    hipStream_t s = nullptr;
    if (cand) RAG_HIP_CHECK(hipMemcpyAsync(dcand, cand, (size_t)m * 8, hipMemcpyHostToDevice, s));
    RAG_HIP_CHECK(hipMemcpyAsync(dids, ids, (size_t)T * 4, hipMemcpyHostToDevice, s));
    RAG_HIP_CHECK(hipMemcpyAsync(didf, idf, (size_t)T * 8, hipMemcpyHostToDevice, s));
    if (int rc = enqueue_score(h, m, cand ? dcand : nullptr, T, dids, didf, avgdl, k1, b, draw, s)) return rc;
    if (rank) {
        if (int rc = enqueue_rank(m, cand ? dcand : nullptr, draw, use_min, min_score, limit, drows, dtn, dtr, norm_all ? dnorm : nullptr, dn, s))
            return rc;
        int32_t n = 0;
        RAG_HIP_CHECK(hipMemcpyAsync(&n, dn, sizeof(int), hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipMemcpyAsync(out_rows, drows, (size_t)limit * 8, hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipMemcpyAsync(out_norm, dtn, (size_t)limit * 8, hipMemcpyDeviceToHost, s));
        if (out_raw) RAG_HIP_CHECK(hipMemcpyAsync(out_raw, dtr, (size_t)limit * 8, hipMemcpyDeviceToHost, s));
        if (norm_all) RAG_HIP_CHECK(hipMemcpyAsync(norm_all, dnorm, (size_t)m * 8, hipMemcpyDeviceToHost, s));
        if (raw_out) RAG_HIP_CHECK(hipMemcpyAsync(raw_out, draw, (size_t)m * 8, hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipStreamSynchronize(s));
        if (n < 0 || n > limit) return fail(RAG_ERR_HIP, "ranking kernel returned a bad count");
        if (m == 1) {
            // a single score is normalised with the sigmoid (:195-197): redone with the host's libm so that the
            // value is the one Python's math.exp gives (the device exp may differ in the last bit)
            double sraw = 0.0;
            RAG_HIP_CHECK(hipMemcpy(&sraw, draw, sizeof(double), hipMemcpyDeviceToHost));
            const double ex = exp(-2.0 * sraw);
            const double nv = std::isinf(ex) ? (sraw < 0.0 ? 0.0 : 1.0) : 1.0 / (1.0 + ex);
            n = (!use_min || nv >= min_score) ? 1 : 0;
            out_rows[0] = n ? (cand ? cand[0] : 0) : -1;
            out_norm[0] = n ? nv : 0.0;
            if (out_raw) out_raw[0] = n ? sraw : 0.0;
            if (norm_all) norm_all[0] = nv;
        }
        *n_out = n;
    } else {
        RAG_HIP_CHECK(hipMemcpyAsync(raw_out, draw, (size_t)m * 8, hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipStreamSynchronize(s));
    }
    return RAG_OK;
}

int rag_bm25_score(rag_bm25_t* h, int64_t m, const int64_t* cand, int n_terms, const int32_t* term_ids,
                   const double* term_idf, double avgdl, double k1, double b, double* raw_out) {
    return host_call(h, m, cand, n_terms, term_ids, term_idf, avgdl, k1, b, false, 0, 0.0, 0, raw_out, nullptr, nullptr, nullptr,
                     nullptr, nullptr);
}

int rag_bm25_rank(rag_bm25_t* h, int64_t m, const int64_t* cand, int n_terms, const int32_t* term_ids,
                  const double* term_idf, double avgdl, double k1, double b, int use_min_score, double min_score, int limit,
                  int64_t* out_rows, double* out_norm, double* out_raw, int32_t* n_out) {
    return host_call(h, m, cand, n_terms, term_ids, term_idf, avgdl, k1, b, true, use_min_score, min_score, limit, nullptr, nullptr,
                     out_rows, out_norm, out_raw, n_out);
}

}  // extern "C"
