// One index over several GPUs of ONE process (rag_multi_*): the row-sharded layout of the
// multi-process path for callers that cannot join a process group -- the reference's server
// answers a search with a blocking call inside a single worker process
// (vector_db/search_optimizer.py:213-222).  Host code only: every shard is an ordinary
// rag_index_t on its device, driven by one persistent host thread per shard so that the
// devices scan at the same time; the per-device answers (k per query) are merged on the host.
#include <float.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "index.h"

namespace {

// a host thread that runs one job at a time for its shard
class Worker {
  public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void start(std::function<int()> job) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = std::move(job);
            busy_ = true;
            rc_ = RAG_OK;
            err_.clear();
        }
        cv_.notify_all();
    }
    // status of the job; its message (if any) goes to *err
    int wait(std::string* err) {
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return !busy_; });
        if (rc_ != RAG_OK && err) *err = err_;
        return rc_;
    }

  private:
    void loop() {
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || (busy_ && job_); });
                if (stop_) {
                    (void)rag_thread_release();        // this thread's staging memory on its device
                    return;
                }
                job = std::move(job_);
                job_ = nullptr;
            }
            const int rc = job();
            const std::string msg = rc != RAG_OK ? std::string(rag_last_error()) : std::string();   // the message is per thread
            {
                std::lock_guard<std::mutex> lk(mu_);
                rc_ = rc;
                err_ = msg;
                busy_ = false;
            }
            done_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::function<int()> job_;
    bool busy_ = false, stop_ = false;
    int rc_ = RAG_OK;
    std::string err_;
    std::thread th_;      // last member: starts when everything above exists
};

// rows [global, global + count) live on shard `shard` at [local, local + count)
struct Piece {
    long long global, count, local;
    int shard;
};

}  // namespace

struct rag_multi {
    int dim = 0, kind = 0, metric = 0;
    std::vector<rag_index_t*> shard;
    std::vector<int> device;
    std::vector<long long> rows;                    // rows per shard
    std::vector<std::unique_ptr<Worker>> worker;
    std::vector<Piece> pieces;                      // sorted by global
    long long n = 0;
    std::mutex mu;
};

using rag::fail;

namespace {

// run job(s) for every shard on its worker; first failure wins
int on_all(rag_multi* h, const std::function<int(int)>& job) {
    const int S = (int)h->shard.size();
    for (int s = 0; s < S; ++s) h->worker[(size_t)s]->start([&job, s] { return job(s); });
    int rc = RAG_OK;
    std::string msg;
    for (int s = 0; s < S; ++s) {
        std::string m;
        const int r = h->worker[(size_t)s]->wait(&m);
        if (r != RAG_OK && rc == RAG_OK) { rc = r; msg = "device " + std::to_string(h->device[(size_t)s]) + ": " + m; }
    }
    return rc == RAG_OK ? RAG_OK : fail(rc, msg);
}

const Piece* piece_of(const rag_multi* h, long long row) {
    auto it = std::upper_bound(h->pieces.begin(), h->pieces.end(), row, [](long long r, const Piece& p) { return r < p.global; });
    if (it == h->pieces.begin()) return nullptr;
    --it;
    return row < it->global + it->count ? &*it : nullptr;
}

}  // namespace

extern "C" {

int rag_multi_create(int dim, int metric, int kind, int n_devices, const int* device_ids, rag_multi_t** out) {
    if (!out) return fail(RAG_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return fail(RAG_ERR_INVALID, "n_devices must be in [1, 64]");
    if (!device_ids) return fail(RAG_ERR_INVALID, "device_ids is NULL");
    std::unique_ptr<rag_multi> h(new rag_multi());
    h->dim = dim;
    h->kind = kind;
    h->metric = metric;
    for (int s = 0; s < n_devices; ++s) {
        rag_index_t* idx = nullptr;
        if (int rc = rag_index_create(dim, metric, kind, device_ids[s], &idx)) {
            for (rag_index_t* made : h->shard) rag_index_free(made);
            return rc;
        }
        h->shard.push_back(idx);
        h->device.push_back(device_ids[s]);
        h->rows.push_back(0);
    }
    for (int s = 0; s < n_devices; ++s) h->worker.emplace_back(new Worker());
    *out = h.release();
    return RAG_OK;
}

int rag_multi_free(rag_multi_t* h) {
    if (!h) return RAG_OK;
    h->worker.clear();                              // joins the threads
    int rc = RAG_OK;
    for (rag_index_t* idx : h->shard)
        if (int r = rag_index_free(idx)) rc = r;
    delete h;
    return rc;
}

int64_t rag_multi_ntotal(const rag_multi_t* h) { return h ? h->n : 0; }
int rag_multi_ndevices(const rag_multi_t* h) { return h ? (int)h->shard.size() : 0; }
int64_t rag_multi_shard_rows(const rag_multi_t* h, int s) {
    return (h && s >= 0 && s < (int)h->shard.size()) ? h->rows[(size_t)s] : 0;
}

int rag_multi_train_ivf(rag_multi_t* h, int nlist, const float* centroids, int64_t nsample, const float* sample) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    // the same centroids on every device: the shards together are one IVF index over all rows
    return on_all(h, [&](int s) { return rag_index_train_ivf(h->shard[(size_t)s], nlist, centroids, nsample, sample); });
}

int rag_multi_set_prefilter(rag_multi_t* h, int mode) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    return on_all(h, [&](int s) { return rag_index_set_prefilter(h->shard[(size_t)s], mode); });
}

int rag_multi_add(rag_multi_t* h, int64_t n, const float* x, const int64_t* ids) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (n < 0) return fail(RAG_ERR_INVALID, "n < 0");
    if (n == 0) return RAG_OK;
    if (!x) return fail(RAG_ERR_INVALID, "x is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    const int S = (int)h->shard.size();
    // part s of the batch = rows [n*s/S, n*(s+1)/S); every shard keeps explicit ids (the global
    // position when the caller gives none) because its rows are not contiguous in the whole
    std::vector<int64_t> own;
    if (!ids) {
        own.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) own[(size_t)i] = h->n + i;
        ids = own.data();
    }
    const int rc = on_all(h, [&](int s) {
        const int64_t b = n * s / S, e = n * (s + 1) / S;
        if (e == b) return (int)RAG_OK;
        return rag_index_add(h->shard[(size_t)s], e - b, x + (size_t)b * h->dim, ids + b);
    });
    if (rc != RAG_OK) return rc;      // shards that took their part keep it; ntotal of the whole is unchanged
    for (int s = 0; s < S; ++s) {
        const int64_t b = n * s / S, e = n * (s + 1) / S;
        if (e == b) continue;
        h->pieces.push_back(Piece{h->n + b, e - b, h->rows[(size_t)s], s});
        h->rows[(size_t)s] += e - b;
    }
    h->n += n;
    return RAG_OK;
}

int rag_multi_set_active(rag_multi_t* h, int64_t n, const int64_t* rows, int active) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (n < 0 || (n > 0 && !rows)) return fail(RAG_ERR_INVALID, "bad row list");
    std::lock_guard<std::mutex> lk(h->mu);
    std::vector<std::vector<int64_t>> local(h->shard.size());
    for (int64_t i = 0; i < n; ++i) {
        const Piece* p = piece_of(h, rows[i]);
        if (!p) return fail(RAG_ERR_INVALID, "row out of range");
        local[(size_t)p->shard].push_back(p->local + (rows[i] - p->global));
    }
    return on_all(h, [&](int s) {
        const auto& l = local[(size_t)s];
        return l.empty() ? (int)RAG_OK : rag_index_set_active(h->shard[(size_t)s], (int64_t)l.size(), l.data(), active);
    });
}

int rag_multi_reconstruct(rag_multi_t* h, int64_t n, const int64_t* rows, float* out) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (n < 0 || (n > 0 && (!rows || !out))) return fail(RAG_ERR_INVALID, "bad row list");
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t S = h->shard.size();
    std::vector<std::vector<int64_t>> local(S), where(S);
    for (int64_t i = 0; i < n; ++i) {
        const Piece* p = piece_of(h, rows[i]);
        if (!p) return fail(RAG_ERR_INVALID, "row out of range");
        local[(size_t)p->shard].push_back(p->local + (rows[i] - p->global));
        where[(size_t)p->shard].push_back(i);
    }
    std::vector<std::vector<float>> got(S);
    const int rc = on_all(h, [&](int s) {
        const auto& l = local[(size_t)s];
        if (l.empty()) return (int)RAG_OK;
        got[(size_t)s].resize(l.size() * (size_t)h->dim);
        return rag_index_reconstruct(h->shard[(size_t)s], (int64_t)l.size(), l.data(), got[(size_t)s].data());
    });
    if (rc != RAG_OK) return rc;
    for (size_t s = 0; s < S; ++s)
        for (size_t j = 0; j < where[s].size(); ++j)
            memcpy(out + (size_t)where[s][j] * h->dim, got[s].data() + j * (size_t)h->dim, (size_t)h->dim * sizeof(float));
    return RAG_OK;
}

int rag_multi_search(rag_multi_t* h, int64_t nq, const float* q, int k, float min_score, int nprobe,
                     const uint8_t* row_mask, float* D, int64_t* I) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (nq < 0 || k < 1) return fail(RAG_ERR_INVALID, "nq < 0 or k < 1");
    if (nq == 0) return RAG_OK;
    if (!q || !D || !I) return fail(RAG_ERR_INVALID, "NULL buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t S = h->shard.size();
    // the caller's mask counts rows in insertion order: cut it into the shards' own orders
    std::vector<std::vector<uint8_t>> mask(S);
    if (row_mask) {
        for (size_t s = 0; s < S; ++s) mask[s].resize((size_t)h->rows[s]);
        for (const Piece& p : h->pieces) memcpy(mask[(size_t)p.shard].data() + p.local, row_mask + p.global, (size_t)p.count);
    }
    const size_t per = (size_t)nq * (size_t)k;
    std::vector<float> Ds(S * per);
    std::vector<int64_t> Is(S * per);
    const int rc = on_all(h, [&](int s) {
        float* d = Ds.data() + (size_t)s * per;
        int64_t* i = Is.data() + (size_t)s * per;
        if (h->rows[(size_t)s] == 0) {
            std::fill(d, d + per, h->metric == RAG_METRIC_L2 ? FLT_MAX : -FLT_MAX);
            std::fill(i, i + per, (int64_t)-1);
            return (int)RAG_OK;
        }
        return rag_index_search(h->shard[(size_t)s], nq, q, k, min_score, nprobe, row_mask ? mask[(size_t)s].data() : nullptr, d, i);
    });
    if (rc != RAG_OK) return rc;
    // k-way merge of S sorted lists per query: score descending (L2: distance ascending), ties to the lower id
    const bool nearest = h->metric == RAG_METRIC_L2;
    std::vector<size_t> head(S);
    for (int64_t qi = 0; qi < nq; ++qi) {
        std::fill(head.begin(), head.end(), (size_t)0);
        for (int j = 0; j < k; ++j) {
            int best = -1;
            float bs = 0.f;
            int64_t bi = 0;
            for (size_t s = 0; s < S; ++s) {
                if (head[s] >= (size_t)k) continue;
                const size_t at = s * per + (size_t)qi * k + head[s];
                const int64_t id = Is[at];
                if (id < 0) { head[s] = (size_t)k; continue; }        // the rest of this list is padding
                const float sc = Ds[at];
                if (best < 0 || (nearest ? sc < bs : sc > bs) || (sc == bs && id < bi)) { best = (int)s; bs = sc; bi = id; }
            }
            float* d = D + (size_t)qi * k + j;
            int64_t* i = I + (size_t)qi * k + j;
            if (best < 0) { *d = nearest ? FLT_MAX : -FLT_MAX; *i = -1; continue; }
            *d = bs;
            *i = bi;
            ++head[(size_t)best];
        }
    }
    return RAG_OK;
}

}  // extern "C"
