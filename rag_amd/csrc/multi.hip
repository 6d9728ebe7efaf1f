// One index over several GPUs of ONE process (rag_multi_*): the row-sharded layout of the
// multi-process path for callers that cannot join a process group -- the reference's server
// answers a search with a blocking call inside a single worker process
// (vector_db/search_optimizer.py:213-222).  Host code only: every shard is an ordinary
// rag_index_t on its device, driven by one persistent host thread per shard so that the
// devices scan at the same time.  The per-device answers (k per query) stay in HBM: every
// shard copies its lists peer to peer into a buffer on the first device, where one kernel
// merges them (score order, ties to the lower id) and only the merged nq x k answer comes
// back to the host.  Where a device cannot reach the first one peer to peer, the lists go
// through the host and are merged there.
#include <float.h>
#include <stdlib.h>
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

// device memory and the stream a shard's worker uses for a search (queries, mask, its k-per-query lists)
struct ShardStage {
    rag::Workspace ws;
    hipStream_t stream = nullptr;
};

struct rag_multi {
    int dim = 0, kind = 0, metric = 0;
    std::vector<ShardStage> stage;                  // per shard, on the shard's device
    rag::Workspace gather;                          // on device[0]: every shard's lists, then the merged answer
    hipStream_t merge_stream = nullptr;             // on device[0]
    const char* last_merge = "none";                // where the last search merged the devices' lists
    int peers = -1;                                 // -1 not probed yet, 1 every device reaches device[0], 0 host merge
    std::vector<rag_index_t*> shard;
    std::vector<int> device;
    std::vector<long long> rows;                    // rows per shard
    std::vector<std::unique_ptr<Worker>> worker;
    std::vector<Piece> pieces;                      // sorted by global
    long long n = 0;
    bool broken = false;                            // a failed add could not be rolled back on some device
    std::mutex mu;
};

namespace rag {
int index_truncate(rag_index* h, long long n);      // api.hip
}

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

// One workgroup per query: S lists of k (score, id), each in answer order (best first, ties to the lower id, -1 ids
// behind the hits).  An entry's place in the merged answer = the entries in front of it in its own list + those of every
// other list that beat it, found by binary search; entries are distinct in (score, id), so places are too.
__global__ void multi_merge_kernel(const float* D, const long long* I, int S, long long d_per, long long i_per, int k, int nearest,
                                   float* Dout, long long* Iout) {
    const long long q = blockIdx.x;
    float* dq = Dout + q * k;
    long long* iq = Iout + q * k;
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        dq[j] = nearest ? FLT_MAX : -FLT_MAX;
        iq[j] = -1;
    }
    __syncthreads();
    const int total = S * k;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int s = e / k, j = e - s * k;
        const long long id = I[(size_t)s * i_per + q * k + j];
        if (id < 0) continue;
        const float sc = D[(size_t)s * d_per + q * k + j];
        int place = j;
        for (int t = 0; t < S && place < k; ++t) {
            if (t == s) continue;
            const float* dt = D + (size_t)t * d_per + q * k;
            const long long* it = I + (size_t)t * i_per + q * k;
            int lo = 0, hi = k;                      // first entry of list t that does not beat (sc, id)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const long long mi = it[mid];
                const float ms = dt[mid];
                const bool beats = mi >= 0 && ((nearest ? ms < sc : ms > sc) || (ms == sc && mi < id));
                if (beats) lo = mid + 1; else hi = mid;
            }
            place += lo;
        }
        if (place < k) {
            dq[place] = sc;
            iq[place] = id;
        }
    }
}

// can every shard's device write into device[0]'s memory?  Probed once; RAG_MULTI_HOST_MERGE=1 keeps the host merge
int probe_peers(rag_multi* h) {
    if (h->peers >= 0) return h->peers;
    const char* e = getenv("RAG_MULTI_HOST_MERGE");
    int ok = !(e && *e == '1');
    const int d0 = h->device[0];
    for (size_t s = 0; ok && s < h->device.size(); ++s) {
        if (h->device[s] == d0) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, h->device[s], d0) != hipSuccess || !can) ok = 0;
    }
    h->peers = ok;
    return ok;
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

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
        (void)rag_index_set_stream_lifetime(idx, 1);       // a shard is searched on its stage stream, which lives as long as the shard
        h->shard.push_back(idx);
        h->device.push_back(device_ids[s]);
        h->rows.push_back(0);
    }
    h->stage.resize((size_t)n_devices);
    for (int s = 0; s < n_devices; ++s) h->worker.emplace_back(new Worker());
    *out = h.release();
    return RAG_OK;
}

int rag_multi_free(rag_multi_t* h) {
    if (!h) return RAG_OK;
    h->worker.clear();                              // joins the threads
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t s = 0; s < h->stage.size(); ++s) {
        if (!h->stage[s].stream && !h->stage[s].ws.ptr) continue;
        (void)hipSetDevice(h->device[s]);
        if (h->stage[s].stream) (void)hipStreamDestroy(h->stage[s].stream);
        h->stage[s].ws.release();
    }
    if (h->gather.ptr || h->merge_stream) {
        (void)hipSetDevice(h->device[0]);
        if (h->merge_stream) (void)hipStreamDestroy(h->merge_stream);
        h->gather.release();
    }
    (void)hipSetDevice(prev);
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

const char* rag_multi_last_merge(const rag_multi_t* h) { return h ? h->last_merge : "none"; }

int rag_multi_train_ivf(rag_multi_t* h, int nlist, const float* centroids, int64_t nsample, const float* sample) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->broken) return fail(RAG_ERR_INVALID, "a failed add left this index inconsistent: free it");
    // the same centroids on every device: the shards together are one IVF index over all rows
    return on_all(h, [&](int s) { return rag_index_train_ivf(h->shard[(size_t)s], nlist, centroids, nsample, sample); });
}

int rag_multi_set_prefilter(rag_multi_t* h, int mode) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->broken) return fail(RAG_ERR_INVALID, "a failed add left this index inconsistent: free it");
    return on_all(h, [&](int s) { return rag_index_set_prefilter(h->shard[(size_t)s], mode); });
}

int rag_multi_add(rag_multi_t* h, int64_t n, const float* x, const int64_t* ids) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (n < 0) return fail(RAG_ERR_INVALID, "n < 0");
    if (n == 0) return RAG_OK;
    if (!x) return fail(RAG_ERR_INVALID, "x is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->broken) return fail(RAG_ERR_INVALID, "a failed add left this index inconsistent: free it");
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
        const int rc_add = rag_index_add(h->shard[(size_t)s], e - b, x + (size_t)b * h->dim, ids + b);
        // test hook: RAG_TEST_MULTI_FAIL_ADD=<shard> makes that shard report a failure (after it took its rows)
        const char* inject = getenv("RAG_TEST_MULTI_FAIL_ADD");
        if (rc_add == RAG_OK && inject && *inject && atoi(inject) == s) return fail(RAG_ERR_OOM, "injected failure");
        return rc_add;
    });
    if (rc != RAG_OK) {
        // all or nothing: the shards that took their part give it back (their row counts are the ones recorded
        // here), so the next add starts from the positions `pieces` knows.  A shard that cannot -- it should
        // always be able to: nothing has searched the new rows -- leaves the whole index unusable.
        const std::string why = rag_last_error();
        for (int s = 0; s < S; ++s)
            if (rag_index_ntotal(h->shard[(size_t)s]) != h->rows[(size_t)s] &&
                rag::index_truncate(h->shard[(size_t)s], h->rows[(size_t)s]) != RAG_OK)
                h->broken = true;
        return fail(rc, why + (h->broken ? " (and a device could not take its rows back: the index refuses further use)" : ""));
    }
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
    if (h->broken) return fail(RAG_ERR_INVALID, "a failed add left this index inconsistent: free it");
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
    if (h->broken) return fail(RAG_ERR_INVALID, "a failed add left this index inconsistent: free it");
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

}  // extern "C"

// a caller filter cut into the shards' own row orders, each part kept on its shard's device
struct rag_multi_filter {
    long long n = 0;
    std::vector<rag_filter_t*> part;
};

static int multi_search(rag_multi_t* h, int64_t nq, const float* q, int k, float min_score, int nprobe,
                        const uint8_t* row_mask, const rag_multi_filter* filter, float* D, int64_t* I) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (nq < 0 || k < 1) return fail(RAG_ERR_INVALID, "nq < 0 or k < 1");
    if (nq == 0) return RAG_OK;
    if (!q || !D || !I) return fail(RAG_ERR_INVALID, "NULL buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->broken) return fail(RAG_ERR_INVALID, "a failed add left this index inconsistent: free it");
    const size_t S = h->shard.size();
    if (filter) {
        if (filter->n != h->n || filter->part.size() != S)
            return fail(RAG_ERR_INVALID, "the filter was made for " + std::to_string(filter->n) + " rows, the index holds " + std::to_string(h->n));
        for (size_t s = 0; s < S; ++s)
            if (rag_filter_rows(filter->part[s]) != h->rows[s]) return fail(RAG_ERR_INVALID, "the filter does not fit the shards of this index");
    }
    // the caller's mask counts rows in insertion order: cut it into the shards' own orders
    std::vector<std::vector<uint8_t>> mask(S);
    if (row_mask) {
        for (size_t s = 0; s < S; ++s) mask[s].resize((size_t)h->rows[s]);
        for (const Piece& p : h->pieces) memcpy(mask[(size_t)p.shard].data() + p.local, row_mask + p.global, (size_t)p.count);
    }
    const size_t per = (size_t)nq * (size_t)k;
    const bool nearest = h->metric == RAG_METRIC_L2;
    if (probe_peers(h)) {
        // ---- the lists stay on the devices: peer copies into device[0], one merge kernel, one copy out ----
        const int d0 = h->device[0];
        const size_t db = al256(per * sizeof(float)), ib = al256(per * sizeof(int64_t));
        int prev = 0;
        RAG_HIP_CHECK(hipGetDevice(&prev));
        struct Restore { int p; ~Restore() { (void)hipSetDevice(p); } } restore{prev};
        RAG_HIP_CHECK(hipSetDevice(d0));
        if (int rc = h->gather.ensure(S * (db + ib) + db + ib)) return rc;
        if (!h->merge_stream) RAG_HIP_CHECK(hipStreamCreateWithFlags(&h->merge_stream, hipStreamNonBlocking));
        char* g = static_cast<char*>(h->gather.ptr);
        float* gD = reinterpret_cast<float*>(g);                                   // [S][db]
        int64_t* gI = reinterpret_cast<int64_t*>(g + S * db);                      // [S][ib]
        float* oD = reinterpret_cast<float*>(g + S * (db + ib));
        int64_t* oI = reinterpret_cast<int64_t*>(g + S * (db + ib) + db);
        const int rc = on_all(h, [&](int s) {
            const int dev = h->device[(size_t)s];
            RAG_HIP_CHECK(hipSetDevice(dev));
            ShardStage& st = h->stage[(size_t)s];
            if (!st.stream) RAG_HIP_CHECK(hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking));
            const size_t qb = al256((size_t)nq * h->dim * sizeof(float));
            const size_t mb = row_mask ? al256((size_t)h->rows[(size_t)s]) : 0;
            if (int r = st.ws.ensure(qb + db + ib + mb + 256)) return r;
            char* w = static_cast<char*>(st.ws.ptr);
            float* q_dev = reinterpret_cast<float*>(w);
            float* D_dev = reinterpret_cast<float*>(w + qb);
            int64_t* I_dev = reinterpret_cast<int64_t*>(w + qb + db);
            const uint8_t* m_dev = row_mask ? reinterpret_cast<uint8_t*>(w + qb + db + ib)
                                            : filter ? rag_filter_device_bytes(filter->part[(size_t)s]) : nullptr;
            RAG_HIP_CHECK(hipMemcpyAsync(q_dev, q, (size_t)nq * h->dim * sizeof(float), hipMemcpyHostToDevice, st.stream));
            if (row_mask && h->rows[(size_t)s] > 0)
                RAG_HIP_CHECK(hipMemcpyAsync(w + qb + db + ib, mask[(size_t)s].data(), (size_t)h->rows[(size_t)s], hipMemcpyHostToDevice, st.stream));
            // (a shard without rows answers with empty lists by itself)
            if (int r = rag_index_search_device(h->shard[(size_t)s], nq, q_dev, k, min_score, nprobe, m_dev, D_dev, I_dev, st.stream)) return r;
            float* dst_d = reinterpret_cast<float*>(reinterpret_cast<char*>(gD) + (size_t)s * db);
            int64_t* dst_i = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(gI) + (size_t)s * ib);
            if (dev == d0) {
                RAG_HIP_CHECK(hipMemcpyAsync(dst_d, D_dev, per * sizeof(float), hipMemcpyDeviceToDevice, st.stream));
                RAG_HIP_CHECK(hipMemcpyAsync(dst_i, I_dev, per * sizeof(int64_t), hipMemcpyDeviceToDevice, st.stream));
            } else {
                RAG_HIP_CHECK(hipMemcpyPeerAsync(dst_d, d0, D_dev, dev, per * sizeof(float), st.stream));
                RAG_HIP_CHECK(hipMemcpyPeerAsync(dst_i, d0, I_dev, dev, per * sizeof(int64_t), st.stream));
            }
            RAG_HIP_CHECK(hipStreamSynchronize(st.stream));
            return (int)RAG_OK;
        });
        if (rc != RAG_OK) return rc;
        hipLaunchKernelGGL(multi_merge_kernel, dim3((unsigned)nq), dim3(256), 0, h->merge_stream, gD,
                           reinterpret_cast<const long long*>(gI), (int)S, (long long)(db / sizeof(float)),
                           (long long)(ib / sizeof(int64_t)), k, nearest ? 1 : 0, oD,
                           reinterpret_cast<long long*>(oI));
        RAG_HIP_CHECK(hipGetLastError());
        RAG_HIP_CHECK(hipMemcpyAsync(D, oD, per * sizeof(float), hipMemcpyDeviceToHost, h->merge_stream));
        RAG_HIP_CHECK(hipMemcpyAsync(I, oI, per * sizeof(int64_t), hipMemcpyDeviceToHost, h->merge_stream));
        RAG_HIP_CHECK(hipStreamSynchronize(h->merge_stream));
        h->last_merge = "device";
        return RAG_OK;
    }
    // ---- a device cannot reach device[0] peer to peer: lists through the host, merged here ----
    h->last_merge = "host";
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
        if (filter) return rag_index_search_filtered(h->shard[(size_t)s], nq, q, k, min_score, nprobe, filter->part[(size_t)s], d, i);
        return rag_index_search(h->shard[(size_t)s], nq, q, k, min_score, nprobe, row_mask ? mask[(size_t)s].data() : nullptr, d, i);
    });
    if (rc != RAG_OK) return rc;
    // k-way merge of S sorted lists per query: score descending (L2: distance ascending), ties to the lower id
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

extern "C" {

int rag_multi_search(rag_multi_t* h, int64_t nq, const float* q, int k, float min_score, int nprobe,
                     const uint8_t* row_mask, float* D, int64_t* I) {
    return multi_search(h, nq, q, k, min_score, nprobe, row_mask, nullptr, D, I);
}

int rag_multi_search_filtered(rag_multi_t* h, int64_t nq, const float* q, int k, float min_score, int nprobe,
                              const rag_multi_filter_t* filter, float* D, int64_t* I) {
    return multi_search(h, nq, q, k, min_score, nprobe, nullptr, filter, D, I);
}

int rag_multi_filter_free(rag_multi_filter_t* f) {
    if (!f) return RAG_OK;
    int rc = RAG_OK;
    for (rag_filter_t* p : f->part) {
        const int r = rag_filter_free(p);
        if (r != RAG_OK) rc = r;
    }
    delete f;
    return rc;
}

int rag_multi_filter_create(rag_multi_t* h, const uint8_t* row_mask, rag_multi_filter_t** out) {
    if (!out) return fail(RAG_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->n > 0 && !row_mask) return fail(RAG_ERR_INVALID, "filter: NULL bytes");
    const size_t S = h->shard.size();
    std::unique_ptr<rag_multi_filter> f(new (std::nothrow) rag_multi_filter);
    if (!f) return fail(RAG_ERR_OOM, "out of host memory");
    f->n = h->n;
    // the caller's bytes count rows in insertion order: cut them into the shards' own orders
    std::vector<std::vector<uint8_t>> mask(S);
    for (size_t s = 0; s < S; ++s) mask[s].resize((size_t)h->rows[s]);
    for (const Piece& p : h->pieces) memcpy(mask[(size_t)p.shard].data() + p.local, row_mask + p.global, (size_t)p.count);
    for (size_t s = 0; s < S; ++s) {
        rag_filter_t* part = nullptr;
        const int rc = rag_filter_create(h->device[s], h->rows[s], mask[s].data(), &part);
        if (rc != RAG_OK) {
            const std::string msg = rag_last_error();
            (void)rag_multi_filter_free(f.release());
            return fail(rc, msg);
        }
        f->part.push_back(part);
    }
    *out = f.release();
    return RAG_OK;
}

}  // extern "C"
