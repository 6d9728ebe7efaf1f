// Binary snapshot of an index: what the reference's JSON backup of a Qdrant
// collection (vector_db/qdrant_client.py:475-621, every vector as a list of
// Python floats) becomes when the vectors live in HBM.
//
//   header     magic "RAGHIP01", dim, metric, n, id_base, flags (bit0 ids, bit1 active,
//              bit2 IVF), nlist (IVF, else 0)
//   centroids  nlist x dim fp32      (IVF)
//   lists      n x int32             (IVF: the list of every row, insertion order)
//   rows       n x dim fp32, row-major, insertion order, as stored (COSINE rows are normalised)
//   ids        n x int64             (flag bit 0)
//   active     n x uint8             (flag bit 1)
// An IVF index comes back with the lists its rows had: nothing is re-assigned on load, so a
// row at a coarse tie stays where it was.  Payloads stay with the caller (a JSON sidecar
// written by the Python layer).
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <vector>

#include "index.h"

using namespace rag;

namespace {
struct SnapHeader {
    char magic[8];
    int32_t dim;
    int32_t metric;
    int64_t n;
    int64_t id_base;
    int32_t flags;
    int32_t nlist;
};
const char kMagic[8] = {'R', 'A', 'G', 'H', 'I', 'P', '0', '1'};
struct FileCloser {
    FILE* f;
    ~FileCloser() { if (f) fclose(f); }
};
}  // namespace

extern "C" int rag_index_save(rag_index_t* h, const char* path) {
    if (!h || !path) return fail(RAG_ERR_INVALID, "NULL argument");
    RAG_LOCK_INDEX(h);
    const bool ivf = h->kind == RAG_KIND_IVF;
    if (ivf && !h->ivf.quantizer) return fail(RAG_ERR_INVALID, "IVF index is not trained");
    FILE* f = fopen(path, "wb");
    if (!f) return fail(RAG_ERR_INVALID, std::string("cannot open for writing: ") + path);
    FileCloser fc{f};
    int prev = 0;
    RAG_HIP_CHECK(hipGetDevice(&prev));
    RAG_HIP_CHECK(hipSetDevice(h->device));
    struct Restore { int p; ~Restore() { (void)hipSetDevice(p); } } restore{prev};
    SnapHeader hd;
    memset(&hd, 0, sizeof(hd));
    memcpy(hd.magic, kMagic, 8);
    hd.dim = h->dim;
    hd.metric = h->metric;
    hd.n = h->n;
    hd.id_base = h->id_base;
    hd.flags = (h->ids_dev ? 1 : 0) | (h->active ? 2 : 0) | (ivf ? 4 : 0);
    hd.nlist = ivf ? h->ivf.nlist : 0;
    if (fwrite(&hd, sizeof(hd), 1, f) != 1) return fail(RAG_ERR_INVALID, "write failed");
    const long long chunk = std::max<long long>(1, (64LL << 20) / ((long long)h->ld * 4));
    if (ivf) {
        // centroids (rows of the quantizer, a FLAT index) and every row's list
        rag_index* qz = h->ivf.quantizer;
        std::vector<float> cen((size_t)std::min<long long>(chunk, hd.nlist) * h->dim);
        std::vector<int64_t> which;
        for (long long r0 = 0; r0 < hd.nlist; r0 += chunk) {
            const long long c = std::min<long long>(chunk, hd.nlist - r0);
            which.resize((size_t)c);
            for (long long i = 0; i < c; ++i) which[(size_t)i] = r0 + i;
            if (int rc = rag_index_reconstruct(qz, c, which.data(), cen.data())) return rc;
            if (fwrite(cen.data(), sizeof(float), (size_t)c * h->dim, f) != (size_t)c * h->dim) return fail(RAG_ERR_INVALID, "write failed");
        }
        if ((long long)h->ivf.assign.size() < h->n) return fail(RAG_ERR_HIP, "IVF bookkeeping is shorter than the index");
        if (h->n > 0 && fwrite(h->ivf.assign.data(), sizeof(int), (size_t)h->n, f) != (size_t)h->n) return fail(RAG_ERR_INVALID, "write failed");
        // rows in insertion order (storage is grouped by list)
        std::vector<float> host((size_t)std::min<long long>(chunk, std::max<long long>(h->n, 1)) * h->dim);
        for (long long r0 = 0; r0 < h->n; r0 += chunk) {
            const long long c = std::min(chunk, h->n - r0);
            which.resize((size_t)c);
            for (long long i = 0; i < c; ++i) which[(size_t)i] = r0 + i;
            if (int rc = rag_index_reconstruct(h, c, which.data(), host.data())) return rc;
            if (fwrite(host.data(), sizeof(float), (size_t)c * h->dim, f) != (size_t)c * h->dim) return fail(RAG_ERR_INVALID, "write failed");
        }
    } else {
        // rows, 64 MiB at a time, padding stripped: two page-locked buffers, so that one chunk travels from the device while the
        // chunk before it is written to the file
        const size_t chunk_bytes = (size_t)std::min<long long>(chunk, std::max<long long>(h->n, 1)) * h->ld * sizeof(float);
        struct Pinned {
            float* p[2] = {nullptr, nullptr};
            hipStream_t s = nullptr;
            ~Pinned() {
                if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
                for (float* q : p) if (q) (void)hipHostFree(q);
            }
        } pin;
        for (int b = 0; b < 2; ++b) RAG_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&pin.p[b]), chunk_bytes, hipHostMallocDefault));
        RAG_HIP_CHECK(hipStreamCreateWithFlags(&pin.s, hipStreamNonBlocking));
        RAG_HIP_CHECK(hipDeviceSynchronize());          // the rows as every stream has left them
        auto fetch = [&](long long r0, int b) -> hipError_t {
            const long long c = std::min(chunk, h->n - r0);
            return hipMemcpyAsync(pin.p[b], h->X + (size_t)r0 * h->ld, (size_t)c * h->ld * sizeof(float), hipMemcpyDeviceToHost, pin.s);
        };
        if (h->n > 0) RAG_HIP_CHECK(fetch(0, 0));
        int b = 0;
        for (long long r0 = 0; r0 < h->n; r0 += chunk, b ^= 1) {
            const long long c = std::min(chunk, h->n - r0);
            RAG_HIP_CHECK(hipStreamSynchronize(pin.s));                    // chunk r0 is in pin.p[b]
            if (r0 + chunk < h->n) RAG_HIP_CHECK(fetch(r0 + chunk, b ^ 1));
            const float* host = pin.p[b];
            if (h->ld == h->dim) {
                if (fwrite(host, sizeof(float), (size_t)c * h->dim, f) != (size_t)c * h->dim) return fail(RAG_ERR_INVALID, "write failed");
            } else {
                for (long long r = 0; r < c; ++r)
                    if (fwrite(host + (size_t)r * h->ld, sizeof(float), (size_t)h->dim, f) != (size_t)h->dim) return fail(RAG_ERR_INVALID, "write failed");
            }
        }
    }
    if (h->ids_dev && h->n > 0) {
        std::vector<long long> ids((size_t)h->n);
        RAG_HIP_CHECK(hipMemcpy(ids.data(), h->ids_dev, (size_t)h->n * sizeof(long long), hipMemcpyDeviceToHost));
        if (fwrite(ids.data(), sizeof(long long), (size_t)h->n, f) != (size_t)h->n) return fail(RAG_ERR_INVALID, "write failed");
    }
    if (h->active && h->n > 0) {
        std::vector<uint8_t> act((size_t)h->n);
        RAG_HIP_CHECK(hipMemcpy(act.data(), h->active, (size_t)h->n, hipMemcpyDeviceToHost));
        if (fwrite(act.data(), 1, (size_t)h->n, f) != (size_t)h->n) return fail(RAG_ERR_INVALID, "write failed");
    }
    return RAG_OK;
}

static int load_body(FILE* f, const SnapHeader& hd, bool ivf, int device, rag_index_t** out);

extern "C" int rag_index_load(const char* path, int device, rag_index_t** out) {
    if (!path || !out) return fail(RAG_ERR_INVALID, "NULL argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(RAG_ERR_INVALID, std::string("cannot open: ") + path);
    FileCloser fc{f};
    SnapHeader hd;
    if (fread(&hd, sizeof(hd), 1, f) != 1 || memcmp(hd.magic, kMagic, 8) != 0) return fail(RAG_ERR_INVALID, "not a rag_hip snapshot");
    const bool ivf = (hd.flags & 4) != 0;
    if (hd.n < 0 || hd.dim < 1 || (hd.flags & ~7) != 0 || (ivf ? hd.nlist < 1 : hd.nlist != 0))
        return fail(RAG_ERR_INVALID, "corrupt snapshot header");
    if (hd.metric != RAG_METRIC_IP && hd.metric != RAG_METRIC_COSINE && hd.metric != RAG_METRIC_L2) return fail(RAG_ERR_INVALID, "snapshot names an unknown metric");
    if (hd.metric == RAG_METRIC_L2 && ivf) return fail(RAG_ERR_INVALID, "corrupt snapshot header");
    // the header's counts size every buffer below: hold them to the limits of an index and to the length of the file first
    if (hd.dim > 65536 || hd.n > 0x7FFFFFF0LL || hd.nlist > (1 << 22)) return fail(RAG_ERR_INVALID, "corrupt snapshot header (counts out of range)");
    {
        struct stat st;
        if (fstat(fileno(f), &st) != 0) return fail(RAG_ERR_INVALID, "cannot stat the snapshot");
        unsigned long long need = sizeof(hd) + (unsigned long long)hd.n * hd.dim * sizeof(float);
        if (ivf) need += (unsigned long long)hd.nlist * hd.dim * sizeof(float) + (unsigned long long)hd.n * sizeof(int);
        if (hd.flags & 1) need += (unsigned long long)hd.n * sizeof(long long);
        if (hd.flags & 2) need += (unsigned long long)hd.n;
        if ((unsigned long long)st.st_size < need)
            return fail(RAG_ERR_INVALID, "truncated snapshot: " + std::to_string((long long)st.st_size) + " bytes, the header needs " + std::to_string(need));
    }
    try {
        return load_body(f, hd, ivf, device, out);
    } catch (const std::bad_alloc&) {
        return fail(RAG_ERR_OOM, "out of host memory while loading the snapshot");
    } catch (const std::exception& e) {
        return fail(RAG_ERR_INVALID, std::string("snapshot load failed: ") + e.what());
    }
}

static int load_body(FILE* f, const SnapHeader& hd, bool ivf, int device, rag_index_t** out) {
    rag_index* h = nullptr;
    // rows in the file are already normalised for COSINE: load them as they are, then restore the metric
    // (L2 rows are stored as given; adding them rebuilds their extra column)
    if (int rc = rag_index_create(hd.dim, hd.metric == RAG_METRIC_L2 ? RAG_METRIC_L2 : RAG_METRIC_IP, ivf ? RAG_KIND_IVF : RAG_KIND_FLAT, device, &h)) return rc;
    struct Guard { rag_index* h; bool keep; ~Guard() { if (!keep) rag_index_free(h); } } guard{h, false};
    if (int rc = rag_index_set_id_base(h, hd.id_base)) return rc;
    const long long chunk = std::max<long long>(1, (64LL << 20) / ((long long)hd.dim * 4));
    std::vector<int> lists;
    if (ivf) {
        std::vector<float> cen((size_t)hd.nlist * hd.dim);
        if (fread(cen.data(), sizeof(float), cen.size(), f) != cen.size()) return fail(RAG_ERR_INVALID, "truncated snapshot (centroids)");
        if (int rc = rag_index_train_ivf(h, hd.nlist, cen.data(), 0, nullptr)) return rc;
        lists.resize((size_t)hd.n);
        if (hd.n > 0 && fread(lists.data(), sizeof(int), (size_t)hd.n, f) != (size_t)hd.n) return fail(RAG_ERR_INVALID, "truncated snapshot (lists)");
        h->ivf.preassigned = lists.data();
    }
    struct Unset { rag_index* h; ~Unset() { h->ivf.preassigned = nullptr; } } unset{h};
    if (int rc = rag_index_reserve(h, hd.n)) return rc;
    std::vector<float> host((size_t)std::min<long long>(chunk, std::max<long long>(hd.n, 1)) * hd.dim);
    for (long long r0 = 0; r0 < hd.n; r0 += chunk) {
        const long long c = std::min(chunk, hd.n - r0);
        if (fread(host.data(), sizeof(float), (size_t)c * hd.dim, f) != (size_t)c * hd.dim) return fail(RAG_ERR_INVALID, "truncated snapshot (rows)");
        if (int rc = rag_index_add(h, c, host.data(), nullptr)) return rc;
    }
    h->ivf.preassigned = nullptr;
    h->metric = hd.metric;
    int prev = 0;
    RAG_HIP_CHECK(hipGetDevice(&prev));
    RAG_HIP_CHECK(hipSetDevice(device));
    struct Restore { int p; ~Restore() { (void)hipSetDevice(p); } } restore{prev};
    if ((hd.flags & 1) && hd.n > 0) {
        std::vector<long long> ids((size_t)hd.n);
        if (fread(ids.data(), sizeof(long long), (size_t)hd.n, f) != (size_t)hd.n) return fail(RAG_ERR_INVALID, "truncated snapshot (ids)");
        RAG_HIP_CHECK(hipMalloc((void**)&h->ids_dev, (size_t)h->cap * sizeof(long long)));
        RAG_HIP_CHECK(hipMemcpy(h->ids_dev, ids.data(), (size_t)hd.n * sizeof(long long), hipMemcpyHostToDevice));
    }
    if ((hd.flags & 2) && hd.n > 0) {
        std::vector<uint8_t> act((size_t)hd.n);
        if (fread(act.data(), 1, (size_t)hd.n, f) != (size_t)hd.n) return fail(RAG_ERR_INVALID, "truncated snapshot (active)");
        RAG_HIP_CHECK(hipMalloc((void**)&h->active, (size_t)std::max<long long>(h->cap, 1)));
        RAG_HIP_CHECK(hipMemset(h->active, 1, (size_t)std::max<long long>(h->cap, 1)));
        RAG_HIP_CHECK(hipMemcpy(h->active, act.data(), (size_t)hd.n, hipMemcpyHostToDevice));
    }
    guard.keep = true;
    *out = h;
    return RAG_OK;
}
