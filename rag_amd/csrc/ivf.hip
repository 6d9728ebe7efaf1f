// Inverted-file index with flat (uncompressed) lists and an inner-product coarse
// quantiser -- FAISS IndexIVFFlat semantics: a row is stored in the list of the
// centroid with the largest inner product; a search scores the query against all
// centroids, takes the nprobe best lists and scans only their rows.
//
// Both steps reuse the exhaustive kernels.  Coarse assignment and probe selection
// are top-k searches of (rows | queries) against the centroid table (GEMM path for
// batches, scan path for single queries).  The list scan is flat_scan_kernel in its
// per-block row-range mode: blockIdx.y = query, blockIdx.x = (probed list, slice).
//
// Storage.  Rows are physically grouped by list so a list is one contiguous HBM
// range; rows that arrived since the last search sit unsorted at the tail and are
// folded in (one device gather) by the next search.  Keys carry the INSERTION
// position (row_alias), so ids, masks, ties and results are the same as for a flat
// index of the same rows.
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>

#include "index.h"
#include "topk_keys.h"

namespace rag {

int flat_search_prepared(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask,
                         float* D, long long* I, hipStream_t s);
int merge_cap(long long per_query, int k);

__global__ void gather_packed_rows_kernel(const float* src, const long long* src_row, long long n, int ld4, float* dst) {
    // one wave per destination row
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* s4 = reinterpret_cast<const f4*>(src);
    f4* d4 = reinterpret_cast<f4*>(dst);
    for (long long r = wave; r < n; r += nw) {
        const long long sr = src_row[r];
        for (int c = lane; c < ld4; c += 64) d4[r * ld4 + c] = s4[sr * ld4 + c];
    }
}

// the coarse step of a few queries: every list scored, then one selection workgroup per query (direct_topk.hip)
bool direct_topk_supported(const rag_index* h, long long nq, int k);
size_t direct_topk_workspace(const rag_index* h, long long nq);
int direct_topk_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask,
                       float* D, long long* I, float* scores, hipStream_t s);

constexpr long long IVF_QCHUNK = 32768;     // queries per launch (grid.y) and workspace bound

// slices per probed list: enough blocks to fill the chip, but a slice keeps >= 32 tiles of
// an average list
static long long ivf_slices(const rag_index* h, long long nq, int nprobe) {
    const int tile = scan_rows_per_tile(h->ld / 4);
    const long long avg_tiles = std::max<long long>(1, (h->n / std::max(1, h->ivf.nlist)) / tile);
    static const int slice_tiles = std::max(1, getenv("RAG_IVF_SLICE_TILES") ? atoi(getenv("RAG_IVF_SLICE_TILES")) : 32);   // measurement knob
    static const int max_blocks = std::max(64, getenv("RAG_IVF_SCAN_BLOCKS") ? atoi(getenv("RAG_IVF_SCAN_BLOCKS")) : 2048);
    long long slices = max_blocks / std::max<long long>(1, nq * nprobe);
    slices = std::min<long long>(slices, std::max<long long>(1, avg_tiles / slice_tiles));
    return slices < 1 ? 1 : slices;
}

bool gemm_scores_shape_ok(const rag_index* h, long long nq, int k);                          // gemm_topk.hip
size_t gemm_scores_workspace(const rag_index* h, long long nq);
int gemm_scores_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask, float* D,
                       long long* I, hipStream_t s);
bool ivf_listmajor_supported(const rag_index* h, long long nq, int k, int nprobe);            // gemm_topk.hip
size_t ivf_listmajor_workspace(const rag_index* h, long long nq, int k, int nprobe);
int ivf_listmajor_scan(rag_index* h, long long nq, const float* qbuf, int k, float min_score, int nprobe,
                       const long long* coarse_lists, const uint8_t* mask, const uint8_t* mask2, float* D, long long* I, hipStream_t s);

// bytes the list scan needs in h->ws after the padded queries
size_t ivf_search_workspace(const rag_index* h, long long nq, int k, int nprobe) {
    if (nprobe > h->ivf.nlist) nprobe = h->ivf.nlist;
    if (nprobe < 1) nprobe = 1;
    if (ivf_listmajor_supported(h, nq, k, nprobe)) return ivf_listmajor_workspace(h, nq, k, nprobe);
    const long long waves = (long long)nprobe * ivf_slices(h, nq, nprobe) * 4;
    return (size_t)std::min(nq, IVF_QCHUNK) * waves * k * sizeof(u64);
}

void ivf_free(rag_index* h) {
    IvfState& v = h->ivf;
    if (v.quantizer) rag_index_free(v.quantizer);
    if (v.list_begin_dev) (void)hipFree(v.list_begin_dev);
    if (v.orig_dev) (void)hipFree(v.orig_dev);
    if (v.pos_dev) (void)hipFree(v.pos_dev);
    v.ws.release();
    v = IvfState();
}

// list of every new row = argmax inner product over the centroids (ties: lower list)
int ivf_assign_new_rows(rag_index* h, long long first, long long count) {
    IvfState& v = h->ivf;
    if (!v.quantizer) return fail(RAG_ERR_INVALID, "IVF index is not trained");
    v.assign.resize((size_t)(first + count));
    if (v.preassigned) {
        // rows of a snapshot come back with the lists they had
        for (long long i = 0; i < count; ++i) {
            const int l = v.preassigned[(size_t)(first + i)];
            if (l < 0 || l >= v.nlist) return fail(RAG_ERR_INVALID, "snapshot names a list outside [0, nlist)");
            v.assign[(size_t)(first + i)] = l;
        }
        return RAG_OK;
    }
    // the rows go through the centroid search 65536 at a time (its workspace is sized per launch); every
    // chunk's answers stay on the device and come back in one copy behind the last launch
    const long long chunk = 65536;
    const size_t db = ((size_t)std::min(chunk, count) * sizeof(float) + 255) & ~(size_t)255;
    if (int rc = v.ws.ensure(db + (size_t)count * sizeof(long long))) return rc;
    float* D = static_cast<float*>(v.ws.ptr);
    long long* I = reinterpret_cast<long long*>(static_cast<char*>(v.ws.ptr) + db);
    for (long long c0 = 0; c0 < count; c0 += chunk) {
        const long long c = std::min(chunk, count - c0);
        // new rows sit at the tail of X in arrival order, already padded to ld
        const float* rows = h->X + (size_t)(first + c0) * h->ld;
        if (int rc = flat_search_prepared(v.quantizer, c, rows, 1, -FLT_MAX, nullptr, D, I + c0, h->stream)) return rc;
    }
    std::vector<long long> host_i((size_t)count);
    RAG_HIP_CHECK(hipMemcpyAsync(host_i.data(), I, (size_t)count * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
    for (long long i = 0; i < count; ++i) {
        const long long l = host_i[(size_t)i];
        v.assign[(size_t)(first + i)] = l < 0 ? 0 : (int)l;     // a NaN row scores nowhere: list 0
    }
    return RAG_OK;
}

// rows from[i] -> rows to[i] of the same buffer (the two sets do not overlap); one wave per row
__global__ void move_rows_kernel(float* X, const long long* from, const long long* to, long long m, int ld4) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* x4 = reinterpret_cast<f4*>(X);
    for (long long i = wave; i < m; i += nw) {
        const long long s = from[i], d = to[i];
        for (int c = lane; c < ld4; c += 64) x4[d * ld4 + c] = x4[s * ld4 + c];
    }
}

// Fold the unsorted tail into list order, IN PLACE: the rows are permuted inside their buffer one destination chunk at a
// time through a staging block of at most ~1 GB, so packing 51 GB of rows needs 52 GB, not 102.  Per chunk [c0, c1) of
// the final order: (1) the rows that belong there are gathered into the staging block (they sit in the chunk itself or
// behind it -- everything in front is final); (2) the rows that sit in the chunk but belong behind it move into the
// places step 1 vacated behind the chunk (as many as it vacated); (3) the staging block is copied over the chunk.
// The host keeps who-is-where up to date; every row moves at most three times.
static int ivf_repack(rag_index* h) {
    IvfState& v = h->ivf;
    const long long n = h->n;
    if (v.n_packed == n) return RAG_OK;
    // current layout: storage position p holds insertion row orig[p]; tail rows are their own position
    v.orig.resize((size_t)n);
    v.pos.resize((size_t)n);
    for (long long p = v.n_packed; p < n; ++p) { v.orig[(size_t)p] = p; v.pos[(size_t)p] = p; }
    // counting sort of insertion rows by list, stable in insertion order
    std::vector<long long> begin((size_t)v.nlist + 1, 0);
    for (long long o = 0; o < n; ++o) begin[(size_t)v.assign[(size_t)o] + 1]++;
    for (int l = 0; l < v.nlist; ++l) begin[(size_t)l + 1] += begin[(size_t)l];
    std::vector<long long> cursor(begin.begin(), begin.end() - 1);
    std::vector<long long> new_orig((size_t)n), new_pos((size_t)n);
    for (long long o = 0; o < n; ++o) {
        const long long p = cursor[(size_t)v.assign[(size_t)o]]++;
        new_orig[(size_t)p] = o;
        new_pos[(size_t)o] = p;
    }
    static const long long chunk_bytes = std::max(1, getenv("RAG_IVF_PACK_CHUNK_MB") ? atoi(getenv("RAG_IVF_PACK_CHUNK_MB")) : 1024) * (1LL << 20);
    const long long B = std::max<long long>(1, std::min<long long>(n, chunk_bytes / ((long long)h->ld * 4)));
    const size_t row_bytes = (size_t)h->ld * sizeof(float);
    const size_t idx_bytes = (((size_t)B * sizeof(long long)) + 255) & ~(size_t)255;
    if (int rc = v.ws.ensure((size_t)B * row_bytes + 3 * idx_bytes)) return rc;
    float* stage = static_cast<float*>(v.ws.ptr);
    long long* d_src = reinterpret_cast<long long*>(static_cast<char*>(v.ws.ptr) + (size_t)B * row_bytes);
    long long* d_from = reinterpret_cast<long long*>(reinterpret_cast<char*>(d_src) + idx_bytes);
    long long* d_to = reinterpret_cast<long long*>(reinterpret_cast<char*>(d_from) + idx_bytes);
    std::vector<long long> src, from, to;
    for (long long c0 = 0; c0 < n; c0 += B) {
        const long long c1 = std::min(n, c0 + B), cnt = c1 - c0;
        src.resize((size_t)cnt);
        from.clear();
        to.clear();
        bool identity = true;
        for (long long p = c0; p < c1; ++p) {
            const long long s = v.pos[(size_t)new_orig[(size_t)p]];        // where the row that belongs at p sits now
            src[(size_t)(p - c0)] = s;
            identity = identity && s == p;
            if (s >= c1) to.push_back(s);                                  // a place behind the chunk falls vacant
        }
        if (identity) continue;                                            // the chunk is in order already (re-packs: the packed head)
        for (long long q = c0; q < c1; ++q)
            if (new_pos[(size_t)v.orig[(size_t)q]] >= c1) from.push_back(q);   // sits in the chunk, belongs behind it
        if (from.size() != to.size()) return fail(RAG_ERR_HIP, "internal: the in-place pack lost count of its rows");
        RAG_HIP_CHECK(hipMemcpyAsync(d_src, src.data(), (size_t)cnt * sizeof(long long), hipMemcpyHostToDevice, h->stream));
        long long blocks = std::min<long long>((cnt + 3) / 4, 8192);
        hipLaunchKernelGGL(gather_packed_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, h->X, d_src, cnt, h->ld / 4, stage);
        RAG_HIP_CHECK(hipGetLastError());
        const long long m = (long long)from.size();
        if (m > 0) {
            RAG_HIP_CHECK(hipMemcpyAsync(d_from, from.data(), (size_t)m * sizeof(long long), hipMemcpyHostToDevice, h->stream));
            RAG_HIP_CHECK(hipMemcpyAsync(d_to, to.data(), (size_t)m * sizeof(long long), hipMemcpyHostToDevice, h->stream));
            blocks = std::min<long long>((m + 3) / 4, 8192);
            hipLaunchKernelGGL(move_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, h->X, d_from, d_to, m, h->ld / 4);
            RAG_HIP_CHECK(hipGetLastError());
        }
        RAG_HIP_CHECK(hipMemcpyAsync(h->X + (size_t)c0 * h->ld, stage, (size_t)cnt * row_bytes, hipMemcpyDeviceToDevice, h->stream));
        // the index arrays above are reused by the next chunk: its uploads wait for this chunk's kernels here
        RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
        // bookkeeping: the moved rows first, then the chunk's final occupants
        for (long long j = 0; j < m; ++j) {
            const long long o = v.orig[(size_t)from[(size_t)j]];
            v.orig[(size_t)to[(size_t)j]] = o;
            v.pos[(size_t)o] = to[(size_t)j];
        }
        for (long long p = c0; p < c1; ++p) {
            const long long o = new_orig[(size_t)p];
            v.orig[(size_t)p] = o;
            v.pos[(size_t)o] = p;
        }
    }
    // (a table that fails to grow is left empty, never pointing at freed memory)
    if (v.orig_cap < n) {
        if (v.orig_dev) (void)hipFree(v.orig_dev);
        v.orig_dev = nullptr;
        const long long cap = std::max(n, v.orig_cap + v.orig_cap / 2);
        v.orig_cap = 0;
        RAG_HIP_CHECK(hipMalloc((void**)&v.orig_dev, (size_t)cap * sizeof(long long)));
        v.orig_cap = cap;
    }
    RAG_HIP_CHECK(hipMemcpy(v.orig_dev, v.orig.data(), (size_t)n * sizeof(long long), hipMemcpyHostToDevice));
    if (v.pos_cap < n) {
        if (v.pos_dev) (void)hipFree(v.pos_dev);
        v.pos_dev = nullptr;
        const long long cap = std::max(n, v.pos_cap + v.pos_cap / 2);
        v.pos_cap = 0;
        RAG_HIP_CHECK(hipMalloc((void**)&v.pos_dev, (size_t)cap * sizeof(long long)));
        v.pos_cap = cap;
    }
    RAG_HIP_CHECK(hipMemcpy(v.pos_dev, v.pos.data(), (size_t)n * sizeof(long long), hipMemcpyHostToDevice));
    RAG_HIP_CHECK(hipMemcpy(v.list_begin_dev, begin.data(), ((size_t)v.nlist + 1) * sizeof(long long), hipMemcpyHostToDevice));
    v.n_packed = n;
    return RAG_OK;
}

int ivf_reconstruct_rows(rag_index* h, long long n, const int64_t* rows, std::vector<long long>& storage_rows) {
    IvfState& v = h->ivf;
    storage_rows.resize((size_t)n);
    for (long long i = 0; i < n; ++i) {
        const long long o = rows[i];
        storage_rows[(size_t)i] = o < v.n_packed ? v.pos[(size_t)o] : o;   // tail rows are still in arrival order
    }
    return RAG_OK;
}

int ivf_search_device(rag_index* h, long long nq, const float* qbuf, int k, float min_score, int nprobe,
                      const uint8_t* mask, const uint8_t* mask2, float* D, long long* I, hipStream_t s) {
    IvfState& v = h->ivf;
    if (!v.quantizer) return fail(RAG_ERR_INVALID, "IVF index is not trained");
    if (nprobe < 1) return fail(RAG_ERR_INVALID, "nprobe must be >= 1");
    if (nprobe > v.nlist) nprobe = v.nlist;
    if (nprobe > 1024) return fail(RAG_ERR_INVALID, "nprobe must be <= 1024");
    if (v.n_packed != h->n) {
        RAG_HIP_CHECK(hipStreamSynchronize(s));
        if (int rc = ivf_repack(h)) return rc;
    }
    const long long n = h->n;
    // --- coarse: the nprobe best lists of every query --------------------------------------
    const size_t cdb = ((size_t)nq * nprobe * sizeof(float) + 255) & ~(size_t)255;
    const size_t cib = ((size_t)nq * nprobe * sizeof(long long) + 255) & ~(size_t)255;
    const size_t mb = 0;          // masks are read through row_alias where a row is tested: no packed copy
    const bool direct = direct_topk_supported(v.quantizer, nq, nprobe);
    const size_t sb = direct ? direct_topk_workspace(v.quantizer, nq) : 0;
    if (int rc = v.ws.ensure(cdb + cib + mb + sb)) return rc;
    float* cD = static_cast<float*>(v.ws.ptr);
    long long* cI = reinterpret_cast<long long*>(static_cast<char*>(v.ws.ptr) + cdb);
    if (direct) {
        float* scores = reinterpret_cast<float*>(static_cast<char*>(v.ws.ptr) + cdb + cib + mb);
        if (int rc = direct_topk_search(v.quantizer, nq, qbuf, nprobe, -FLT_MAX, nullptr, cD, cI, scores, s)) return rc;
    } else if (gemm_scores_shape_ok(v.quantizer, nq, nprobe)) {
        // a large batch against a short centroid table: every (query, list) score written by the matrix-core kernel,
        // then one selection workgroup per query
        const size_t cq = ((size_t)nq * v.quantizer->ld * sizeof(float) + 255) & ~(size_t)255;
        if (int rc = v.quantizer->ws.ensure(cq + gemm_scores_workspace(v.quantizer, nq))) return rc;
        if (int rc = gemm_scores_search(v.quantizer, nq, qbuf, nprobe, -FLT_MAX, nullptr, cD, cI, s)) return rc;
    } else if (int rc = flat_search_prepared(v.quantizer, nq, qbuf, nprobe, -FLT_MAX, nullptr, cD, cI, s)) {
        return rc;
    }
    // --- list scan ------------------------------------------------------------------------------
    if (ivf_listmajor_supported(h, nq, k, nprobe)) {
        if (int rc = ivf_listmajor_scan(h, nq, qbuf, k, min_score, nprobe, cI, mask, mask2, D, I, s)) return rc;
        h->last_path = "ivf-listmajor";
        return RAG_OK;
    }
    const long long slices = ivf_slices(h, nq, nprobe);
    const int gx = (int)(nprobe * slices);
    const size_t qbytes = ((size_t)nq * h->ld * sizeof(float) + 255) & ~(size_t)255;
    const long long qchunk = IVF_QCHUNK;
    if (h->ws.bytes < qbytes + ivf_search_workspace(h, nq, k, nprobe))
        return fail(RAG_ERR_INVALID, "internal: workspace not sized for the IVF scan");
    u64* part = reinterpret_cast<u64*>(static_cast<char*>(h->ws.ptr) + qbytes);
    for (long long q0 = 0; q0 < nq; q0 += qchunk) {
        const long long qc = std::min(qchunk, nq - q0);
        ScanArgs a = {};
        a.X = h->X;
        a.Q = qbuf + (size_t)q0 * h->ld;
        a.mask = mask;
        a.mask2 = mask2;
        a.mask_aliased = 1;
        a.part = part;
        a.n = n;
        a.d4 = h->ld / 4;
        a.nch = 0;
        a.k = k;
        a.capw = 0;
        a.nt = 1;
        a.ivf_lists = cI + (size_t)q0 * nprobe;
        a.list_begin = v.list_begin_dev;
        a.row_alias = v.orig_dev;
        a.ivf_nprobe = nprobe;
        a.ivf_slices = (int)slices;
        a.grid_y = (int)qc;
        if (int rc = profile_begin(h, s)) return rc;
        RAG_HIP_CHECK(launch_flat_scan(a, 1, gx, s));
        if (int rc = profile_end(h, s)) return rc;
        MergeArgs m = {};
        m.keys = part;
        m.per_query = (long long)gx * scan_lists_per_block(k) * k;
        m.k = k;
        m.cap = merge_cap(m.per_query, k);
        m.min_score = min_score;
        m.row_ids = h->ids_dev;
        m.id_base = h->id_base;
    m.rows_out = h->merge_rows_out;
        m.slot_ids = nullptr;
        m.slot_stride = 0;
        m.D = D + (size_t)q0 * k;
        m.I = I + (size_t)q0 * k;
        RAG_HIP_CHECK(launch_merge_keys(m, qc, s));
    }
    h->last_path = "ivf";
    return RAG_OK;
}

}  // namespace rag

using namespace rag;

extern "C" int rag_index_train_ivf(rag_index_t* h, int nlist, const float* centroids, int64_t nsample, const float* sample) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    RAG_LOCK_INDEX(h);
    if (h->kind != RAG_KIND_IVF) return fail(RAG_ERR_INVALID, "not an IVF index");
    if (h->n != 0) return fail(RAG_ERR_INVALID, "train before adding rows");
    if (nlist < 1 || nlist > (1 << 22)) return fail(RAG_ERR_INVALID, "nlist must be in [1, 4194304]");
    const float* c = centroids;
    if (!c) {
        if (!sample || nsample < nlist) return fail(RAG_ERR_INVALID, "need centroids, or a sample of at least nlist rows");
        c = sample;   // the first nlist sample rows are the centroids
    }
    ivf_free(h);
    IvfState& v = h->ivf;
    rag_index* qz = nullptr;
    // centroids are compared as given (never normalised), whatever the index metric
    if (int rc = rag_index_create(h->dim, RAG_METRIC_IP, RAG_KIND_FLAT, h->device, &qz)) return rc;
    if (int rc = rag_index_add(qz, nlist, c, nullptr)) { rag_index_free(qz); return rc; }
    // assignment of new rows is a large batch against the centroid table: bf16 first pass,
    // exact fp32 re-score (same lists as the fp32 kernel would give)
    {
        const char* e = getenv("RAG_IVF_ASSIGN_BF16");
        if (!(e && *e == '0') && (qz->ld % 64) == 0) qz->prefilter = 1;
    }
    v.quantizer = qz;
    v.nlist = nlist;
    int prev = 0;
    RAG_HIP_CHECK(hipGetDevice(&prev));
    RAG_HIP_CHECK(hipSetDevice(h->device));
    hipError_t e = hipMalloc((void**)&v.list_begin_dev, ((size_t)nlist + 1) * sizeof(long long));
    if (e == hipSuccess) e = hipMemset(v.list_begin_dev, 0, ((size_t)nlist + 1) * sizeof(long long));
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return fail(RAG_ERR_OOM, std::string("list offsets: ") + hipGetErrorString(e));
    return RAG_OK;
}
