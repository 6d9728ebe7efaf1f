// C ABI of librag_hip.so: index lifetime, ingest, search dispatch, result merge.
// The declarations (and the reference interfaces they stand in for) are in
// include/rag_hip.h.
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "index.h"
#include "topk_keys.h"

namespace rag {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

int Workspace::ensure(size_t need) {
    if (need <= bytes) return RAG_OK;
    size_t want = std::max(need, bytes + bytes / 2);
    want = (want + 255) & ~(size_t)255;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return fail(RAG_ERR_OOM, "workspace hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(e));
    if (ptr) (void)hipFree(ptr);
    ptr = p;
    bytes = want;
    return RAG_OK;
}
void Workspace::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

static thread_local Workspace g_thread_ws[TWS_KINDS][64];
Workspace& thread_workspace(int which, int device) { return g_thread_ws[which][device & 63]; }

// Page-locked host memory of the calling thread for the small transfers of the host-pointer entry points (a query in, a
// result list out): a copy from or to pageable memory is staged by the runtime and waits for it, a copy on page-locked
// memory is one DMA command.  Grow-only, handed back by rag_thread_release().
struct PinnedBlock {
    void* ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return RAG_OK;
        size_t want = std::max<size_t>(need, 64 * 1024);
        want = (want + 4095) & ~(size_t)4095;
        void* p = nullptr;
        const hipError_t e = hipHostMalloc(&p, want, hipHostMallocPortable);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(RAG_ERR_OOM, "hipHostMalloc(" + std::to_string(want) + "): " + hipGetErrorString(e));
        }
        if (ptr) (void)hipHostFree(ptr);
        ptr = p;
        bytes = want;
        return RAG_OK;
    }
    void release() {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
};
static thread_local PinnedBlock g_thread_pinned;
int thread_pinned(size_t need, char** ptr) {
    if (int rc = g_thread_pinned.ensure(need)) return rc;
    *ptr = static_cast<char*>(g_thread_pinned.ptr);
    return RAG_OK;
}
// the block for a result copy at the end of a call whose inputs may still be travelling from it: growing replaces the
// block, so the stream is drained first (only the first calls of a thread grow it)
static int pinned_for_results(size_t need, hipStream_t s) {
    if (need <= g_thread_pinned.bytes) return RAG_OK;
    RAG_HIP_CHECK(hipStreamSynchronize(s));
    return g_thread_pinned.ensure(need);
}
constexpr size_t PINNED_STAGE_MAX = 4u << 20;      // larger transfers go from / to the caller's memory directly

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// a few KB between device memory and the calling thread's page-locked block (mapped into the device's address space): as a
// kernel on the search's stream it costs a launch, not a DMA command with its own completion signalling
__global__ void copy_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ void fill_results_kernel(float* D, long long* I, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        D[i] = -FLT_MAX;
        I[i] = -1;
    }
}
__global__ void and_masks_kernel(const uint8_t* a, const uint8_t* b, long long n, uint8_t* out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = (a[i] != 0 && b[i] != 0) ? 1 : 0;
}
__global__ void scatter_bytes_kernel(uint8_t* dst, const long long* rows, long long n, long long limit, uint8_t v) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = rows[i];
        if (r >= 0 && r < limit) dst[r] = v;
    }
}
__global__ void iota_ids_kernel(long long* ids, long long begin, long long end, long long base) {
    for (long long i = begin + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < end; i += (long long)gridDim.x * blockDim.x)
        ids[i] = base + i;
}

// The event that marks the end of a search costs a marker packet on the device (2.4-3 us between two searches).  It is only
// needed when the NEXT search arrives on another stream, so where the stream is known to be there at that time -- the
// index's own stream, or any stream once the caller has promised so (rag_index_set_stream_lifetime) -- it is recorded
// then, on the old stream's present end (which lies behind that search's last launch), and not at all between searches
// that follow each other on one stream.
int ws_acquire(rag_index* h, hipStream_t s) {
    if (!h->ws_event) RAG_HIP_CHECK(hipEventCreateWithFlags(&h->ws_event, hipEventDisableTiming));
    if (h->ws_pending && h->ws_stream != s) {
        if (h->ws_unmarked) {
            RAG_HIP_CHECK(hipEventRecord(h->ws_event, h->ws_stream));
            h->ws_unmarked = false;
        }
        RAG_HIP_CHECK(hipStreamWaitEvent(s, h->ws_event, 0));
    }
    return RAG_OK;
}
int ws_release(rag_index* h, hipStream_t s) {
    static const int eager = getenv("RAG_WS_EAGER_EVENTS") ? atoi(getenv("RAG_WS_EAGER_EVENTS")) : 0;      // measurement knob
    if (!eager && (s == h->stream || h->streams_outlive)) {
        h->ws_unmarked = true;
    } else {
        RAG_HIP_CHECK(hipEventRecord(h->ws_event, s));
        h->ws_unmarked = false;
    }
    h->ws_stream = s;
    h->ws_pending = true;
    return RAG_OK;
}

int profile_begin(rag_index* h, hipStream_t s) {
    if (!h->profiling) return RAG_OK;
    if (h->prof_used + 2 > h->prof_events.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            RAG_HIP_CHECK(hipEventCreate(&e));
            h->prof_events.push_back(e);
        }
    }
    RAG_HIP_CHECK(hipEventRecord(h->prof_events[h->prof_used], s));
    return RAG_OK;
}
int profile_end(rag_index* h, hipStream_t s) {
    if (!h->profiling) return RAG_OK;
    RAG_HIP_CHECK(hipEventRecord(h->prof_events[h->prof_used + 1], s));
    h->prof_used += 2;
    return RAG_OK;
}

static int grid_for(long long n, int block = 256, int maxb = 4096) {
    long long b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > maxb) b = maxb;
    return (int)b;
}

// make room for `extra` more rows (and their ids / active bytes)
static int ensure_rows(rag_index* h, long long extra) {
    const long long need = h->n + extra;
    if (need <= h->cap) return RAG_OK;
    if (h->adopted) return fail(RAG_ERR_INVALID, "index uses an adopted device buffer and cannot grow");
    long long ncap = std::max(need, h->reserve_hint);
    if (h->cap > 0 && ncap < h->cap + h->cap / 2 && need > h->reserve_hint) ncap = h->cap + h->cap / 2;
    float* nx = nullptr;
    RAG_HIP_CHECK(hipMalloc((void**)&nx, (size_t)ncap * h->ld * sizeof(float)));
    if (h->n > 0) RAG_HIP_CHECK(hipMemcpy(nx, h->X, (size_t)h->n * h->ld * sizeof(float), hipMemcpyDeviceToDevice));
    if (h->X) (void)hipFree(h->X);
    h->X = nx;
    if (h->ids_dev) {
        long long* ni = nullptr;
        RAG_HIP_CHECK(hipMalloc((void**)&ni, (size_t)ncap * sizeof(long long)));
        if (h->n > 0) RAG_HIP_CHECK(hipMemcpy(ni, h->ids_dev, (size_t)h->n * sizeof(long long), hipMemcpyDeviceToDevice));
        (void)hipFree(h->ids_dev);
        h->ids_dev = ni;
    }
    if (h->active) {
        uint8_t* na = nullptr;
        RAG_HIP_CHECK(hipMalloc((void**)&na, (size_t)ncap));
        RAG_HIP_CHECK(hipMemset(na, 1, (size_t)ncap));
        if (h->n > 0) RAG_HIP_CHECK(hipMemcpy(na, h->active, (size_t)h->n, hipMemcpyDeviceToDevice));
        (void)hipFree(h->active);
        h->active = na;
    }
    h->cap = ncap;
    return RAG_OK;
}

// record ids for rows [n, n+count) (ids may be nullptr)
static int append_ids(rag_index* h, long long count, const int64_t* ids) {
    if (!ids && !h->ids_dev) return RAG_OK;
    if (!h->ids_dev) {
        RAG_HIP_CHECK(hipMalloc((void**)&h->ids_dev, (size_t)h->cap * sizeof(long long)));
        if (h->n > 0) {
            hipLaunchKernelGGL(iota_ids_kernel, dim3(grid_for(h->n)), dim3(256), 0, h->stream, h->ids_dev, 0LL, h->n, h->id_base);
            RAG_HIP_CHECK(hipGetLastError());
        }
    }
    if (ids) {
        RAG_HIP_CHECK(hipMemcpyAsync(h->ids_dev + h->n, ids, (size_t)count * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    } else {
        hipLaunchKernelGGL(iota_ids_kernel, dim3(grid_for(count)), dim3(256), 0, h->stream, h->ids_dev, h->n, h->n + count, h->id_base);
        RAG_HIP_CHECK(hipGetLastError());
    }
    RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
    return RAG_OK;
}

int ivf_assign_new_rows(rag_index* h, long long first, long long count);   // ivf.hip
int ivf_search_device(rag_index* h, long long nq, const float* qbuf, int k, float min_score, int nprobe,
                      const uint8_t* mask, const uint8_t* mask2, float* D, long long* I, hipStream_t s);
void ivf_free(rag_index* h);
size_t ivf_search_workspace(const rag_index* h, long long nq, int k, int nprobe);
int gemm_topk_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask,
                     float* D, long long* I, hipStream_t s);                 // gemm_topk.hip
bool gemm_topk_supported(const rag_index* h, long long nq, int k);
bool gemm_scores_supported(const rag_index* h, long long nq, int k);
size_t gemm_scores_workspace(const rag_index* h, long long nq);
int gemm_scores_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask, float* D,
                       long long* I, hipStream_t s);
bool gemm_segments_supported(const rag_index* h, long long nq, int k);
size_t gemm_segments_workspace(const rag_index* h, long long nq, int k);
int gemm_segments_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask, float* D,
                         long long* I, hipStream_t s);
bool prefilter_scan_supported(const rag_index* h, long long nq, int k);       // gemm_topk.hip
size_t prefilter_scan_workspace(const rag_index* h, int k);
int prefilter_scan_search(rag_index* h, const float* qbuf, int k, float min_score, const uint8_t* mask,
                          float* D, long long* I, hipStream_t s);
size_t gemm_topk_workspace(const rag_index* h, long long nq, int k);
bool narrow_topk_supported(const rag_index* h, long long nq, int k);          // narrow_topk.hip
size_t narrow_topk_workspace(const rag_index* h, long long nq, int k);
size_t fanout_workspace(long long nq, long long nseg, int k, int limit);                  // fanout.hip
int fanout_fuse_enqueue(const float* D, const long long* R, const double* weights_host, const double* weights_dev, int nq,
                        long long nseg, int k, int limit, char* ws, long long** rows_out, double** scores_out, int** strat_out,
                        int** n_out, hipStream_t s);
bool direct_topk_supported(const rag_index* h, long long nq, int k);                      // direct_topk.hip
size_t direct_topk_workspace(const rag_index* h, long long nq);
int direct_topk_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask,
                       float* D, long long* I, float* scores, hipStream_t s);
int narrow_topk_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask,
                       float* D, long long* I, hipStream_t s);
int ivf_reconstruct_rows(rag_index* h, long long n, const int64_t* rows, std::vector<long long>& storage_rows);

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// LDS capacity (keys) of merge_keys_kernel: everything at once when the input is small
int merge_cap(long long per_query, int k) {
    if (per_query <= 2048) return std::max(64, pow2_ceil((int)std::max<long long>(per_query, 2 * k)));
    return std::max(2048, 2 * pow2_ceil(k));
}

// blocks for the scan: enough waves to fill the chip, not more lists than rows justify
int scan_blocks(const rag_index* h, int k) {
    static const int max_blocks = env_int("RAG_SCAN_BLOCKS", 512);
    const int tile = scan_rows_per_tile(h->ld / 4);
    const long long ntiles = (h->n + tile - 1) / tile;
    // a wave should have a few tiles to walk before another wave (and its partial list, which
    // the merge has to read) is worth adding -- more of them when the lists are long; at most 2
    // blocks per CU.  Measured: 100k x 384, k 10: 8 tiles 48.9 us per query, 16 tiles 51.7;
    // 16384 x 1024, k 64: 8 tiles 44.1 us, 16 tiles 43.6, 4 tiles 53.7.
    static const int min_tiles_env = env_int("RAG_SCAN_MIN_TILES", 0);
    const int min_tiles = min_tiles_env > 0 ? min_tiles_env : (k <= 16 ? 8 : 16);
    long long waves = (ntiles + min_tiles - 1) / min_tiles;
    long long blocks = (waves + 3) / 4;
    if (blocks < 1) blocks = 1;
    if (blocks > max_blocks) blocks = max_blocks;
    return (int)blocks;
}

size_t flat_search_workspace(const rag_index* h, long long nq, int k);

static int flat_scan_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask,
                            float* D, long long* I, hipStream_t s) {
    const int blocks = scan_blocks(h, k);
    const int waves = blocks * 4;
    const int group_max = (k > 64) ? 1 : 8;
    // workspace: partial keys for one group of queries (after the padded queries)
    const size_t qbytes = ((size_t)nq * h->ld * sizeof(float) + 255) & ~(size_t)255;
    const size_t pbytes = (size_t)group_max * waves * k * sizeof(u64);
    // (the caller placed qbuf at ws.ptr; keep it there)
    if (int rc = h->ws.ensure(qbytes + pbytes)) return rc;
    u64* part = reinterpret_cast<u64*>(static_cast<char*>(h->ws.ptr) + qbytes);
    const long long lists = (long long)blocks * scan_lists_per_block(k);
    const int cap = merge_cap(lists * k, k);
    long long q0 = 0;
    while (q0 < nq) {
        // queries per pass: 8, or the next power of two above what is left (a pass costs one
        // read of the corpus however many of its query slots are used; the surplus slots repeat
        // the last query and their lists are never merged)
        int g = 1;
        const long long left = nq - q0;
        if (k <= 64) g = left >= 8 ? 8 : pow2_ceil((int)left);
        const int real = (int)std::min<long long>(left, g);
        ScanArgs a = {};
        a.X = h->X;
        a.Q = qbuf + q0 * h->ld;
        a.mask = mask;
        a.part = part;
        a.n = h->n;
        a.d4 = h->ld / 4;
        a.nch = 0;
        a.k = k;
        a.capw = 0;
        a.ivf_lists = nullptr;
        a.list_begin = nullptr;
        a.row_alias = nullptr;
        a.ivf_nprobe = 0;
        a.ivf_slices = 1;
        a.grid_y = 1;
        a.nq_valid = real;
        // rows larger than the Infinity Cache are streamed; smaller corpora stay cacheable
        static const int nt_mode = env_int("RAG_SCAN_NT", -1);
        a.nt = nt_mode >= 0 ? nt_mode : ((size_t)h->n * h->ld * sizeof(float) > (192ull << 20) ? 1 : 0);
        if (int rc = profile_begin(h, s)) return rc;
        RAG_HIP_CHECK(launch_flat_scan(a, g, blocks, s));
        if (int rc = profile_end(h, s)) return rc;
        MergeArgs m = {};
        m.keys = part;
        m.per_query = lists * k;
        m.k = k;
        m.cap = cap;
        m.min_score = min_score;
        m.row_ids = h->ids_dev;
        m.id_base = h->id_base;
    m.rows_out = h->merge_rows_out;
        m.slot_ids = nullptr;
        m.slot_stride = 0;
        m.D = D + q0 * k;
        m.I = I + q0 * k;
        RAG_HIP_CHECK(launch_merge_keys(m, real, s));
        q0 += real;
    }
    h->last_path = "scan";
    return RAG_OK;
}

// bytes the exhaustive search needs in h->ws after the padded queries
// 5..32 queries without the bf16 first pass: the register-streaming kernel (one corpus pass at the HBM rate)
static bool use_narrow(const rag_index* h, long long nq, int k) { return !h->prefilter && narrow_topk_supported(h, nq, k); }
// 1 to 4 queries over a corpus of at most 32768 rows: every row scored, one selection workgroup per query
static bool use_direct(const rag_index* h, long long nq, int k) { return !h->prefilter && direct_topk_supported(h, nq, k); }

size_t flat_search_workspace(const rag_index* h, long long nq, int k) {
    if (use_direct(h, nq, k)) return direct_topk_workspace(h, nq);
    if (gemm_scores_supported(h, nq, k)) return gemm_scores_workspace(h, nq);
    if (use_narrow(h, nq, k)) return narrow_topk_workspace(h, nq, k);
    if (gemm_topk_supported(h, nq, k)) return gemm_topk_workspace(h, nq, k);
    if (gemm_segments_supported(h, nq, k)) return gemm_segments_workspace(h, nq, k);
    if (prefilter_scan_supported(h, nq, k)) return prefilter_scan_workspace(h, k);
    return (size_t)((k > 64) ? 1 : 8) * scan_blocks(h, k) * 4 * k * sizeof(u64);
}

// exhaustive search of queries that are already padded to h->ld (and normalised if
// the metric asks for it).  qbuf may lie at the start of h->ws, in which case the
// caller has sized the workspace with flat_search_workspace() beforehand.
int flat_search_prepared(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask,
                         float* D, long long* I, hipStream_t s) {
    const size_t qbytes = ((size_t)nq * h->ld * sizeof(float) + 255) & ~(size_t)255;
    if (int rc = h->ws.ensure(qbytes + flat_search_workspace(h, nq, k))) return rc;
    if (use_direct(h, nq, k))
        return direct_topk_search(h, nq, qbuf, k, min_score, mask, D, I, reinterpret_cast<float*>(static_cast<char*>(h->ws.ptr) + qbytes), s);
    if (gemm_scores_supported(h, nq, k)) return gemm_scores_search(h, nq, qbuf, k, min_score, mask, D, I, s);
    if (use_narrow(h, nq, k)) return narrow_topk_search(h, nq, qbuf, k, min_score, mask, D, I, s);
    if (gemm_topk_supported(h, nq, k)) return gemm_topk_search(h, nq, qbuf, k, min_score, mask, D, I, s);
    if (gemm_segments_supported(h, nq, k)) return gemm_segments_search(h, nq, qbuf, k, min_score, mask, D, I, s);
    if (prefilter_scan_supported(h, nq, k)) return prefilter_scan_search(h, qbuf, k, min_score, mask, D, I, s);
    return flat_scan_search(h, nq, qbuf, k, min_score, mask, D, I, s);
}

}  // namespace rag

using namespace rag;

extern "C" {

int rag_version(void) { return RAG_ABI_VERSION; }
const char* rag_last_error(void) { return g_err.c_str(); }

int rag_thread_release(void) {
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    for (int dev = 0; dev < 64; ++dev) {
        bool any = false;
        for (int w = 0; w < TWS_KINDS; ++w) any = any || g_thread_ws[w][dev].ptr != nullptr;
        if (!any) continue;
        if (hipSetDevice(dev) != hipSuccess) continue;
        for (int w = 0; w < TWS_KINDS; ++w) g_thread_ws[w][dev].release();
    }
    g_thread_pinned.release();
    if (prev >= 0) (void)hipSetDevice(prev);
    return RAG_OK;
}

int rag_device_count(int* count) {
    if (!count) return fail(RAG_ERR_INVALID, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(RAG_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = c;
    return RAG_OK;
}

int rag_index_create(int dim, int metric, int kind, int device, rag_index_t** out) {
    if (!out) return fail(RAG_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (dim < 1 || dim > 65536) return fail(RAG_ERR_INVALID, "dim must be in [1, 65536]");
    if (metric != RAG_METRIC_IP && metric != RAG_METRIC_COSINE && metric != RAG_METRIC_L2) return fail(RAG_ERR_INVALID, "unknown metric");
    if (kind != RAG_KIND_FLAT && kind != RAG_KIND_IVF) return fail(RAG_ERR_INVALID, "unknown index kind");
    if (metric == RAG_METRIC_L2 && kind != RAG_KIND_FLAT) return fail(RAG_ERR_UNSUPPORTED, "the L2 metric needs a FLAT index");
    int ndev = 0;
    RAG_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(RAG_ERR_INVALID, "device " + std::to_string(device) + " out of range (" + std::to_string(ndev) + " visible)");
    DeviceGuard g(device);
    if (!g.ok) return fail(RAG_ERR_HIP, "hipSetDevice failed");
    rag_index* h = new rag_index();
    h->dim = dim;
    // L2 rows carry -|x|^2 / 2 in an extra column (the query a 1.0), so that the inner product of the padded
    // vectors ranks rows by distance; their stride is a multiple of 32 floats, which keeps the matrix-core path
    h->ld = metric == RAG_METRIC_L2 ? (dim + 1 + 31) & ~31 : (dim + 3) & ~3;
    h->metric = metric;
    h->kind = kind;
    h->device = device;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete h;
        return fail(RAG_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    *out = h;
    return RAG_OK;
}

int rag_index_free(rag_index_t* h) {
    if (!h) return RAG_OK;
    DeviceGuard g(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->X && !h->adopted) (void)hipFree(h->X);
    if (h->ids_dev) (void)hipFree(h->ids_dev);
    if (h->active) (void)hipFree(h->active);
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    if (h->ws_event) (void)hipEventDestroy(h->ws_event);
    if (h->tail_args_event) (void)hipEventDestroy(h->tail_args_event);
    if (h->tail_args_host) (void)hipHostFree(h->tail_args_host);
    if (h->Xb) (void)hipFree(h->Xb);
    if (h->xb_maxnorm) (void)hipFree(h->xb_maxnorm);
    ivf_free(h);
    h->ws.release();
    h->ws2.release();
    h->ws3.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return RAG_OK;
}

int rag_index_reserve(rag_index_t* h, int64_t nrows) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (nrows < 0) return fail(RAG_ERR_INVALID, "nrows < 0");
    RAG_LOCK_INDEX(h);
    DeviceGuard g(h->device);
    h->reserve_hint = nrows;
    if (nrows > h->cap && !h->adopted) return ensure_rows(h, nrows - h->n);
    return RAG_OK;
}

static int add_common(rag_index_t* h, int64_t n, const float* x, bool x_on_device, const int64_t* ids) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (n < 0) return fail(RAG_ERR_INVALID, "n < 0");
    if (n == 0) return RAG_OK;
    if (!x) return fail(RAG_ERR_INVALID, "x is NULL");
    if (h->n + n > 0x7FFFFFF0LL) return fail(RAG_ERR_UNSUPPORTED, "more than 2^31 rows per index");
    if (h->kind == RAG_KIND_IVF && h->ivf.nlist == 0) return fail(RAG_ERR_INVALID, "IVF index must be trained before rows are added");
    RAG_LOCK_INDEX(h);
    DeviceGuard g(h->device);
    if (int rc = ws_acquire(h, h->stream)) return rc;
    if (int rc = ensure_rows(h, n)) return rc;
    const int normalize = h->metric == RAG_METRIC_COSINE ? 1 : h->metric == RAG_METRIC_L2 ? 2 : 0;      // ingest mode
    float* dst = h->X + (size_t)h->n * h->ld;
    if (x_on_device) {
        if (!normalize && h->ld == h->dim) {
            RAG_HIP_CHECK(hipMemcpyAsync(dst, x, (size_t)n * h->dim * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        } else {
            RAG_HIP_CHECK(launch_ingest_rows(x, n, h->dim, h->ld, normalize, dst, h->stream));
        }
    } else if (!normalize && h->ld == h->dim) {
        RAG_HIP_CHECK(hipMemcpyAsync(dst, x, (size_t)n * h->dim * sizeof(float), hipMemcpyHostToDevice, h->stream));
    } else {
        // stage host rows through the workspace in chunks, then pad / normalise on device
        const long long chunk_rows = std::max<long long>(1, (64LL << 20) / ((long long)h->dim * 4));
        for (long long r0 = 0; r0 < n; r0 += chunk_rows) {
            const long long c = std::min<long long>(chunk_rows, n - r0);
            if (int rc = h->ws.ensure((size_t)c * h->dim * sizeof(float))) return rc;
            RAG_HIP_CHECK(hipMemcpyAsync(h->ws.ptr, x + (size_t)r0 * h->dim, (size_t)c * h->dim * sizeof(float), hipMemcpyHostToDevice, h->stream));
            RAG_HIP_CHECK(launch_ingest_rows(static_cast<const float*>(h->ws.ptr), c, h->dim, h->ld, normalize, dst + (size_t)r0 * h->ld, h->stream));
            RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
        }
    }
    RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (int rc = append_ids(h, n, ids)) return rc;
    const long long first = h->n;
    h->n += n;
    if (h->kind == RAG_KIND_IVF) {
        if (int rc = ivf_assign_new_rows(h, first, n)) {
            // rows without a list cannot stay: the add did not happen
            h->n = first;
            if (h->ivf.assign.size() > (size_t)first) h->ivf.assign.resize((size_t)first);
            return rc;
        }
    }
    return RAG_OK;
}

int rag_index_add(rag_index_t* h, int64_t n, const float* x, const int64_t* ids) { return add_common(h, n, x, false, ids); }
int rag_index_add_device(rag_index_t* h, int64_t n, const float* x_dev, const int64_t* ids) { return add_common(h, n, x_dev, true, ids); }

int rag_index_adopt_device(rag_index_t* h, int64_t n, float* x_dev) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (h->n != 0 || h->X) return fail(RAG_ERR_INVALID, "adopt needs an empty index");
    if (h->kind != RAG_KIND_FLAT) return fail(RAG_ERR_UNSUPPORTED, "adopt is for FLAT indexes");
    if (h->metric == RAG_METRIC_L2) return fail(RAG_ERR_UNSUPPORTED, "an L2 index stores an extra column per row: add the rows instead");
    if (n < 0 || (n > 0 && !x_dev)) return fail(RAG_ERR_INVALID, "bad buffer");
    if (h->dim % 4 != 0) return fail(RAG_ERR_INVALID, "adopt needs dim % 4 == 0");
    if ((reinterpret_cast<uintptr_t>(x_dev) & 15) != 0) return fail(RAG_ERR_INVALID, "adopt needs a 16-byte aligned buffer");
    if (n > 0x7FFFFFF0LL) return fail(RAG_ERR_UNSUPPORTED, "more than 2^31 rows per index");
    RAG_LOCK_INDEX(h);
    DeviceGuard g(h->device);
    if (h->metric == RAG_METRIC_COSINE && n > 0) {
        RAG_HIP_CHECK(launch_ingest_rows(x_dev, n, h->dim, h->ld, 1, x_dev, h->stream));
        RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    h->X = x_dev;
    h->n = n;
    h->cap = n;
    h->adopted = true;
    return RAG_OK;
}

int rag_index_profile_enable(rag_index_t* h, int enable) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    h->profiling = enable != 0;
    return RAG_OK;
}

int rag_index_profile_read(rag_index_t* h, double* kernel_ms, int64_t* launches, int reset) {
    if (!h || !kernel_ms || !launches) return fail(RAG_ERR_INVALID, "NULL argument");
    RAG_LOCK_INDEX(h);
    DeviceGuard g(h->device);
    double total = 0.0;
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        RAG_HIP_CHECK(hipEventSynchronize(h->prof_events[i + 1]));
        float ms = 0.f;
        RAG_HIP_CHECK(hipEventElapsedTime(&ms, h->prof_events[i], h->prof_events[i + 1]));
        total += ms;
    }
    *kernel_ms = total;
    *launches = (int64_t)(h->prof_used / 2);
    if (reset) h->prof_used = 0;
    return RAG_OK;
}

int rag_index_set_stream_lifetime(rag_index_t* h, int streams_outlive_searches) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    RAG_LOCK_INDEX(h);
    h->streams_outlive = streams_outlive_searches != 0;
    return RAG_OK;
}

int rag_index_set_prefilter(rag_index_t* h, int mode) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (mode != 0 && mode != 1) return fail(RAG_ERR_INVALID, "prefilter mode must be 0 (off) or 1 (bf16 first pass)");
    if (mode == 1 && h->kind != RAG_KIND_FLAT) return fail(RAG_ERR_UNSUPPORTED, "the bf16 first pass needs a FLAT index");
    if (mode == 1 && h->metric == RAG_METRIC_L2) return fail(RAG_ERR_UNSUPPORTED, "the bf16 first pass is not available with the L2 metric");
    RAG_LOCK_INDEX(h);
    DeviceGuard g(h->device);
    if (mode == 0 && h->Xb) {
        if (int rc = ws_acquire(h, h->stream)) return rc;
        RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
        RAG_HIP_CHECK(hipFree(h->Xb));
        h->Xb = nullptr;
        h->xb_rows = h->xb_cap = 0;
    }
    h->prefilter = mode;
    return RAG_OK;
}

int rag_index_set_id_base(rag_index_t* h, int64_t base) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    RAG_LOCK_INDEX(h);
    h->id_base = base;
    return RAG_OK;
}

int64_t rag_index_ntotal(const rag_index_t* h) { return h ? h->n : 0; }
int rag_index_dim(const rag_index_t* h) { return h ? h->dim : 0; }
int rag_index_metric(const rag_index_t* h) { return h ? h->metric : -1; }
int rag_index_kind(const rag_index_t* h) { return h ? h->kind : -1; }
const char* rag_index_last_path(const rag_index_t* h) { return h ? h->last_path : "none"; }

int rag_index_set_active(rag_index_t* h, int64_t n, const int64_t* rows, int active) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (n < 0 || (n > 0 && !rows)) return fail(RAG_ERR_INVALID, "bad rows");
    if (n == 0) return RAG_OK;
    RAG_LOCK_INDEX(h);
    for (int64_t i = 0; i < n; ++i)
        if (rows[i] < 0 || rows[i] >= h->n) return fail(RAG_ERR_INVALID, "row " + std::to_string(rows[i]) + " out of range");
    DeviceGuard g(h->device);
    if (!h->active) {
        if (active) return RAG_OK;   // everything is active already
        RAG_HIP_CHECK(hipMalloc((void**)&h->active, (size_t)std::max<long long>(h->cap, 1)));
        RAG_HIP_CHECK(hipMemset(h->active, 1, (size_t)std::max<long long>(h->cap, 1)));
    }
    if (int rc = ws_acquire(h, h->stream)) return rc;
    if (int rc = h->ws2.ensure((size_t)n * sizeof(long long))) return rc;
    RAG_HIP_CHECK(hipMemcpyAsync(h->ws2.ptr, rows, (size_t)n * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(scatter_bytes_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, h->active,
                       static_cast<const long long*>(h->ws2.ptr), (long long)n, h->n, (uint8_t)(active ? 1 : 0));
    RAG_HIP_CHECK(hipGetLastError());
    RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
    return RAG_OK;
}

int rag_index_reconstruct(rag_index_t* h, int64_t n, const int64_t* rows, float* out) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (n < 0 || (n > 0 && (!rows || !out))) return fail(RAG_ERR_INVALID, "bad arguments");
    if (n == 0) return RAG_OK;
    RAG_LOCK_INDEX(h);
    for (int64_t i = 0; i < n; ++i)
        if (rows[i] < 0 || rows[i] >= h->n) return fail(RAG_ERR_INVALID, "row " + std::to_string(rows[i]) + " out of range");
    DeviceGuard g(h->device);
    const size_t rbytes = ((size_t)n * sizeof(long long) + 255) & ~(size_t)255;
    if (int rc = ws_acquire(h, h->stream)) return rc;
    if (int rc = h->ws2.ensure(rbytes + (size_t)n * h->dim * sizeof(float))) return rc;
    long long* drows = static_cast<long long*>(h->ws2.ptr);
    float* dout = reinterpret_cast<float*>(static_cast<char*>(h->ws2.ptr) + rbytes);
    std::vector<long long> storage_rows;
    const void* src_rows = rows;
    if (h->kind == RAG_KIND_IVF) {
        if (int rc = ivf_reconstruct_rows(h, n, rows, storage_rows)) return rc;
        src_rows = storage_rows.data();
    }
    RAG_HIP_CHECK(hipMemcpyAsync(drows, src_rows, (size_t)n * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
    RAG_HIP_CHECK(launch_gather_rows(h->X, h->dim, h->ld, n, drows, h->n, dout, h->stream));
    RAG_HIP_CHECK(hipMemcpyAsync(out, dout, (size_t)n * h->dim * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
    return RAG_OK;
}

static int search_device_ordered(rag_index* h, int64_t nq, const float* q_dev, int k, float min_score, int nprobe,
                                 const uint8_t* row_mask_dev, float* D_dev, int64_t* I_dev, hipStream_t s);

int rag_index_search_device(rag_index_t* h, int64_t nq, const float* q_dev, int k, float min_score, int nprobe,
                            const uint8_t* row_mask_dev, float* D_dev, int64_t* I_dev, void* stream) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (nq < 0) return fail(RAG_ERR_INVALID, "nq < 0");
    if (k < 1 || k > 1024) return fail(RAG_ERR_INVALID, "k must be in [1, 1024]");
    if (nq == 0) return RAG_OK;
    if (!q_dev || !D_dev || !I_dev) return fail(RAG_ERR_INVALID, "NULL buffer");
    if (min_score != min_score) return fail(RAG_ERR_INVALID, "min_score is NaN");
    RAG_LOCK_INDEX(h);
    DeviceGuard g(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = ws_acquire(h, s)) return rc;
    const int rc = search_device_ordered(h, nq, q_dev, k, min_score, nprobe, row_mask_dev, D_dev, I_dev, s);
    if (rc) return rc;
    return ws_release(h, s);
}

static int search_device_ordered(rag_index* h, int64_t nq, const float* q_dev, int k, float min_score, int nprobe,
                                 const uint8_t* row_mask_dev, float* D_dev, int64_t* I_dev, hipStream_t s) {
    long long* I = reinterpret_cast<long long*>(I_dev);
    if (h->n == 0) {
        hipLaunchKernelGGL(fill_results_kernel, dim3(grid_for(nq * k)), dim3(256), 0, s, D_dev, I, (long long)nq * k);
        RAG_HIP_CHECK(hipGetLastError());
        if (h->metric == RAG_METRIC_L2)      // the same empty slots in the convention of distances: (-1, FLT_MAX)
            RAG_HIP_CHECK(launch_l2_finish(q_dev, h->X, nq, 0, 0, k, 0, 0.f, nullptr, 0, D_dev, I, s));
        h->last_path = "empty";
        return RAG_OK;
    }
    // effective row mask = soft-delete mask AND caller's filter mask
    const uint8_t* mask = row_mask_dev;
    if (h->active && h->kind == RAG_KIND_IVF) {
        // the IVF kernels take both masks and read them where a row is tested
    } else if (h->active) {
        if (row_mask_dev) {
            if (int rc = h->ws2.ensure((size_t)h->n)) return rc;
            hipLaunchKernelGGL(and_masks_kernel, dim3(grid_for(h->n)), dim3(256), 0, s, h->active, row_mask_dev, h->n,
                               static_cast<uint8_t*>(h->ws2.ptr));
            RAG_HIP_CHECK(hipGetLastError());
            mask = static_cast<const uint8_t*>(h->ws2.ptr);
        } else {
            mask = h->active;
        }
    }
    // workspace = [padded queries | what the chosen kernel needs]; sized before anything is
    // enqueued, because growing it frees the old allocation
    const size_t qbytes = ((size_t)nq * h->ld * sizeof(float) + 255) & ~(size_t)255;
    const size_t extra = h->kind == RAG_KIND_FLAT ? flat_search_workspace(h, nq, k) : ivf_search_workspace(h, nq, k, nprobe);
    if (int rc = h->ws.ensure(qbytes + extra)) return rc;
    // padded (and for COSINE normalised) queries at the start of the workspace; inner-product
    // queries that need neither are used where they are (one launch less per search)
    const float* qbuf = static_cast<const float*>(h->ws.ptr);
    if (h->metric == RAG_METRIC_IP && h->ld == h->dim && (reinterpret_cast<uintptr_t>(q_dev) & 15) == 0)
        qbuf = q_dev;
    else
        RAG_HIP_CHECK(launch_prep_queries(q_dev, nq, h->dim, h->ld, h->metric == RAG_METRIC_COSINE ? 1 : h->metric == RAG_METRIC_L2 ? 2 : 0,
                                          static_cast<float*>(h->ws.ptr), s));
    if (h->kind == RAG_KIND_IVF)
        return ivf_search_device(h, nq, qbuf, k, min_score, nprobe, h->active, row_mask_dev, D_dev, I, s);
    if (h->metric == RAG_METRIC_L2) {
        // the k nearest rows are the k best under q.x - |x|^2 / 2.  That search leaves row positions (the id table and
        // base are set aside for it; the index is locked and they are read when a launch is built, not later); the
        // finishing kernel computes the distances themselves, applies the limit -- a distance -- and names the ids
        long long* const ids = h->ids_dev;
        const long long base = h->id_base;
        h->ids_dev = nullptr;
        h->id_base = 0;
        const int rc = flat_search_prepared(h, nq, qbuf, k, -FLT_MAX, mask, D_dev, I, s);
        h->ids_dev = ids;
        h->id_base = base;
        if (rc) return rc;
        RAG_HIP_CHECK(launch_l2_finish(qbuf, h->X, nq, h->dim, h->ld, k, min_score >= 0.f, min_score, ids, base, D_dev, I, s));
        return RAG_OK;
    }
    return flat_search_prepared(h, nq, qbuf, k, min_score, mask, D_dev, I, s);
}

}  // extern "C"

namespace rag {
void launch_fill_results(float* D, long long* I, long long n, hipStream_t s) {
    hipLaunchKernelGGL(fill_results_kernel, dim3(grid_for(n)), dim3(256), 0, s, D, I, n);
}

// Forget the rows at and behind position n -- the way a caller that spreads one add over several indexes takes back
// the parts that went through when another part failed.  Only rows that no search has seen: an IVF index must not
// have packed them into its lists yet (they still wait at the tail in arrival order).
int index_truncate(rag_index* h, long long n) {
    RAG_LOCK_INDEX(h);
    if (n < 0 || n > h->n) return fail(RAG_ERR_INVALID, "truncate: row count out of range");
    if (n == h->n) return RAG_OK;
    if (h->adopted) return fail(RAG_ERR_INVALID, "truncate: the rows are in an adopted buffer");
    if (h->kind == RAG_KIND_IVF) {
        if (h->ivf.n_packed > n) return fail(RAG_ERR_INVALID, "truncate: rows are already packed into their lists");
        if (h->ivf.assign.size() > (size_t)n) h->ivf.assign.resize((size_t)n);
    }
    DeviceGuard g(h->device);
    if (int rc = ws_acquire(h, h->stream)) return rc;
    RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->active) RAG_HIP_CHECK(hipMemset(h->active + n, 1, (size_t)(h->n - n)));       // a later add starts with active rows
    if (h->xb_rows > n) h->xb_rows = n;
    h->n = n;
    return RAG_OK;
}

// a search enqueued for another entry point of the library (index locked, device current, workspaces acquired)
int search_device_enqueue(rag_index* h, int64_t nq, const float* q_dev, int k, float min_score, int nprobe,
                          const uint8_t* row_mask_dev, float* D_dev, int64_t* I_dev, hipStream_t s) {
    return search_device_ordered(h, nq, q_dev, k, min_score, nprobe, row_mask_dev, D_dev, I_dev, s);
}
}  // namespace rag

extern "C" {

// A caller filter kept in device memory (rag_filter_create): one byte per row of the index it was made for.
struct rag_filter {
    int device;
    long long n;
    uint8_t* dev;
};

// the filter's bytes for a search on h, or an error when it belongs to another device or another row count
static int filter_bytes(const rag_index* h, const rag_filter* f, const uint8_t** out) {
    *out = nullptr;
    if (!f) return RAG_OK;
    if (f->device != h->device) return fail(RAG_ERR_INVALID, "the filter lives on another device than the index");
    if (f->n != h->n) return fail(RAG_ERR_INVALID, "the filter was made for " + std::to_string(f->n) + " rows, the index holds " + std::to_string(h->n));
    *out = f->dev;
    return RAG_OK;
}

static int search_host(rag_index_t* h, int64_t nq, const float* q, int k, float min_score, int nprobe,
                       const uint8_t* row_mask, const rag_filter* filter, float* D, int64_t* I) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (nq < 0) return fail(RAG_ERR_INVALID, "nq < 0");
    if (k < 1 || k > 1024) return fail(RAG_ERR_INVALID, "k must be in [1, 1024]");
    if (nq == 0) return RAG_OK;
    if (!q || !D || !I) return fail(RAG_ERR_INVALID, "NULL buffer");
    RAG_LOCK_INDEX(h);
    const uint8_t* f_dev = nullptr;
    if (int rc = filter_bytes(h, filter, &f_dev)) return rc;
    DeviceGuard g(h->device);
    // staging in a separate allocation from the search workspaces
    Workspace& st = thread_workspace(TWS_STAGE, h->device);
    const size_t qb = ((size_t)nq * h->dim * sizeof(float) + 255) & ~(size_t)255;
    const size_t db = ((size_t)nq * k * sizeof(float) + 255) & ~(size_t)255;
    const size_t ib = ((size_t)nq * k * sizeof(long long) + 255) & ~(size_t)255;
    const size_t mb = row_mask ? (((size_t)h->n + 255) & ~(size_t)255) : 0;
    if (int rc = st.ensure(qb + db + ib + mb)) return rc;
    char* base = static_cast<char*>(st.ptr);
    float* q_dev = reinterpret_cast<float*>(base);
    float* D_dev = reinterpret_cast<float*>(base + qb);
    int64_t* I_dev = reinterpret_cast<int64_t*>(base + qb + db);
    const uint8_t* m_dev = row_mask ? reinterpret_cast<uint8_t*>(base + qb + db + ib) : f_dev;
    const size_t q_bytes = (size_t)nq * h->dim * sizeof(float);
    // small batches travel through the thread's page-locked block: the queries in one copy, scores and ids (adjacent
    // in the staging block) back in one
    const bool pinned = qb + db + ib <= PINNED_STAGE_MAX && g_thread_pinned.ensure(qb + db + ib) == RAG_OK;
    char* pin = static_cast<char*>(g_thread_pinned.ptr);
    static const int copy_kernels = env_int("RAG_STAGE_COPY_KERNELS", 1);      // measurement knob: 0 = DMA copies on the pinned block
    const bool by_kernel = pinned && copy_kernels && qb + db + ib <= (256u << 10);
    if (pinned) {
        memcpy(pin, q, q_bytes);
        if (by_kernel) {
            const long long words = (long long)(q_bytes / 4);
            hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)std::min<long long>(64, (words + 255) / 256)), dim3(256), 0, h->stream,
                               reinterpret_cast<const uint32_t*>(pin), reinterpret_cast<uint32_t*>(q_dev), words);
            RAG_HIP_CHECK(hipGetLastError());
        } else {
            RAG_HIP_CHECK(hipMemcpyAsync(q_dev, pin, q_bytes, hipMemcpyHostToDevice, h->stream));
        }
    } else {
        RAG_HIP_CHECK(hipMemcpyAsync(q_dev, q, q_bytes, hipMemcpyHostToDevice, h->stream));
    }
    if (row_mask && h->n > 0)
        RAG_HIP_CHECK(hipMemcpyAsync(base + qb + db + ib, row_mask, (size_t)h->n, hipMemcpyHostToDevice, h->stream));
    if (int rc = rag_index_search_device(h, nq, q_dev, k, min_score, nprobe, m_dev, D_dev, I_dev, h->stream)) return rc;
    if (pinned) {
        const size_t out_bytes = db + (size_t)nq * k * sizeof(long long);
        if (by_kernel) {
            const long long words = (long long)(out_bytes / 4);
            hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)std::min<long long>(64, (words + 255) / 256)), dim3(256), 0, h->stream,
                               reinterpret_cast<const uint32_t*>(D_dev), reinterpret_cast<uint32_t*>(pin + qb), words);
            RAG_HIP_CHECK(hipGetLastError());
        } else {
            RAG_HIP_CHECK(hipMemcpyAsync(pin + qb, D_dev, out_bytes, hipMemcpyDeviceToHost, h->stream));
        }
        RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
        memcpy(D, pin + qb, (size_t)nq * k * sizeof(float));
        memcpy(I, pin + qb + db, (size_t)nq * k * sizeof(long long));
        return RAG_OK;
    }
    RAG_HIP_CHECK(hipMemcpyAsync(D, D_dev, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    RAG_HIP_CHECK(hipMemcpyAsync(I, I_dev, (size_t)nq * k * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    RAG_HIP_CHECK(hipStreamSynchronize(h->stream));
    return RAG_OK;
}

int rag_index_search(rag_index_t* h, int64_t nq, const float* q, int k, float min_score, int nprobe,
                     const uint8_t* row_mask, float* D, int64_t* I) {
    return search_host(h, nq, q, k, min_score, nprobe, row_mask, nullptr, D, I);
}

int rag_index_search_filtered(rag_index_t* h, int64_t nq, const float* q, int k, float min_score, int nprobe,
                              const rag_filter_t* filter, float* D, int64_t* I) {
    return search_host(h, nq, q, k, min_score, nprobe, nullptr, filter, D, I);
}

int rag_filter_create(int device, int64_t n, const uint8_t* row_mask, rag_filter_t** out) {
    if (!out) return fail(RAG_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n < 0 || (n > 0 && !row_mask)) return fail(RAG_ERR_INVALID, "filter: n < 0 or NULL bytes");
    int ndev = 0;
    RAG_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(RAG_ERR_INVALID, "no such device");
    DeviceGuard g(device);
    uint8_t* dev = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&dev), (size_t)std::max<int64_t>(n, 1)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RAG_ERR_OOM, "out of device memory for a filter of " + std::to_string(n) + " rows");
    }
    // a blocking copy: the bytes are in place for any stream when this returns
    if (n > 0) {
        const hipError_t e = hipMemcpy(dev, row_mask, (size_t)n, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(dev);
            return fail(RAG_ERR_HIP, std::string("filter upload: ") + hipGetErrorString(e));
        }
    }
    rag_filter* f = new (std::nothrow) rag_filter{device, (long long)n, dev};
    if (!f) {
        (void)hipFree(dev);
        return fail(RAG_ERR_OOM, "out of host memory");
    }
    *out = f;
    return RAG_OK;
}

int rag_filter_free(rag_filter_t* f) {
    if (!f) return RAG_OK;
    DeviceGuard g(f->device);
    const hipError_t e = hipFree(f->dev);        // waits for the searches that still read it
    delete f;
    if (e != hipSuccess) return fail(RAG_ERR_HIP, std::string("filter free: ") + hipGetErrorString(e));
    return RAG_OK;
}

int64_t rag_filter_rows(const rag_filter_t* f) { return f ? f->n : -1; }

const uint8_t* rag_filter_device_bytes(const rag_filter_t* f) { return f ? f->dev : nullptr; }

// device part of a segment search: the k best of every (query, segment) pair left in h->ws2 as
// D_dev / I_dev [nq][nseg][k] (and the row positions in R_dev when asked for); nothing synchronised.
// `tail` extra bytes are reserved behind them in ws2 for the caller (*tail_dev).
static int segments_check(rag_index_t* h, int64_t nq, const float* q, int64_t nseg, const int64_t* seg_begin,
                          const int64_t* seg_end, int k, long long* rows_total) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (h->kind != RAG_KIND_FLAT) return fail(RAG_ERR_UNSUPPORTED, "segment search needs a FLAT index");
    if (h->metric == RAG_METRIC_L2) return fail(RAG_ERR_UNSUPPORTED, "segment search ranks by similarity: not available with the L2 metric");
    if (nq < 0 || nseg < 0) return fail(RAG_ERR_INVALID, "nq / nseg < 0");
    if (k < 1 || k > 1024) return fail(RAG_ERR_INVALID, "k must be in [1, 1024]");
    if (nq == 0 || nseg == 0) return RAG_OK;
    if (!q || !seg_begin || !seg_end) return fail(RAG_ERR_INVALID, "NULL buffer");
    if (nq > 65535) return fail(RAG_ERR_INVALID, "at most 65535 queries per segment search");
    *rows_total = 0;
    for (int64_t i = 0; i < nseg; ++i) {
        if (seg_begin[i] < 0 || seg_end[i] < seg_begin[i] || seg_end[i] > h->n)
            return fail(RAG_ERR_INVALID, "segment " + std::to_string(i) + " is not a row range of the index");
        *rows_total += seg_end[i] - seg_begin[i];
    }
    return RAG_OK;
}

static int segments_enqueue(rag_index_t* h, int64_t nq, const float* q, int64_t nseg, const int64_t* seg_begin,
                            const int64_t* seg_end, long long rows_total, int k, float min_score, const uint8_t* row_mask,
                            const uint8_t* row_mask_dev, bool want_rows, size_t tail, float** D_out, long long** I_out, long long** R_out, char** tail_dev,
                            hipStream_t s, const double* weights = nullptr, const double** weights_dev = nullptr) {
    // effective mask
    const uint8_t* mask = nullptr;
    const size_t mb = (((size_t)h->n + 255) & ~(size_t)255);
    const size_t sb = ((size_t)nseg * sizeof(long long) + 255) & ~(size_t)255;
    const size_t db = ((size_t)nq * nseg * k * sizeof(float) + 255) & ~(size_t)255;
    const size_t ib = ((size_t)nq * nseg * k * sizeof(long long) + 255) & ~(size_t)255;
    const size_t rb = want_rows ? ib : 0;
    const size_t qhb = ((size_t)nq * h->dim * sizeof(float) + 255) & ~(size_t)255;
    const size_t wb = weights ? (((size_t)nq * sizeof(double) + 255) & ~(size_t)255) : 0;
    if (int rc = h->ws2.ensure(2 * mb + 2 * sb + qhb + wb + db + ib + rb + tail)) return rc;
    char* base = static_cast<char*>(h->ws2.ptr);
    uint8_t* m_user = reinterpret_cast<uint8_t*>(base);
    uint8_t* m_comb = reinterpret_cast<uint8_t*>(base + mb);
    // what the caller brings -- segment bounds, queries, weights -- sits together: one copy from the thread's page-locked block
    char* in_dev = base + 2 * mb;
    long long* d_sb = reinterpret_cast<long long*>(in_dev);
    long long* d_se = reinterpret_cast<long long*>(in_dev + sb);
    float* q_dev = reinterpret_cast<float*>(in_dev + 2 * sb);
    double* w_dev = weights ? reinterpret_cast<double*>(in_dev + 2 * sb + qhb) : nullptr;
    const size_t in_bytes = 2 * sb + qhb + wb;
    char* out_base = in_dev + in_bytes;
    float* D_dev = reinterpret_cast<float*>(out_base);
    long long* I_dev = reinterpret_cast<long long*>(out_base + db);
    long long* R_dev = want_rows ? reinterpret_cast<long long*>(out_base + db + ib) : nullptr;
    if (tail_dev) *tail_dev = out_base + db + ib + rb;
    if (weights_dev) *weights_dev = w_dev;
    if (row_mask) {
        RAG_HIP_CHECK(hipMemcpyAsync(m_user, row_mask, (size_t)h->n, hipMemcpyHostToDevice, s));
        mask = m_user;
    } else if (row_mask_dev) {
        mask = row_mask_dev;
    }
    if (h->active) {
        if (mask) {
            hipLaunchKernelGGL(and_masks_kernel, dim3(grid_for(h->n)), dim3(256), 0, s, h->active, mask, h->n, m_comb);
            RAG_HIP_CHECK(hipGetLastError());
            mask = m_comb;
        } else {
            mask = h->active;
        }
    }
    if (in_bytes <= PINNED_STAGE_MAX && g_thread_pinned.ensure(in_bytes) == RAG_OK) {
        char* pin = static_cast<char*>(g_thread_pinned.ptr);
        memcpy(pin, seg_begin, (size_t)nseg * sizeof(long long));
        memcpy(pin + sb, seg_end, (size_t)nseg * sizeof(long long));
        memcpy(pin + 2 * sb, q, (size_t)nq * h->dim * sizeof(float));
        if (weights) memcpy(pin + 2 * sb + qhb, weights, (size_t)nq * sizeof(double));
        RAG_HIP_CHECK(hipMemcpyAsync(in_dev, pin, in_bytes, hipMemcpyHostToDevice, s));
    } else {
        RAG_HIP_CHECK(hipMemcpyAsync(d_sb, seg_begin, (size_t)nseg * sizeof(long long), hipMemcpyHostToDevice, s));
        RAG_HIP_CHECK(hipMemcpyAsync(d_se, seg_end, (size_t)nseg * sizeof(long long), hipMemcpyHostToDevice, s));
        RAG_HIP_CHECK(hipMemcpyAsync(q_dev, q, (size_t)nq * h->dim * sizeof(float), hipMemcpyHostToDevice, s));
        if (weights) RAG_HIP_CHECK(hipMemcpyAsync(w_dev, weights, (size_t)nq * sizeof(double), hipMemcpyHostToDevice, s));
    }
    // slices per segment: a slice keeps >= 8 tiles of an average segment (100 documents x 3 queries of ~225 rows x 384:
    // scan 30.8 us with one slice per pair, 20.7 us with three; 18.9 us at 4 tiles per slice, where the merge starts to grow)
    const int tile = scan_rows_per_tile(h->ld / 4);
    const long long avg_tiles = std::max<long long>(1, (rows_total / nseg) / tile);
    long long slices = 2048 / std::max<long long>(1, nq * nseg);
    static const int slice_tiles = std::max(1, env_int("RAG_SEG_SLICE_TILES", 8));        // measurement knob
    slices = std::min<long long>(slices, std::max<long long>(1, avg_tiles / slice_tiles));
    if (slices < 1) slices = 1;
    const long long gx = nseg * slices;
    if (gx > 0x7FFFFFFFLL) return fail(RAG_ERR_INVALID, "too many segments");
    const int waves = (int)slices * 4;                      // per (query, segment)
    const size_t qbytes = ((size_t)nq * h->ld * sizeof(float) + 255) & ~(size_t)255;
    const size_t pbytes = (size_t)nq * nseg * waves * k * sizeof(u64);
    if (int rc = h->ws.ensure(qbytes + pbytes)) return rc;
    float* qbuf = static_cast<float*>(h->ws.ptr);
    u64* part = reinterpret_cast<u64*>(static_cast<char*>(h->ws.ptr) + qbytes);
    RAG_HIP_CHECK(launch_prep_queries(q_dev, nq, h->dim, h->ld, h->metric == RAG_METRIC_COSINE, qbuf, s));
    ScanArgs a = {};
    a.X = h->X;
    a.Q = qbuf;
    a.mask = mask;
    a.part = part;
    a.n = h->n;
    a.d4 = h->ld / 4;
    a.k = k;
    a.nt = 0;
    a.seg_begin = d_sb;
    a.seg_end = d_se;
    a.ivf_slices = (int)slices;
    a.grid_y = (int)nq;
    if (int rc = profile_begin(h, s)) return rc;
    RAG_HIP_CHECK(launch_flat_scan(a, 1, (int)gx, s));
    if (int rc = profile_end(h, s)) return rc;
    MergeArgs m = {};
    m.keys = part;
    m.per_query = (long long)slices * scan_lists_per_block(k) * k;          // lists of one (query, segment)
    m.k = k;
    m.cap = merge_cap(m.per_query, k);
    m.min_score = min_score;
    m.row_ids = h->ids_dev;
    m.id_base = h->id_base;
    m.rows_out = want_rows ? R_dev : h->merge_rows_out;
    m.D = D_dev;
    m.I = I_dev;
    RAG_HIP_CHECK(launch_merge_keys(m, nq * nseg, s));
    *D_out = D_dev;
    *I_out = I_dev;
    if (R_out) *R_out = R_dev;
    return RAG_OK;
}

static int search_segments_host(rag_index_t* h, int64_t nq, const float* q, int64_t nseg, const int64_t* seg_begin,
                                const int64_t* seg_end, int k, float min_score, const uint8_t* row_mask,
                                const rag_filter* filter, float* D, int64_t* I) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    RAG_LOCK_INDEX(h);
    long long rows_total = 0;
    if (int rc = segments_check(h, nq, q, nseg, seg_begin, seg_end, k, &rows_total)) return rc;
    const uint8_t* f_dev = nullptr;
    if (int rc = filter_bytes(h, filter, &f_dev)) return rc;
    if (nq == 0 || nseg == 0) return RAG_OK;
    if (!D || !I) return fail(RAG_ERR_INVALID, "NULL buffer");
    DeviceGuard g(h->device);
    hipStream_t s = h->stream;
    if (int rc = ws_acquire(h, s)) return rc;
    float* D_dev = nullptr;
    long long* I_dev = nullptr;
    if (int rc = segments_enqueue(h, nq, q, nseg, seg_begin, seg_end, rows_total, k, min_score, row_mask, f_dev, false, 0, &D_dev,
                                  &I_dev, nullptr, nullptr, s))
        return rc;
    const size_t d_bytes = (size_t)nq * nseg * k * sizeof(float), i_bytes = (size_t)nq * nseg * k * sizeof(long long);
    const size_t i_off = reinterpret_cast<const char*>(I_dev) - reinterpret_cast<const char*>(D_dev);     // scores, padding, ids
    if (i_off + i_bytes <= PINNED_STAGE_MAX && pinned_for_results(i_off + i_bytes, s) == RAG_OK) {
        char* pin = static_cast<char*>(g_thread_pinned.ptr);
        RAG_HIP_CHECK(hipMemcpyAsync(pin, D_dev, i_off + i_bytes, hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipStreamSynchronize(s));
        memcpy(D, pin, d_bytes);
        memcpy(I, pin + i_off, i_bytes);
    } else {
        RAG_HIP_CHECK(hipMemcpyAsync(D, D_dev, d_bytes, hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipMemcpyAsync(I, I_dev, i_bytes, hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipStreamSynchronize(s));
    }
    h->ws_pending = false;
    h->last_path = "segments";
    return RAG_OK;
}

int rag_index_search_segments(rag_index_t* h, int64_t nq, const float* q, int64_t nseg, const int64_t* seg_begin,
                              const int64_t* seg_end, int k, float min_score, const uint8_t* row_mask, float* D,
                              int64_t* I) {
    return search_segments_host(h, nq, q, nseg, seg_begin, seg_end, k, min_score, row_mask, nullptr, D, I);
}

int rag_index_search_segments_filtered(rag_index_t* h, int64_t nq, const float* q, int64_t nseg, const int64_t* seg_begin,
                                       const int64_t* seg_end, int k, float min_score, const rag_filter_t* filter, float* D,
                                       int64_t* I) {
    return search_segments_host(h, nq, q, nseg, seg_begin, seg_end, k, min_score, nullptr, filter, D, I);
}

// vector_db/search_manager.py:877-932 on the device: see include/rag_hip.h
static int search_segments_fused_host(rag_index_t* h, int64_t nq, const float* q, const double* weights, int64_t nseg,
                                      const int64_t* seg_begin, const int64_t* seg_end, int k, float min_score,
                                      const uint8_t* row_mask, const rag_filter* filter, int limit, int64_t* out_rows,
                                      double* out_scores, int32_t* out_strategy, int32_t* n_out) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    RAG_LOCK_INDEX(h);
    long long rows_total = 0;
    if (int rc = segments_check(h, nq, q, nseg, seg_begin, seg_end, k, &rows_total)) return rc;
    const uint8_t* f_dev = nullptr;
    if (int rc = filter_bytes(h, filter, &f_dev)) return rc;
    if (!n_out) return fail(RAG_ERR_INVALID, "n_out is NULL");
    *n_out = 0;
    if (limit < 1 || limit > 1024) return fail(RAG_ERR_INVALID, "limit must be in [1, 1024]");
    if (nq == 0 || nseg == 0) return RAG_OK;
    if (!weights || !out_rows || !out_scores || !out_strategy) return fail(RAG_ERR_INVALID, "NULL buffer");
    if (nq > 16 || (long long)nq * k > 1024) return fail(RAG_ERR_INVALID, "at most 16 query vectors and nq * k <= 1024");
    const size_t tail = fanout_workspace(nq, nseg, k, limit);
    if (tail == 0) return fail(RAG_ERR_UNSUPPORTED, "too many (document, candidate) pairs for the fused ranking");
    DeviceGuard g(h->device);
    hipStream_t s = h->stream;
    if (int rc = ws_acquire(h, s)) return rc;
    float* D_dev = nullptr;
    long long* I_dev = nullptr;
    long long* R_dev = nullptr;
    char* tail_dev = nullptr;
    const double* weights_dev = nullptr;
    if (int rc = segments_enqueue(h, nq, q, nseg, seg_begin, seg_end, rows_total, k, min_score, row_mask, f_dev, true, tail, &D_dev,
                                  &I_dev, &R_dev, &tail_dev, s, weights, &weights_dev))
        return rc;
    long long* rows_dev = nullptr;
    double* scores_dev = nullptr;
    int* strat_dev = nullptr;
    int* n_dev = nullptr;
    if (int rc = fanout_fuse_enqueue(D_dev, R_dev, weights, weights_dev, (int)nq, nseg, k, limit, tail_dev, &rows_dev, &scores_dev,
                                     &strat_dev, &n_dev, s))
        return rc;
    int32_t n = 0;
    // the four results lie in one stretch of the workspace (rows, scores, strategies, count): one copy to the page-locked block
    const char* lo = reinterpret_cast<const char*>(rows_dev);
    const char* hi = reinterpret_cast<const char*>(n_dev) + sizeof(int);
    const bool one_stretch = lo < reinterpret_cast<const char*>(scores_dev) && reinterpret_cast<const char*>(scores_dev) < reinterpret_cast<const char*>(strat_dev) &&
                             reinterpret_cast<const char*>(strat_dev) < reinterpret_cast<const char*>(n_dev) && (size_t)(hi - lo) <= PINNED_STAGE_MAX;
    if (one_stretch && pinned_for_results((size_t)(hi - lo), s) == RAG_OK) {
        char* pin = static_cast<char*>(g_thread_pinned.ptr);
        RAG_HIP_CHECK(hipMemcpyAsync(pin, lo, (size_t)(hi - lo), hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipStreamSynchronize(s));
        memcpy(&n, pin + (reinterpret_cast<const char*>(n_dev) - lo), sizeof(int));
        memcpy(out_rows, pin, (size_t)limit * sizeof(long long));
        memcpy(out_scores, pin + (reinterpret_cast<const char*>(scores_dev) - lo), (size_t)limit * sizeof(double));
        memcpy(out_strategy, pin + (reinterpret_cast<const char*>(strat_dev) - lo), (size_t)limit * sizeof(int));
    } else {
        RAG_HIP_CHECK(hipMemcpyAsync(&n, n_dev, sizeof(int), hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipMemcpyAsync(out_rows, rows_dev, (size_t)limit * sizeof(long long), hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipMemcpyAsync(out_scores, scores_dev, (size_t)limit * sizeof(double), hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipMemcpyAsync(out_strategy, strat_dev, (size_t)limit * sizeof(int), hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipStreamSynchronize(s));
    }
    h->ws_pending = false;
    h->last_path = "segments-fused";
    if (n < 0 || n > limit) return fail(RAG_ERR_HIP, "fan-out ranking returned a bad count");
    *n_out = n;
    return RAG_OK;
}

int rag_index_search_segments_fused(rag_index_t* h, int64_t nq, const float* q, const double* weights, int64_t nseg,
                                    const int64_t* seg_begin, const int64_t* seg_end, int k, float min_score,
                                    const uint8_t* row_mask, int limit, int64_t* out_rows, double* out_scores,
                                    int32_t* out_strategy, int32_t* n_out) {
    return search_segments_fused_host(h, nq, q, weights, nseg, seg_begin, seg_end, k, min_score, row_mask, nullptr, limit, out_rows,
                                      out_scores, out_strategy, n_out);
}

int rag_index_search_segments_fused_filtered(rag_index_t* h, int64_t nq, const float* q, const double* weights, int64_t nseg,
                                             const int64_t* seg_begin, const int64_t* seg_end, int k, float min_score,
                                             const rag_filter_t* filter, int limit, int64_t* out_rows, double* out_scores,
                                             int32_t* out_strategy, int32_t* n_out) {
    return search_segments_fused_host(h, nq, q, weights, nseg, seg_begin, seg_end, k, min_score, nullptr, filter, limit, out_rows,
                                      out_scores, out_strategy, n_out);
}

// shared by the two merge entry points: argument checks, lists -> keys in `keys`, merge
static int merge_topk_with_keys(int nlists, int64_t nq, int k, const float* D_dev, int64_t d_list_stride, const int64_t* I_dev,
                                int64_t i_list_stride, float* D_out_dev, int64_t* I_out_dev, u64* keys, hipStream_t s) {
    RAG_HIP_CHECK(launch_lists_to_keys(nlists, nq, k, D_dev, d_list_stride, reinterpret_cast<const long long*>(I_dev), i_list_stride, keys, s));
    MergeArgs m = {};
    m.keys = keys;
    m.per_query = (long long)nlists * k;
    m.k = k;
    m.cap = merge_cap(m.per_query, k);
    m.min_score = -FLT_MAX;
    m.row_ids = nullptr;
    m.id_base = 0;
    m.slot_ids = reinterpret_cast<const long long*>(I_dev);
    m.slot_stride = i_list_stride;
    m.D = D_out_dev;
    m.I = reinterpret_cast<long long*>(I_out_dev);
    RAG_HIP_CHECK(launch_merge_keys(m, nq, s));
    return RAG_OK;
}

}  // extern "C"

namespace rag {
int merge_lists_device(int nlists, long long nq, int k, const float* D_dev, long long d_list_stride, const long long* I_dev,
                       long long i_list_stride, float* D_out, long long* I_out, u64* keys, hipStream_t s) {
    return merge_topk_with_keys(nlists, nq, k, D_dev, d_list_stride, reinterpret_cast<const int64_t*>(I_dev), i_list_stride, D_out,
                                reinterpret_cast<int64_t*>(I_out), keys, s);
}
}  // namespace rag

extern "C" {

static int merge_topk_check(int nlists, int64_t nq, int k, const float* D_dev, int64_t& d_list_stride, const int64_t* I_dev,
                            int64_t& i_list_stride, float* D_out_dev, int64_t* I_out_dev) {
    if (nlists < 1 || nq < 0 || k < 1 || k > 1024) return fail(RAG_ERR_INVALID, "bad nlists / nq / k");
    if (d_list_stride == 0) d_list_stride = nq * k;
    if (i_list_stride == 0) i_list_stride = nq * k;
    if (d_list_stride < nq * k || i_list_stride < nq * k) return fail(RAG_ERR_INVALID, "list stride smaller than nq*k");
    if (nq > 0 && (!D_dev || !I_dev || !D_out_dev || !I_out_dev)) return fail(RAG_ERR_INVALID, "NULL buffer");
    if ((long long)nlists * k > 0x7FFFFFFFLL) return fail(RAG_ERR_INVALID, "nlists * k too large");
    return RAG_OK;
}

int rag_merge_topk_device(int nlists, int64_t nq, int k, const float* D_dev, int64_t d_list_stride, const int64_t* I_dev,
                          int64_t i_list_stride, float* D_out_dev, int64_t* I_out_dev, int device, void* stream) {
    if (int rc = merge_topk_check(nlists, nq, k, D_dev, d_list_stride, I_dev, i_list_stride, D_out_dev, I_out_dev)) return rc;
    if (nq == 0) return RAG_OK;
    DeviceGuard g(device);
    if (!g.ok) return fail(RAG_ERR_HIP, "hipSetDevice failed");
    Workspace& ws = thread_workspace(TWS_MERGE, device);
    if (int rc = ws.ensure((size_t)nq * nlists * k * sizeof(u64))) return rc;
    return merge_topk_with_keys(nlists, nq, k, D_dev, d_list_stride, I_dev, i_list_stride, D_out_dev, I_out_dev,
                                static_cast<u64*>(ws.ptr), static_cast<hipStream_t>(stream));
}

}  // extern "C"

// ---- the merge without any allocation -------------------------------------------------------
// The candidate keys of rag_merge_topk_device_static live in this array of the code object
// instead of a hipMalloc'd workspace.  Every GPU the library runs on holds its own copy at its own
// address for the life of the process; host code reaches it only through the address
// hipGetSymbolAddress returns for the current device (the name below is a host-side stand-in).
static __device__ __align__(256) unsigned char scratch[RAG_MERGE_STATIC_MAX_KEYS * sizeof(unsigned long long)];

namespace rag {
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[64] = {};      // per device, filled on the first call on that device

// device address of `scratch` on `device` (which must be current)
static int scratch_address(int device, void** out) {
    if (device < 0 || device >= 64) return fail(RAG_ERR_INVALID, "device out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return fail(RAG_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return RAG_OK;
}
}  // namespace rag

extern "C" {

int rag_merge_topk_device_static(int nlists, int64_t nq, int k, const float* D_dev, int64_t d_list_stride, const int64_t* I_dev,
                                 int64_t i_list_stride, float* D_out_dev, int64_t* I_out_dev, int device, void* stream) {
    if (int rc = merge_topk_check(nlists, nq, k, D_dev, d_list_stride, I_dev, i_list_stride, D_out_dev, I_out_dev)) return rc;
    if (nq == 0) return RAG_OK;
    if (nq > RAG_MERGE_STATIC_MAX_KEYS || (long long)nq * nlists * k > RAG_MERGE_STATIC_MAX_KEYS)
        return fail(RAG_ERR_INVALID, "nq * nlists * k = " + std::to_string((long long)nq * nlists * k) + " keys exceed the static scratch of " +
                                         std::to_string((long long)RAG_MERGE_STATIC_MAX_KEYS) + " keys; use rag_merge_topk_device");
    DeviceGuard g(device);
    if (!g.ok) return fail(RAG_ERR_HIP, "hipSetDevice failed");
    void* keys = nullptr;
    if (int rc = scratch_address(device, &keys)) return rc;
    // every key slot [0, nq*nlists*k) is written by the conversion kernel before the merge reads
    // it, so nothing of the previous call's contents has to be cleared
    return merge_topk_with_keys(nlists, nq, k, D_dev, d_list_stride, I_dev, i_list_stride, D_out_dev, I_out_dev,
                                static_cast<u64*>(keys), static_cast<hipStream_t>(stream));
}

}  // extern "C"
