// The index object behind rag_index_t and the error plumbing of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/rag_hip.h"
#include "kernels.h"

struct rag_index;

namespace rag {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define RAG_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::rag::fail(_e == hipErrorOutOfMemory ? RAG_ERR_OOM : RAG_ERR_HIP,        \
                               std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

// Grow-only device scratch owned by an index (or by the library per device).
struct Workspace {
    void* ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);   // returns rag_status
    void release();
};

// Staging memory of the calling host thread, per purpose and device (host-pointer searches, the
// multi-GPU merge, fusion / MMR): grow-only like every workspace, handed back by rag_thread_release().
enum { TWS_STAGE = 0, TWS_MERGE = 1, TWS_FUSE = 2, TWS_KINDS = 3 };
Workspace& thread_workspace(int which, int device);

// Page-locked host memory of the calling thread (grow-only, at least `need` bytes; handed back by rag_thread_release()):
// the small transfers of host-pointer entry points go through it as one copy each way.  A call that still has a copy
// in flight from the block must drain its stream before asking for more.
int thread_pinned(size_t need, char** ptr);

// RAII-free helpers: record an event pair around a launch when profiling is on
int profile_begin(rag_index* h, hipStream_t s);
int profile_end(rag_index* h, hipStream_t s);

// The workspaces of an index are shared by every search on it.  Work enqueued on
// stream `s` first waits for the previous user of the workspaces when that user
// ran on another stream (ws_acquire), and marks its own end (ws_release), so
// back-to-back searches on different streams are ordered on the device.
int ws_acquire(rag_index* h, hipStream_t s);
int ws_release(rag_index* h, hipStream_t s);

// IVF bookkeeping.  Rows are stored grouped by list ("storage order"); ids, the
// soft-delete bytes and every caller-visible row number stay in insertion order.
struct IvfState {
    int nlist = 0;
    rag_index* quantizer = nullptr;       // FLAT inner-product index over the centroids
    std::vector<int> assign;              // insertion position -> list
    const int* preassigned = nullptr;     // while set (snapshot load): lists of the rows being added, by insertion position
    std::vector<long long> orig;          // storage position -> insertion position
    std::vector<long long> pos;           // insertion position -> storage position
    long long n_packed = 0;               // rows [0, n_packed) of X are in list order
    long long* list_begin_dev = nullptr;  // [nlist + 1]
    long long* orig_dev = nullptr;        // [n_packed]
    long long orig_cap = 0;
    long long* pos_dev = nullptr;         // [n_packed] insertion position -> storage position
    long long pos_cap = 0;
    Workspace ws;                         // coarse results, packed mask
};

}  // namespace rag

#define RAG_LOCK_INDEX(h) std::lock_guard<std::recursive_mutex> rag_index_lock_((h)->mu)

struct rag_index {
    int dim = 0;
    int ld = 0;          // row stride in floats, multiple of 4
    int metric = 0;
    int kind = 0;
    int device = 0;
    float* X = nullptr;
    long long n = 0;
    long long cap = 0;
    bool adopted = false;
    long long* ids_dev = nullptr;   // explicit ids (cap entries) or nullptr
    long long id_base = 0;
    uint8_t* active = nullptr;      // nullptr = every row active
    long long n_inactive = 0;
    long long reserve_hint = 0;
    // optional bf16 first pass of large batches: a bf16 copy of the rows and the largest row norm
    int prefilter = 0;
    uint16_t* Xb = nullptr;         // [xb_cap][ld] bf16
    long long xb_rows = 0;          // rows [0, xb_rows) of Xb are current
    long long xb_cap = 0;
    int* xb_maxnorm = nullptr;      // device: bit pattern of the largest row norm converted so far
    rag::Workspace ws;              // queries, partial keys, staging
    rag::Workspace ws2;             // combined masks, host-API result staging
    rag::Workspace ws3;             // search -> fusion -> MMR pipeline: candidate lists, fp64 rows, pair matrix
    long long* merge_rows_out = nullptr;   // while set, the final merge of a search also writes every hit's row position here
    hipStream_t stream = nullptr;   // used by the host-pointer entry points
    // pinned staging of the batched search -> fusion -> MMR call (its per-query kernel arguments on their way to the
    // device) and the event behind the last copy out of it
    void* tail_args_host = nullptr;
    size_t tail_args_cap = 0;
    hipEvent_t tail_args_event = nullptr;
    hipEvent_t ws_event = nullptr;  // end of the last work that used ws / ws2
    hipStream_t ws_stream = nullptr;
    bool ws_pending = false;
    bool ws_unmarked = false;       // the last search's end has no event yet: ws_stream is marked when another stream asks
    bool streams_outlive = false;   // the caller's promise that a search's stream is still there when the next search comes
    const char* last_path = "none";
    rag::IvfState ivf;
    // measurement: event pairs around the dominant kernel of each search
    // every entry point that takes the index holds this while it runs on the host (enqueues, copies,
    // bookkeeping), so host threads may share an index; device work of consecutive calls is ordered
    // by ws_acquire / ws_release
    std::recursive_mutex mu;
    bool profiling = false;
    std::vector<hipEvent_t> prof_events;   // start, stop, start, stop, ...
    size_t prof_used = 0;
};
