// Kernel argument blocks and host-side launchers shared by the translation units
// of librag_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rag {

typedef unsigned long long u64;

struct ScanArgs {
    const float* X;        // corpus rows, row stride ld = 4*d4 floats, 16-byte aligned
    const float* Q;        // nq padded queries, stride ld
    const uint8_t* mask;   // nullptr or n bytes, 1 = row may be returned
    const uint8_t* mask2;  // a second mask of the same kind (soft deletes and the caller's filter need not be AND-ed first)
    int mask_aliased;      // 1: the masks count rows as row_alias does (insertion order), not storage order
    u64* part;             // [nq][waves][k] per-wave candidate keys
    long long n;           // rows
    int d4;                // float4 per row
    int nch;               // chunks per lane per row (run-time variant)
    int k;
    int capw;              // k > 64: keys per wave-private LDS buffer
    int nt;                // 1: non-temporal row loads
    // IVF mode (ivf_lists != nullptr): grid = (nprobe * ivf_slices, nq), one query per block row
    const long long* ivf_lists;   // [nq][ivf_nprobe] probed list numbers (-1 = none)
    const long long* list_begin;  // [nlist + 1] first storage row of every list
    const long long* row_alias;   // nullptr or storage position -> position reported in keys
    // segment mode (seg_begin != nullptr): grid = (nseg * ivf_slices, nq); block row = query,
    // block column = (segment, slice); rows [seg_begin[s], seg_end[s])
    const long long* seg_begin;
    const long long* seg_end;
    int ivf_nprobe;
    int ivf_slices;
    int grid_y;
    int b16;                      // rows are bf16: d4 = 16-byte chunks of a bf16 row, Q stays fp32 (one query)
    const int* run_if;            // when not NULL the kernel runs only if *run_if != 0
    int nq_valid;                 // query slots of a pass that hold real queries (0 = all); the rest repeat the last one
};

struct MergeArgs {
    const u64* keys;             // [nq][per_query], descending runs of k keys (one per producer)
    long long per_query;
    long long stride;            // keys of query q start at q * stride (0: q * per_query)
    int k;
    int cap;                     // LDS buffer capacity (power of two, >= 2k, >= 2048)
    int fast_cap;                // set by the launcher: candidate slots of the run-head fast path (0 = off)
    float min_score;
    const long long* row_ids;    // nullptr: id = id_base + row
    long long id_base;
    const long long* slot_ids;   // merge of finished lists: slot = list*k + rank lives at
    long long slot_stride;       //   slot_ids[list*slot_stride + q*k + rank]
    const int* run_if;           // when not NULL the merge runs only if *run_if != 0
    float* D;                    // [nq][k]
    long long* I;                // [nq][k]
    long long* rows_out;         // nullptr or [nq][k]: row position of every hit (-1 where I is -1)
};

int scan_rows_per_tile(int d4);
int scan_lists_per_block(int k);
int scan_lanes_per_row(int d4);
hipError_t launch_flat_scan(ScanArgs a, int nq, int blocks, hipStream_t s);
hipError_t launch_merge_keys(const MergeArgs& a, long long nq, hipStream_t s);
hipError_t launch_lists_to_keys(int nlists, long long nq, int k, const float* D, long long d_stride, const long long* I,
                                long long i_stride, u64* keys, hipStream_t s);
hipError_t launch_prep_queries(const float* q, long long nq, int d, int ld, int normalize, float* out, hipStream_t s);
hipError_t launch_ingest_rows(const float* src, long long n, int d, int ld, int normalize, float* dst, hipStream_t s);
hipError_t launch_l2_finish(const float* qbuf, const float* X, long long nq, int d, int ld, int k, int use_limit, float max_d2,
                            const long long* row_ids, long long id_base, float* D, long long* I, hipStream_t s);
hipError_t launch_gather_rows(const float* X, int d, int ld, long long n, const long long* rows, long long ntotal, float* out, hipStream_t s);

}  // namespace rag

// Diagnostic builds (-DRAG_STAMPS): thread 0 of a single-workgroup kernel records the 100 MHz wall clock at
// numbered points in a per-source-file array; RAG_STAMPS_DEFINE(tag) in that file defines the array and
// rag_debug_stamps_<tag>() (not part of the ABI), which copies it out.  Compiled away otherwise.
#ifdef RAG_STAMPS
#define RAG_STAMPS_DEFINE(tag)                                                                        \
    static __device__ long long g_rag_stamps[64];                                                     \
    extern "C" int rag_debug_stamps_##tag(long long* out) {                                           \
        return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rag_stamps), sizeof(long long) * 64);        \
    }
#define RAG_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_rag_stamps[(i)] = (long long)wall_clock64(); } while (0)
#else
#define RAG_STAMPS_DEFINE(tag)
#define RAG_STAMP(i) do { } while (0)
#endif
