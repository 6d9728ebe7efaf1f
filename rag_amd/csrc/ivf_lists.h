// The list-major IVF scan's second form (ivf_lists.hip): argument block and launcher.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

struct rag_index;

namespace rag {

#ifndef RAG_LISTS_WGS_PER_CU
#define RAG_LISTS_WGS_PER_CU 2     // measurement knob: the register budget of the kernel follows it
#endif
constexpr int LS_GRID = 256 * RAG_LISTS_WGS_PER_CU;      // persistent workgroups
constexpr int LS_QT = 32;         // (query, probe) pairs per work item
constexpr int LS_MAX_K = 32;

struct ListScanArgs {
    const float* X;               // rows grouped by list (storage order), row stride ld
    const float* Q;               // [nq][ld] padded queries
    const uint8_t* mask;          // nullptr or one byte per row in INSERTION order (read through row_alias)
    const uint8_t* mask2;         // a second mask of the same kind
    u64* cand;                    // [LS_GRID][waves][LS_QT][64] candidate buffers
    u64* part;                    // [pairs][k]: pair p's k keys at part + pair_slot[p] * k
    unsigned int* gthr;           // [nq]: the best k-th score any list of a query has reached (ordered-float bits, 0 = none)
    int* counters;                // [0] number of work items (device-written), [1] next work item (zero at launch)
    const int* task_list;         // [work items] list of the work item
    const int* task_begin;        // [nlist] first work item of a list
    const int* pair_begin;        // [nlist] first (query, probe) pair of a list
    const int* pair_cnt;          // [nlist] pairs of a list
    const int* pair_q;            // [pairs] query of a pair, grouped by list
    const int* pair_slot;         // [pairs] where the pair's k keys go in part
    const long long* list_begin;  // [nlist + 1] rows of a list (storage order)
    const long long* row_alias;   // nullptr or storage position -> position reported in keys
    int ld;                       // floats per row, a multiple of 128
    int k;                        // <= LS_MAX_K
};

bool ivf_lists_kernel_ok(const rag_index* h, int k);
size_t ivf_lists_cand_bytes();
int ivf_lists_launch(const ListScanArgs& a, hipStream_t s);

}  // namespace rag
