// Stable descending sort of (fp64 score, candidate position) pairs inside one workgroup,
// shared by the fusion, hybrid and keyword-ranking kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "topk_keys.h"

namespace rag {

struct SortEntry {
    u64 key;        // ordered fp64 score (larger = better)
    uint32_t idx;   // candidate position (smaller = earlier)
    uint32_t pad;
};

__device__ __forceinline__ u64 f64_ordered(double v) {
    if (v == 0.0) v = 0.0;   // -0.0 and +0.0 compare equal in the reference's sort
    const u64 b = (u64)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// a sorts before b: higher score first, equal scores in candidate order (stable)
__device__ __forceinline__ bool entry_before(const SortEntry& a, const SortEntry& b) {
    return a.key > b.key || (a.key == b.key && a.idx < b.idx);
}

// bitonic sort of n (power of two) entries so that entry_before holds along the array
__device__ inline void sort_entries(SortEntry* e, int n, int tid, int nthreads) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (n >> 1); t += nthreads) {
                const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                const int hi = lo | stride;
                const bool fwd = ((lo & size) == 0);
                const SortEntry x = e[lo], y = e[hi];
                if (entry_before(y, x) == fwd) { e[lo] = y; e[hi] = x; }
            }
        }
    }
    __syncthreads();
}

}  // namespace rag
