// Sortable 64-bit candidate keys and the wave / workgroup top-k primitives built
// on them (gfx950, wave64).
//
// A candidate (score fp32, row position u32) is packed as
//     key = (ordered(score) << 32) | (0xFFFFFFFF - row)
// where ordered() maps fp32 to u32 monotonically.  Larger key = better hit:
// higher score first, equal scores -> lower row first.  This is the single
// ordering rule of the whole library, so a result does not depend on how the
// scan was partitioned over waves, workgroups or GPUs.  key 0 is "no candidate".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rag {

typedef unsigned long long u64;
constexpr u64 KEY_EMPTY = 0ull;

__host__ __device__ __forceinline__ uint32_t f32_ordered(float f) {
    uint32_t b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = __float_as_uint(f);
#else
    __builtin_memcpy(&b, &f, 4);
#endif
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float ordered_f32(uint32_t o) {
    uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(b);
#else
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
#endif
}
// NaN scores never become candidates.
__device__ __forceinline__ u64 make_key(float score, uint32_t row) {
    if (score != score) return KEY_EMPTY;
    return ((u64)f32_ordered(score) << 32) | (u64)(0xFFFFFFFFu - row);
}
__host__ __device__ __forceinline__ float key_score(u64 k) { return ordered_f32((uint32_t)(k >> 32)); }
__host__ __device__ __forceinline__ uint32_t key_row(u64 k) { return 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull); }

// ---------------------------------------------------------------------------
// DPP helpers.  dpp_ctrl encodings (GCN3+/CDNA): quad_perm 0x00-0xFF,
// row_shr:n 0x110+n, wave_shr:1 0x138, row_mirror 0x140, row_half_mirror 0x141,
// row_bcast:15 0x142, row_bcast:31 0x143.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND_CTRL = true>
__device__ __forceinline__ float dpp_f32(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL,
                                                      ROW_MASK, BANK_MASK, BOUND_CTRL));
}

// Sum over the 64 lanes of a wave; the total is valid in lane 63 only.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_f32<0xB1>(0.f, v);          // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(0.f, v);          // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(0.f, v);         // row_half_mirror
    v += dpp_f32<0x140>(0.f, v);         // row_mirror  -> every lane of a 16-row holds the row sum
    v += dpp_f32<0x142, 0xA, 0xF, false>(0.f, v);   // row_bcast:15 into rows 1 and 3
    v += dpp_f32<0x143, 0xC, 0xF, false>(0.f, v);   // row_bcast:31 into rows 2 and 3
    return v;
}
// Sum over each aligned group of 32 lanes; valid in lanes 31 and 63.
__device__ __forceinline__ float half_sum_to_lane31(float v) {
    v += dpp_f32<0xB1>(0.f, v);
    v += dpp_f32<0x4E>(0.f, v);
    v += dpp_f32<0x141>(0.f, v);
    v += dpp_f32<0x140>(0.f, v);
    v += dpp_f32<0x142, 0xA, 0xF, false>(0.f, v);
    return v;
}
// Sum over each aligned group of 16 lanes; valid in every lane of the group.
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(0.f, v);
    v += dpp_f32<0x4E>(0.f, v);
    v += dpp_f32<0x141>(0.f, v);
    v += dpp_f32<0x140>(0.f, v);
    return v;
}

__device__ __forceinline__ u64 shfl_up1_u64(u64 v) {
    // lane j receives lane j-1 (lane 0 keeps its own value)
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, 0x138, 0xF, 0xF, false);
    hi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, 0x138, 0xF, 0xF, false);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 readlane_u64(u64 v, int lane) {
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}

// Wave-resident sorted list of 64 keys, lane j holds the j-th best.  Insert one
// wave-uniform key (caller has checked key > list[k-1]).
__device__ __forceinline__ void wave_list_insert(u64& list, u64 key, int lane) {
    const u64 better = __ballot(list > key);          // lanes 0..pos-1
    const int pos = __popcll(better);
    const u64 up = shfl_up1_u64(list);
    list = (lane < pos) ? list : ((lane == pos) ? key : up);
}

// ---------------------------------------------------------------------------
// One key per lane: sorting and merging across the 64 lanes of a wave.
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int mask) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((u64)hi << 32) | lo;
}
// a bitonic sequence over the lanes -> descending order (lane 0 the largest)
__device__ __forceinline__ u64 wave_bitonic_merge_desc(u64 v, int lane) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const u64 o = shfl_xor_u64(v, s);
        const bool lower = (lane & s) == 0;
        v = (lower == (v > o)) ? v : o;         // lower lane keeps the larger key
    }
    return v;
}
// any order -> descending order
__device__ __forceinline__ u64 wave_sort_desc(u64 v, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int s = size >> 1; s > 0; s >>= 1) {
            const u64 o = shfl_xor_u64(v, s);
            const bool desc = (size == 64) || ((lane & size) == 0);
            const bool lower = (lane & s) == 0;
            v = ((lower == desc) == (v > o)) ? v : o;
        }
    }
    return v;
}
// the 64 largest of two descending lists, descending
__device__ __forceinline__ u64 wave_merge_top64(u64 a, u64 b, int lane) {
    const u64 r = shfl_u64(b, 63 - lane);
    return wave_bitonic_merge_desc(a > r ? a : r, lane);
}

// ---------------------------------------------------------------------------
// Bitonic sort, descending, of n (power of two) keys in LDS by `nthreads`
// threads that all call it.  SYNC() separates the stages: __syncthreads() for a
// workgroup-shared array, a wave-local fence for a wave-private one.
struct SyncBlock {
    __device__ __forceinline__ void operator()() const { __syncthreads(); }
};
struct SyncWave {
    // LDS operations of one wave complete in issue order; only the compiler and
    // the outstanding-counter need fencing.
    __device__ __forceinline__ void operator()() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

template <typename Sync>
__device__ __forceinline__ void bitonic_sort_desc_lds(u64* a, int n, int tid, int nthreads, Sync sync) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            sync();
            for (int t = tid; t < (n >> 1); t += nthreads) {
                const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                const int hi = lo | stride;
                const bool desc = ((lo & size) == 0);
                const u64 x = a[lo], y = a[hi];
                if ((x < y) == desc) { a[lo] = y; a[hi] = x; }
            }
        }
    }
    sync();
}

__host__ __device__ __forceinline__ int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace rag
