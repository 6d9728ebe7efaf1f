// Exhaustive inner-product scan for one query or a handful of queries, fused with
// top-k selection (gfx950).
//
// Roofline: HBM.  Every corpus row is read exactly once per launch with 16-byte
// per-lane loads (one wave instruction = 1 KiB, or 2 x 512 B, contiguous);
// arithmetic is ~1/30 of the VALU budget at batch 1, so the kernel's job is to
// keep enough loads in flight and never touch a row twice.  Algorithmic bytes
// per row = 4*ld (+1 with a row mask).
//
// Selection: each wave keeps its own candidate list and never synchronises with
// other waves inside the scan loop.
//   k <= 64  : a sorted list of 64 keys, one per lane (DPP shift insert).
//   k <= 1024: a wave-private LDS buffer of 2*pow2(k) keys, bitonic-sorted and cut
//              to k whenever it fills.
// A row is a candidate only if its key beats the wave's current k-th key, which
// after warm-up is rare.  For k <= 64 the four lists of a block are merged in
// registers before they are written (one list per block and query in `part`);
// above, every wave writes its own.  A second, tiny kernel merges the lists per
// query: merge_keys_small_kernel (k <= 64, register lists and bitonic merges across
// lanes) or merge_keys_kernel (larger k, bounded by the run heads).
//
// Three row sets: the whole corpus (tiles dealt over all waves), the probed lists
// of an IVF index, or caller-given row ranges (one block per range slice,
// blockIdx.y = query).  Rows may also be the bf16 copy of the corpus (B16): same
// kernel, half the bytes, fp32 query and accumulation.
#include <float.h>

#include "topk_keys.h"
#include "kernels.h"
#include "sort_entries.h"

namespace rag {

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (LPR == 64) return wave_sum_to_lane63(v);
    else if constexpr (LPR == 32) return half_sum_to_lane31(v);
    else return row16_sum(v);
}

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dot4(const f4& a, const f4& b, float acc) {
    acc = fmaf(a[0], b[0], acc);
    acc = fmaf(a[1], b[1], acc);
    acc = fmaf(a[2], b[2], acc);
    acc = fmaf(a[3], b[3], acc);
    return acc;
}
__device__ __forceinline__ f4 f4_zero() { return (f4){0.f, 0.f, 0.f, 0.f}; }
// a 16-byte chunk of a bf16 row holds elements e0..e7, two per dword (even in the low half):
// bf16 -> fp32 is a shift, so the chunk unpacks into (e0,e2,e4,e6) and (e1,e3,e5,e7)
__device__ __forceinline__ f4 bf16_even(const f4& raw) {
    f4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __uint_as_float(__float_as_uint(raw[j]) << 16);
    return o;
}
__device__ __forceinline__ f4 bf16_odd(const f4& raw) {
    f4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __uint_as_float(__float_as_uint(raw[j]) & 0xFFFF0000u);
    return o;
}

// LPR  lanes that share one row (64 / 32 / 16); RPS = 64/LPR rows per wave step
// CH   float4 chunks per lane per row, compile time (0: run-time count a.nch)
// U    wave steps per tile (rows in flight per wave = U*RPS)
// NQ   queries scored against every row (1: query fragment in registers; >1: LDS)
// BIGK false: register list (k <= 64); true: LDS buffer (NQ must be 1)
template <bool NT>
__device__ __forceinline__ f4 load_row4(const f4* p) {
    // rows are read once per launch: a non-temporal load keeps them from
    // displacing the queries and partial lists in L2 / Infinity Cache
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// B16  rows are bf16 (d4 counts their 16-byte chunks, 8 elements each), the query stays fp32;
//      one query, compile-time chunk count only
template <int LPR, int CH, int U, int NQ, bool BIGK, bool NT, bool B16 = false>
__global__ __launch_bounds__(256) void flat_scan_kernel(ScanArgs a) {
    constexpr int RPS = 64 / LPR;
    constexpr int TILE = RPS * U;
    static_assert(!BIGK || NQ == 1, "LDS-buffer selection handles one query per launch");
    static_assert(!B16 || (NQ == 1 && CH > 0), "bf16 rows: one query, compile-time width");
    if (a.run_if && *a.run_if == 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int nwaves = gridDim.x * 4;
    const int w = blockIdx.x * 4 + wib;
    const int d4 = a.d4;
    const int nch = CH ? CH : a.nch;
    const f4* __restrict__ X4 = reinterpret_cast<const f4*>(a.X);
    const f4* __restrict__ Q4 = reinterpret_cast<const f4*>(a.Q) + (size_t)blockIdx.y * d4 * NQ;

    // rows this wave walks: tiles t0, t0 + tstride, ... of [r_begin, r_end).
    //   exhaustive: the whole corpus, tiles dealt round-robin over every wave of the grid
    //   IVF: blockIdx.y = query, blockIdx.x = (probed list, slice of the list); the
    //        block's four waves share the slice
    long long r_begin = 0, r_end = a.n, t0 = w, tstride = nwaves;
    if (a.ivf_lists || a.seg_begin) {
        const int probe = blockIdx.x / a.ivf_slices, slice = blockIdx.x % a.ivf_slices;
        long long lb = 0, le = 0;
        if (a.seg_begin) {
            lb = a.seg_begin[probe];
            le = a.seg_end[probe];
        } else {
            const long long lst = a.ivf_lists[(size_t)blockIdx.y * a.ivf_nprobe + probe];
            if (lst >= 0) { lb = a.list_begin[lst]; le = a.list_begin[lst + 1]; }
        }
        const long long ltiles = (le - lb + TILE - 1) / TILE;
        r_begin = lb + (ltiles * slice / a.ivf_slices) * TILE;
        const long long e = lb + (ltiles * (slice + 1) / a.ivf_slices) * TILE;
        r_end = e < le ? e : le;
        if (r_end < r_begin) r_end = r_begin;
        t0 = wib;
        tstride = 4;
    }

    // ---- query fragments --------------------------------------------------
    constexpr int QREG = (NQ == 1 && CH > 0) ? (B16 ? 2 * CH : CH) : 1;
    f4 qf[QREG];
    f4* qlds = reinterpret_cast<f4*>(smem);          // [NQ][d4] when not in registers
    constexpr bool Q_IN_REGS = (NQ == 1 && CH > 0);
    if constexpr (Q_IN_REGS) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = c * LPR + l;
            if constexpr (B16) {
                // the fp32 query elements that face the even / odd elements of row chunk `col`
                const f4 lo = (col < d4) ? Q4[2 * col] : f4_zero();
                const f4 hi = (col < d4) ? Q4[2 * col + 1] : f4_zero();
                qf[2 * c] = (f4){lo[0], lo[2], hi[0], hi[2]};
                qf[2 * c + 1] = (f4){lo[1], lo[3], hi[1], hi[3]};
            } else {
                qf[c] = (col < d4) ? Q4[col] : f4_zero();
            }
        }
    } else {
        const int valid = (a.nq_valid > 0 && a.nq_valid < NQ) ? a.nq_valid : NQ;
        for (int i = threadIdx.x; i < NQ * d4; i += 256) {
            const int qi = i / d4;
            qlds[i] = Q4[qi < valid ? i : i - (qi - (valid - 1)) * d4];
        }
        __syncthreads();
    }
    // wave-private LDS candidate buffer (BIGK), placed after the query image
    u64* wbuf = nullptr;
    int capw = 0;
    if constexpr (BIGK) {
        capw = a.capw;
        const size_t qbytes = Q_IN_REGS ? 0 : (((size_t)NQ * d4 * 16 + 15) & ~(size_t)15);
        wbuf = reinterpret_cast<u64*>(smem + qbytes) + (size_t)wib * capw;
    }

    // ---- selection state ----------------------------------------------------
    u64 list[NQ];
    u64 thr[NQ];
    float thr_f[NQ];
    int cnt = 0;   // BIGK: keys in wbuf
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) { list[qi] = KEY_EMPTY; thr[qi] = KEY_EMPTY; thr_f[qi] = -INFINITY; }
    const int k = a.k;

    const long long ntiles = (r_end - r_begin + TILE - 1) / TILE;
    for (long long tile = t0; tile < ntiles; tile += tstride) {
        const long long row0 = r_begin + tile * TILE;
        float acc[U][NQ];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) acc[u][qi] = 0.f;

        // rows of the tile that exist and are not masked out: bit j = row0 + j
        bool rv = false;
        if (lane < TILE) {
            const long long r = row0 + lane;
            rv = r < r_end;
            if (rv && (a.mask || a.mask2)) {
                const long long mr = a.mask_aliased ? a.row_alias[r] : r;
                rv = (!a.mask || a.mask[mr] != 0) && (!a.mask2 || a.mask2[mr] != 0);
            }
        }
        const u64 rowvalid = __ballot(rv);

        if constexpr (CH > 0) {
            f4 x[U][CH];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                long long r = row0 + u * RPS + sub;
                r = r < r_end ? r : r_end - 1;
                const f4* xr = X4 + r * (long long)d4;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int col = c * LPR + l;
                    x[u][c] = (col < d4) ? load_row4<NT>(xr + col) : f4_zero();
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (B16) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        acc[u][0] = dot4(bf16_even(x[u][c]), qf[2 * c], acc[u][0]);
                        acc[u][0] = dot4(bf16_odd(x[u][c]), qf[2 * c + 1], acc[u][0]);
                    }
                } else if constexpr (Q_IN_REGS) {
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[u][0] = dot4(x[u][c], qf[c], acc[u][0]);
                } else {
                    const int col = c * LPR + l;
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) {
                        const f4 qv = (col < d4) ? qlds[qi * d4 + col] : f4_zero();
#pragma unroll
                        for (int u = 0; u < U; ++u) acc[u][qi] = dot4(x[u][c], qv, acc[u][qi]);
                    }
                }
            }
        } else {
            // run-time chunk count: three chunks of every row of the tile per round, so that a
            // wave keeps 3*U loads in flight as the compile-time paths do
            constexpr int CG = 3;
            for (int c0 = 0; c0 < nch; c0 += CG) {
                f4 x[U][CG];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    long long r = row0 + u * RPS + sub;
                    r = r < r_end ? r : r_end - 1;
                    const f4* xr = X4 + r * (long long)d4;
#pragma unroll
                    for (int g = 0; g < CG; ++g) {
                        // always load (a clamped column of the same row) and select afterwards: a
                        // load under a branch is waited for before the next one is issued
                        const int col = (c0 + g) * LPR + l;
                        const f4 v = load_row4<NT>(xr + (col < d4 ? col : d4 - 1));
                        x[u][g] = (col < d4) ? v : f4_zero();
                    }
                }
#pragma unroll
                for (int g = 0; g < CG; ++g) {
                    const int col = (c0 + g) * LPR + l;
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) {
                        const f4 qraw = qlds[qi * d4 + (col < d4 ? col : d4 - 1)];
                        const f4 qv = (col < d4) ? qraw : f4_zero();
#pragma unroll
                        for (int u = 0; u < U; ++u) acc[u][qi] = dot4(x[u][g], qv, acc[u][qi]);
                    }
                }
            }
        }

        // ---- reduce over the lanes of a row; result sits in the group's last lane
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) acc[u][qi] = group_sum<LPR>(acc[u][qi]);

        // ---- candidate test + insert (wave-uniform control flow) ---------------
        const bool result_lane = (l == LPR - 1);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u64 uvalid = (rowvalid >> (u * RPS)) & ((RPS == 64) ? ~0ull : ((1ull << RPS) - 1));
            // expand "row s of this step is valid" to the result lane of group s
            const bool lane_ok = result_lane && ((uvalid >> sub) & 1ull);
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                u64 m = __ballot(lane_ok && acc[u][qi] >= thr_f[qi]);
                while (m) {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[u][qi]), j));
                    long long rowp = row0 + u * RPS + j / LPR;
                    if (a.row_alias) rowp = a.row_alias[rowp];      // storage position -> insertion position
                    const u64 key = make_key(sc, (uint32_t)rowp);
                    if (key > thr[qi]) {
                        if constexpr (!BIGK) {
                            wave_list_insert(list[qi], key, lane);
                            thr[qi] = readlane_u64(list[qi], k - 1);
                        } else {
                            if (lane == 0) wbuf[cnt] = key;
                            ++cnt;
                            if (cnt == capw) {
                                bitonic_sort_desc_lds(wbuf, capw, lane, 64, SyncWave());
                                cnt = k;
                                thr[qi] = wbuf[k - 1];
                            }
                        }
                        thr_f[qi] = (thr[qi] == KEY_EMPTY) ? -INFINITY : key_score(thr[qi]);
                    }
                }
            }
        }
    }

    // ---- publish this wave's k keys per query -----------------------------------
    if constexpr (!BIGK) {
        // the block's four lists are merged here (three bitonic merges in registers), so the
        // merge kernel reads one list per block
        const size_t qbytes = Q_IN_REGS ? 0 : (((size_t)NQ * d4 * 16 + 15) & ~(size_t)15);
        u64* comb = reinterpret_cast<u64*>(smem + qbytes);          // [NQ][3][64]
        if (wib > 0) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) comb[(qi * 3 + wib - 1) * 64 + lane] = list[qi];
        }
        __syncthreads();
        if (wib == 0) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                u64 best = list[qi];
#pragma unroll
                for (int j = 0; j < 3; ++j) best = wave_merge_top64(best, comb[(qi * 3 + j) * 64 + lane], lane);
                if (lane < k) a.part[(((size_t)blockIdx.y * NQ + qi) * gridDim.x + blockIdx.x) * k + lane] = best;
            }
        }
    } else {
        // pad with empties, sort, write the first k
        for (int i = cnt + lane; i < capw; i += 64) wbuf[i] = KEY_EMPTY;
        bitonic_sort_desc_lds(wbuf, capw, lane, 64, SyncWave());
        for (int i = lane; i < k; i += 64) a.part[((size_t)blockIdx.y * nwaves + w) * k + i] = wbuf[i];
    }
}

// final step of every merge: key -> (score, id) with the min_score cut and (-FLT_MAX, -1) padding
__device__ __forceinline__ void finalise_hit(const MergeArgs& a, long long q, int j, u64 key) {
    const int k = a.k;
    float s = -3.402823466e+38f;
    long long id = -1;
    if (key != KEY_EMPTY) {
        const float sc = key_score(key);
        if (sc >= a.min_score) {
            const uint32_t slot = key_row(key);
            s = sc;
            if (a.slot_ids) {   // merge of finished lists: slot = list*k + rank
                const uint32_t g = slot / (uint32_t)k, r = slot % (uint32_t)k;
                id = a.slot_ids[(size_t)g * a.slot_stride + (size_t)q * k + r];
            }
            else if (a.row_ids) id = a.row_ids[slot];
            else id = a.id_base + (long long)slot;
        }
    }
    a.D[(size_t)q * k + j] = s;
    a.I[(size_t)q * k + j] = id;
    if (a.rows_out) a.rows_out[(size_t)q * k + j] = s < -3.0e38f ? -1 : (long long)key_row(key);
}

// ---------------------------------------------------------------------------
// Merge P key lists of length k per query into the final k hits (one workgroup
// per query).  Keys stream through an LDS buffer: a key is appended only if it
// beats the current k-th key; the buffer is bitonic-sorted and cut to k when it
// cannot take another 1024 keys.
RAG_STAMPS_DEFINE(merge)
__global__ __launch_bounds__(1024) void merge_keys_kernel(MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* buf = reinterpret_cast<u64*>(smem);
    const int tid = threadIdx.x;
    const int cap = a.cap;
    const int k = a.k;
    u64* s_thr = buf + cap;                               // [cap]   current k-th key
    int* s_cnt = reinterpret_cast<int*>(buf + cap + 1);   // [cap+1] keys in buf
    u64* cand = buf + cap + 2;                            // [MERGE_FAST_CAP] fast path only
    const long long q = blockIdx.x;
    if (a.run_if && *a.run_if == 0) return;
    const u64* in = a.keys + (size_t)q * (a.stride ? a.stride : a.per_query);

    // ---- fast path: the input is P descending runs of k keys.  The k-th largest run head is a
    // lower bound of the answer's k-th key (k different keys reach it), and only keys at or above
    // it matter: each thread walks its runs from the head and stops at the first key below.
    const long long P = a.per_query / k;
    if (a.fast_cap > 0 && P >= k && P <= cap && P * k == a.per_query) {
        const int Pn = (int)P;
        RAG_STAMP(0);
        // a lower bound of the answer's k-th key from the run heads: any value with k heads at or above it.
        // The heads (thousands when every wave of the scan hands over a run) are binned over their own range
        // into 1024 buckets; the lower edge of the bucket in which the count from the top reaches k is within
        // 1/1024 of the range of the k-th largest head itself -- a few hundred keys are then left to rank, where
        // the minimum over k groups of heads left thousands and an exact rank of the heads costs P^2 compares
        int* hist = reinterpret_cast<int*>(cand);               // [1024], before cand is filled
        u64* s_lo = s_thr;                                      // the smallest head lives in *s_thr until the bound replaces it
        u64* s_hi = cand + 512;                                 // behind the histogram
        for (int i = tid; i < 1024; i += 1024) hist[i] = 0;
        if (tid == 0) { *s_cnt = 0; *s_lo = ~0ull; *s_hi = 0ull; }
        __syncthreads();
        {
            // range of the heads: reduced inside each wave first (thousands of LDS atomics on one address
            // would run one after the other: 19 us for 2048 heads)
            u64 mn = ~0ull, mx = 0ull;
            for (int i = tid; i < Pn; i += 1024) {
                const u64 hd = in[(size_t)i * k];
                buf[i] = hd;
                if (hd != KEY_EMPTY) { mn = hd < mn ? hd : mn; mx = hd > mx ? hd : mx; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const u64 omn = shfl_xor_u64(mn, off), omx = shfl_xor_u64(mx, off);
                mn = omn < mn ? omn : mn;
                mx = omx > mx ? omx : mx;
            }
            if ((tid & 63) == 0 && mx != 0ull) { atomicMin(s_lo, mn); atomicMax(s_hi, mx); }
        }
        __syncthreads();
        RAG_STAMP(1);
        const u64 lo = *s_lo, hi = *s_hi;
        const int shift = (hi > lo) ? max(0, 64 - __clzll((long long)(hi - lo)) - 10) : 0;      // (hi - lo) >> shift < 1024
        __syncthreads();
        if (hi >= lo)
            for (int i = tid; i < Pn; i += 1024)
                if (buf[i] != KEY_EMPTY) atomicAdd(&hist[(int)((buf[i] - lo) >> shift)], 1);
        if (tid == 0) *s_thr = KEY_EMPTY;
        __syncthreads();
        if (tid < 64 && hi >= lo) {
            // wave 0: lane l owns buckets 16 l .. 16 l + 15.  suf = heads in the buckets of lanes >= l (suffix scan),
            // the lane in which the count from the top reaches k walks its own 16 buckets downwards
            int own = 0;
#pragma unroll
            for (int b2 = 0; b2 < 16; ++b2) own += hist[tid * 16 + b2];
            int suf = own;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_down(suf, off, 64);
                if (tid + off < 64) suf += up;
            }
            const int higher = suf - own;                       // heads in the buckets of the lanes above
            if (higher < k && suf >= k) {
                int cnt = higher, b2 = 15;
                for (; b2 >= 0; --b2) {
                    cnt += hist[tid * 16 + b2];
                    if (cnt >= k) break;
                }
                *s_thr = lo + ((u64)(tid * 16 + b2) << shift);  // every key at or above it falls in this bucket or a higher one
            }
        }
        __syncthreads();
        RAG_STAMP(2);
        const u64 bound = *s_thr;
        if (bound != KEY_EMPTY) {
            // a run is walked four keys at a time (four independent loads, one memory latency), its head
            // taken from LDS
            for (int p = tid; p < Pn; p += 1024) {
                if (buf[p] < bound) continue;
                const u64* run = in + (size_t)p * k;
                bool more = true;
                for (int j0 = 0; j0 < k && more; j0 += 4) {
                    u64 key[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) key[j] = j0 + j < k ? run[j0 + j] : KEY_EMPTY;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!more || key[j] < bound || key[j] == KEY_EMPTY) { more = false; continue; }
                        const int pos = atomicAdd(s_cnt, 1);
                        if (pos < a.fast_cap) cand[pos] = key[j];
                    }
                }
            }
            __syncthreads();
            RAG_STAMP(3);
            const int c = *s_cnt;                        // c >= k: k run heads reach the bound
#ifdef RAG_STAMPS
            if (tid == 0) { g_rag_stamps[8] = c; g_rag_stamps[9] = Pn; }
#endif
            if (c <= a.fast_cap) {
                // the k best of the c keys by counting, not sorting: keys are distinct (the row sits in the low
                // bits), so "the number of keys above mine" is a rank, and a count over an LDS array that every
                // lane reads at the same address (a broadcast) needs no barrier between its steps
                for (int i = tid; i < c; i += 1024) {
                    const u64 mine = cand[i];
                    int above = 0;
#pragma unroll 8
                    for (int j = 0; j < c; ++j) above += cand[j] > mine ? 1 : 0;
                    if (above < k) finalise_hit(a, q, above, mine);
                }
                RAG_STAMP(4);
                return;
            }
        }
        __syncthreads();      // fewer than k non-empty runs, or too many keys tie near the bound: stream everything
    }

    for (int i = tid; i < cap; i += 1024) buf[i] = KEY_EMPTY;
    if (tid == 0) { *s_cnt = 0; *s_thr = KEY_EMPTY; }
    __syncthreads();
    for (long long base = 0; base < a.per_query; base += 1024) {
        const int c = *s_cnt;
        __syncthreads();              // every thread has read the count before anyone appends
        const long long left = a.per_query - base;
        if (c + (left < 1024 ? (int)left : 1024) > cap) {
            bitonic_sort_desc_lds(buf, cap, tid, 1024, SyncBlock());
            const u64 t = buf[k - 1];
            __syncthreads();
            for (int i = k + tid; i < cap; i += 1024) buf[i] = KEY_EMPTY;
            if (tid == 0) { *s_cnt = k; *s_thr = t; }
            __syncthreads();
        }
        const u64 thr = *s_thr;
        const long long i = base + tid;
        const u64 key = (i < a.per_query) ? in[i] : KEY_EMPTY;
        if (key > thr) {
            const int pos = atomicAdd(s_cnt, 1);
            buf[pos] = key;
        }
        __syncthreads();
    }
    bitonic_sort_desc_lds(buf, cap, tid, 1024, SyncBlock());
    for (int j = tid; j < k; j += 1024) finalise_hit(a, q, j, buf[j]);
}

// The same merge for k <= 64 in registers: every wave streams its share of the keys through a
// sorted list held one key per lane.  Of each 64 keys only those that beat the wave's k-th are
// looked at: a few are inserted one by one, many are sorted across the lanes and merged in one
// bitonic step.  The waves' lists are then merged pairwise (tree over LDS).
__global__ __launch_bounds__(1024) void merge_keys_small_kernel(MergeArgs a) {
    __shared__ u64 s_lists[16 * 64];
    if (a.run_if && *a.run_if == 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    const int k = a.k;
    const long long q = blockIdx.x;
    const u64* in = a.keys + (size_t)q * (a.stride ? a.stride : a.per_query);
    u64 list = KEY_EMPTY, thr = KEY_EMPTY;
    // a wave's chunks of 64 keys are requested MU at a time before the first is looked at: the
    // keys sit in L2 / HBM, and one dependent load per chunk made the latency of every chunk the
    // cost of the merge (8 chunks per wave: 11 us; requested together: one latency)
    constexpr int MU = 8;
    const long long step = (long long)nw * 64;
    for (long long base0 = (long long)wave * 64; base0 < a.per_query; base0 += step * MU) {
        u64 keys[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const long long i = base0 + u * step + lane;
            keys[u] = i < a.per_query ? in[i] : KEY_EMPTY;
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const u64 key = keys[u];
            const bool cand = key > thr;
            u64 m = __ballot(cand);
            if (m == 0) continue;
            if (__popcll(m) <= 6) {
                while (m) {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    const u64 kj = readlane_u64(key, j);
                    if (kj > thr) {
                        wave_list_insert(list, kj, lane);
                        thr = readlane_u64(list, k - 1);
                    }
                }
            } else {
                // every producer writes descending runs of k keys, so with k = 64 the 64 keys a wave
                // reads are one run, and cutting its tail at the threshold keeps it descending
                u64 run = cand ? key : KEY_EMPTY;
                if (k != 64) run = wave_sort_desc(run, lane);
                list = wave_merge_top64(list, run, lane);
                thr = readlane_u64(list, k - 1);
            }
        }
    }
    s_lists[wave * 64 + lane] = list;
    for (int stride = 1; stride < nw; stride <<= 1) {
        __syncthreads();
        if ((wave & (2 * stride - 1)) == 0 && wave + stride < nw) {
            list = wave_merge_top64(list, s_lists[(wave + stride) * 64 + lane], lane);
            s_lists[wave * 64 + lane] = list;
        }
    }
    if (wave == 0 && lane < k) finalise_hit(a, q, lane, list);
}

// (D, I) result lists -> keys whose "row" is the slot (list*k + rank) in the
// gathered set; output keys [nq][nlists*k]; an unfilled slot (I < 0) gives no key.
__global__ void lists_to_keys_kernel(int nlists, long long nq, int k, const float* D, long long d_stride,
                                     const long long* I, long long i_stride, u64* keys) {
    const long long total = (long long)nlists * nq * k;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(t % k);
        const long long q = (t / k) % nq;
        const int g = (int)(t / ((long long)k * nq));
        const uint32_t slot = (uint32_t)(g * k + j);
        const long long id = I[(size_t)g * i_stride + (size_t)q * k + j];
        const float sc = D[(size_t)g * d_stride + (size_t)q * k + j];
        keys[(size_t)q * nlists * k + slot] = (id < 0) ? KEY_EMPTY : make_key(sc, slot);
    }
}

// queries -> padded (ld) copies; mode 1 (COSINE): normalised; mode 2 (L2): column d holds 1.0, which
// meets the rows' -|x|^2 / 2 there, so the inner product of the padded vectors is q.x - |x|^2 / 2
__global__ void prep_queries_kernel(const float* q, int d, int ld, int normalize, float* out) {
    const long long qi = blockIdx.x;
    const float* src = q + qi * d;
    float* dst = out + qi * ld;
    __shared__ float s_part[4];
    float ss = 0.f;
    for (int i = threadIdx.x; i < d; i += blockDim.x) { const float v = src[i]; ss += v * v; }
    ss = wave_sum_to_lane63(ss);
    if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += s_part[i];
    float inv = 1.f;
    if (normalize == 1) { const float nrm = sqrtf(tot); inv = nrm > 0.f ? 1.f / nrm : 1.f; }
    for (int i = threadIdx.x; i < ld; i += blockDim.x) dst[i] = (i < d) ? src[i] * inv : (normalize == 2 && i == d) ? 1.f : 0.f;
}

// L2 index: the search ranked the rows by q.x - |x|^2 / 2 and left the ROW POSITIONS of the k best in I.
// Here every kept row gets its squared distance as FAISS's and Qdrant's scans compute it -- the sum of
// the squared fp32 differences (q - x)_i, accumulated in fp64 and rounded once, so a row equal to the
// query is at distance exactly 0 -- the list is sorted by (distance, row position), rows farther than
// max_d2 are dropped when a limit is given, and positions become ids.  Empty slots: (-1, FLT_MAX).
// One workgroup per query; a wave per candidate row.
__global__ __launch_bounds__(256) void l2_finish_kernel(const float* qbuf, const float* X, int d, int ld, int k, int use_limit,
                                                        float max_d2, const long long* row_ids, long long id_base, float* D,
                                                        long long* I) {
    __shared__ SortEntry e[1024];
    const long long qi = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* q = qbuf + qi * ld;
    int P = 64;
    while (P < k) P <<= 1;
    for (int j = wave; j < P; j += 4) {
        const long long pos = j < k ? I[(size_t)qi * k + j] : -1;      // wave-uniform
        double ss = 0.0;
        if (pos >= 0) {
            const float* x = X + pos * (long long)ld;
            for (int i = lane; i < d; i += 64) {
                const float f = q[i] - x[i];
                ss = fma((double)f, (double)f, ss);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        }
        if (lane == 0) {
            const float dist = (float)ss;
            SortEntry t;
            t.key = pos >= 0 ? f64_ordered(-(double)dist) : 0ull;       // nearest first; empty slots last
            t.idx = pos >= 0 ? (uint32_t)pos : 0xFFFFFFFFu;
            t.pad = __float_as_uint(dist);
            e[j] = t;
        }
    }
    sort_entries(e, P, tid, 256);
    for (int j = tid; j < k; j += 256) {
        const SortEntry t = e[j];
        float dist = FLT_MAX;
        long long id = -1;
        if (t.idx != 0xFFFFFFFFu) {
            const float v = __uint_as_float(t.pad);
            if (!use_limit || v <= max_d2) {
                dist = v;
                id = row_ids ? row_ids[t.idx] : id_base + (long long)t.idx;
            }
        }
        D[(size_t)qi * k + j] = dist;
        I[(size_t)qi * k + j] = id;
    }
}

// rows: copy n x d -> n x ld (zero padded); mode 1: normalised; mode 2 (L2): column d = -|x|^2 / 2,
// the norm summed in fp64 (one wave per row)
__global__ void ingest_rows_kernel(const float* src, long long n, int d, int ld, int normalize, float* dst) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave; r < n; r += nw) {
        const float* s = src + r * d;
        float ss = 0.f;
        for (int i = lane; i < d; i += 64) { const float v = s[i]; ss += v * v; }
        float inv = 1.f;
        float aug = 0.f;
        if (normalize == 2) {
            double sd = 0.0;
            for (int i = lane; i < d; i += 64) sd = fma((double)s[i], (double)s[i], sd);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sd += __shfl_xor(sd, off, 64);
            aug = (float)(-0.5 * sd);
        } else if (normalize) {
            ss = wave_sum_to_lane63(ss);
            const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ss), 63));
            const float nrm = sqrtf(tot);
            inv = nrm > 0.f ? 1.f / nrm : 1.f;
        }
        float* o = dst + r * ld;
        for (int i = lane; i < ld; i += 64) o[i] = (i < d) ? s[i] * inv : (i == d) ? aug : 0.f;
    }
}

__global__ void gather_rows_kernel(const float* X, int d, int ld, long long n, const long long* rows, long long ntotal, float* out) {
    const long long r = blockIdx.x;
    if (r >= n) return;
    const long long src = rows[r];
    for (int i = threadIdx.x; i < d; i += blockDim.x)
        out[r * d + i] = (src >= 0 && src < ntotal) ? X[src * ld + i] : 0.f;
}

// ---------------------------------------------------------------------------
// host-side launchers
template <int LPR, int CH, int U, int NQ, bool BIGK, bool NT, bool B16 = false>
static hipError_t launch_scan_t2(const ScanArgs& a, int blocks, size_t lds, hipStream_t s) {
    auto kern = flat_scan_kernel<LPR, CH, U, NQ, BIGK, NT, B16>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(blocks, (a.ivf_lists || a.seg_begin) ? a.grid_y : 1), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int LPR, int CH, int U, int NQ, bool BIGK>
static hipError_t launch_scan_t(const ScanArgs& a, int blocks, size_t lds, hipStream_t s) {
    if constexpr (NQ == 1 && CH > 0) {
        if (a.b16) return a.nt ? launch_scan_t2<LPR, CH, U, 1, BIGK, true, true>(a, blocks, lds, s)
                               : launch_scan_t2<LPR, CH, U, 1, BIGK, false, true>(a, blocks, lds, s);
    }
    if (a.b16) return hipErrorInvalidValue;
    return a.nt ? launch_scan_t2<LPR, CH, U, NQ, BIGK, true>(a, blocks, lds, s)
                : launch_scan_t2<LPR, CH, U, NQ, BIGK, false>(a, blocks, lds, s);
}

template <int NQ, bool BIGK>
static hipError_t launch_scan_nq(const ScanArgs& a, int blocks, size_t lds, hipStream_t s) {
    const int d4 = a.d4;
    // compile-time chunk counts for the common embedding widths
    if (d4 == 192) return launch_scan_t<64, 3, 4, NQ, BIGK>(a, blocks, lds, s);          // d = 768
    if (d4 == 256) return launch_scan_t<64, 4, 2, NQ, BIGK>(a, blocks, lds, s);          // d = 1024
    if (d4 == 384) return launch_scan_t<64, 6, 2, NQ, BIGK>(a, blocks, lds, s);          // d = 1536
    if (d4 == 768) return launch_scan_t<64, 12, 1, NQ, BIGK>(a, blocks, lds, s);         // d = 3072
    if (d4 == 96) return launch_scan_t<32, 3, 4, NQ, BIGK>(a, blocks, lds, s);           // d = 384
    if (d4 == 128) return launch_scan_t<64, 2, 4, NQ, BIGK>(a, blocks, lds, s);          // d = 512
    if (d4 == 64) return launch_scan_t<64, 1, 4, NQ, BIGK>(a, blocks, lds, s);           // d = 256
    if (d4 == 32) return launch_scan_t<32, 1, 4, NQ, BIGK>(a, blocks, lds, s);           // d = 128
    if (d4 <= 16) return launch_scan_t<16, 1, 4, NQ, BIGK>(a, blocks, lds, s);           // d <= 64
    if (d4 <= 32) return launch_scan_t<32, 0, 4, NQ, BIGK>(a, blocks, lds, s);
    return launch_scan_t<64, 0, 4, NQ, BIGK>(a, blocks, lds, s);
}

// partial lists a scan block leaves per query: one (merged in the block) for k <= 64, one per wave above
int scan_lists_per_block(int k) { return k > 64 ? 4 : 1; }

int scan_rows_per_tile(int d4) {
    if (d4 == 256 || d4 == 384) return 2;
    if (d4 == 768) return 1;
    if (d4 == 96 || d4 == 32) return 8;
    if (d4 <= 16) return 16;
    if (d4 < 32) return 8;
    return 4;
}
int scan_lanes_per_row(int d4) {
    if (d4 == 96 || d4 == 32) return 32;
    if (d4 <= 16) return 16;
    if (d4 < 32) return 32;
    return 64;
}

hipError_t launch_flat_scan(ScanArgs a, int nq, int blocks, hipStream_t s) {
    const bool bigk = a.k > 64;
    const int lpr = scan_lanes_per_row(a.d4);
    a.nch = (a.d4 + lpr - 1) / lpr;
    // queries live in registers only for NQ == 1 with a compile-time chunk count
    const bool q_in_regs = (nq == 1) && (a.d4 == 192 || a.d4 == 256 || a.d4 == 384 || a.d4 == 768 || a.d4 == 96 || a.d4 == 128 || a.d4 == 64 || a.d4 == 32 || a.d4 <= 16);
    size_t lds = q_in_regs ? 0 : (((size_t)nq * a.d4 * 16 + 15) & ~(size_t)15);
    if (!bigk) lds += (size_t)nq * 3 * 64 * sizeof(u64);          // lists of waves 1..3, merged by wave 0
    if (bigk) {
        a.capw = 2 * pow2_ceil(a.k);
        lds += (size_t)4 * a.capw * sizeof(u64);
        return launch_scan_nq<1, true>(a, blocks, lds, s);
    }
    switch (nq) {
        case 1: return launch_scan_nq<1, false>(a, blocks, lds, s);
        case 2: return launch_scan_nq<2, false>(a, blocks, lds, s);
        case 4: return launch_scan_nq<4, false>(a, blocks, lds, s);
        case 8: return launch_scan_nq<8, false>(a, blocks, lds, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_merge_keys(const MergeArgs& a, long long nq, hipStream_t s) {
    static const bool small_ok = !(getenv("RAG_MERGE_SORT") && atoi(getenv("RAG_MERGE_SORT")) != 0);
    if (a.k <= 64 && small_ok) {
        long long waves = (a.per_query + 511) / 512;           // ~8 ballots of 64 keys per wave
        waves = waves < 1 ? 1 : waves > 16 ? 16 : waves;
        hipLaunchKernelGGL(merge_keys_small_kernel, dim3((unsigned)nq), dim3((unsigned)waves * 64), 0, s, a);
        return hipGetLastError();
    }
    MergeArgs b = a;
    static const bool fast_ok = !(getenv("RAG_MERGE_FAST") && atoi(getenv("RAG_MERGE_FAST")) == 0);
    const long long runs = a.per_query / a.k;
    b.fast_cap = (fast_ok && runs >= a.k && runs <= a.cap) ? 4096 : 0;
    const size_t lds = (size_t)(a.cap + 2 + b.fast_cap) * sizeof(u64);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(merge_keys_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(merge_keys_kernel, dim3((unsigned)nq), dim3(1024), lds, s, b);
    return hipGetLastError();
}

hipError_t launch_lists_to_keys(int nlists, long long nq, int k, const float* D, long long d_stride, const long long* I,
                                long long i_stride, u64* keys, hipStream_t s) {
    const long long total = (long long)nlists * nq * k;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(lists_to_keys_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, nlists, nq, k, D, d_stride, I, i_stride, keys);
    return hipGetLastError();
}

hipError_t launch_prep_queries(const float* q, long long nq, int d, int ld, int normalize, float* out, hipStream_t s) {
    hipLaunchKernelGGL(prep_queries_kernel, dim3((unsigned)nq), dim3(256), 0, s, q, d, ld, normalize, out);
    return hipGetLastError();
}

hipError_t launch_l2_finish(const float* qbuf, const float* X, long long nq, int d, int ld, int k, int use_limit, float max_d2,
                            const long long* row_ids, long long id_base, float* D, long long* I, hipStream_t s) {
    hipLaunchKernelGGL(l2_finish_kernel, dim3((unsigned)nq), dim3(256), 0, s, qbuf, X, d, ld, k, use_limit, max_d2, row_ids, id_base, D, I);
    return hipGetLastError();
}

hipError_t launch_ingest_rows(const float* src, long long n, int d, int ld, int normalize, float* dst, hipStream_t s) {
    long long blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ingest_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, n, d, ld, normalize, dst);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const float* X, int d, int ld, long long n, const long long* rows, long long ntotal, float* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n), dim3(256), 0, s, X, d, ld, n, rows, ntotal, out);
    return hipGetLastError();
}

}  // namespace rag
