// Few queries over a corpus that one selection workgroup can hold in its registers (1024 .. 32768 rows;
// 1 to 4 queries, k <= 1024): every row is scored without any selection in the scan, then one
// workgroup per query selects the k best of the n scores.  A scan whose waves see a few dozen rows each
// hands all of them to the merge anyway; this is two short launches instead.  Used for the coarse step
// of an IVF search (16384 centroids x 1024: 31.5 -> 23 us) and for small collections.  Above 32768 rows
// the selection workgroup becomes the bottleneck (100k x 384: 80 us against 47 us on the general path).
//
// Roofline: HBM / Infinity Cache -- bytes = n * ld * 4 per query group; at these sizes the launches'
// fixed costs weigh as much as the bytes.
//
// direct_scores_kernel: 16 lanes per row, 4 rows per wave instruction (4 x 256 contiguous bytes); the
// queries sit in LDS; per (row, query) an fma chain per lane, then a 4-step xor tree.  Scores are a
// k-ordered fp32 fma chain per (row, query), independent of the launch geometry.
#include <float.h>
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "index.h"
#include "topk_keys.h"

namespace rag {

typedef float f4c __attribute__((ext_vector_type(4)));
template <int NQ>
__global__ __launch_bounds__(256) void direct_scores_kernel(const float* C, int n, int ld, const float* Q, float* scores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f4c* q4 = reinterpret_cast<f4c*>(smem);                  // [NQ][ld4]
    const int ld4 = ld / 4;
    for (int i = threadIdx.x; i < NQ * ld4; i += 256) q4[i] = reinterpret_cast<const f4c*>(Q)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int sub = lane >> 4, c = lane & 15;
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * 256) >> 6;
    const f4c* C4 = reinterpret_cast<const f4c*>(C);
    for (long long r0 = wave * 4; r0 < n; r0 += nwaves * 4) {
        const long long r = r0 + sub;
        const long long rr = r < n ? r : n - 1;
        float acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
        // 16 loads (4 KB of the row group) go out together: one memory latency per 1024 floats of row width.
        // Cached loads: the table is re-read by every query.
        for (int i0 = c; i0 < ld4; i0 += 256) {
            f4c x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = i0 + 16 * u;
                x[u] = i < ld4 ? C4[rr * ld4 + i] : (f4c){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = i0 + 16 * u;
                if (i < ld4) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const f4c w = q4[q * ld4 + i];
                        acc[q] = fmaf(x[u].x, w.x, acc[q]);
                        acc[q] = fmaf(x[u].y, w.y, acc[q]);
                        acc[q] = fmaf(x[u].z, w.z, acc[q]);
                        acc[q] = fmaf(x[u].w, w.w, acc[q]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) acc[q] += __shfl_xor(acc[q], off, 64);
            if (c == 0 && r < n) scores[(size_t)q * n + r] = acc[q];
        }
    }
}

// direct_select_kernel: the k best (score descending, ties to the lower list) of n scores, one
// workgroup per query.  Keys (ordered score | ~list) are distinct, so the selection is a search for a
// bound with at least k and at most SEL_CAP keys at or above it, followed by a rank count among the
// survivors.  The bound comes from a 1024-bucket histogram over a key range that is narrowed to the bucket
// of the k-th key for as long as too many keys share it.  The first attempt bins only the 1024
// per-thread maxima ("heads": the k-th largest head has k keys at or above it) -- one LDS atomic
// per thread, spread over the upper tail, where binning all n keys piles most of them onto the few
// buckets of the bulk; if the heads do not settle it (fewer than k heads, or thousands of equal
// scores behind the bound) every key is binned.
constexpr int SEL_CAP = 4096;
constexpr int SEL_REG_MAX = 32;        // scores per thread held in registers: n <= 32768
struct SelectArgs {
    const float* scores;         // [nq][n]
    int n, k;
    const uint8_t* mask;         // nullptr or n bytes: rows that may be returned
    float min_score;
    const long long* row_ids;    // nullptr: id = id_base + row
    long long id_base;
    float* D;                    // [nq][k]
    long long* I;                // [nq][k]
    long long* rows_out;         // nullptr or [nq][k]: row position of every hit (-1 where I is -1)
    // a query's score row may hold several segments of n rows each (the last one shorter): workgroup b then serves
    // query b / segs, segment b % segs, and segment g of query q writes its list at g * out_stride + q * k
    int segs;                    // 1: one segment per query
    long long n_row;             // scores per query row (n when segs == 1)
    long long out_stride;        // hits between the lists of consecutive segments (0 when segs == 1)
    // nullptr, or per query a score no answer can lie below (floor[q * floor_stride]: the k-th best score of rows that
    // come EARLIER in the order rule, -FLT_MAX when there are fewer than k): scores under it are left out before anything
    // is binned, and when what is left fits the candidate array it is ranked directly
    const float* floor;
    long long floor_stride;
};

template <int SEL_REG>                  // scores per thread: n <= SEL_REG * 1024
__global__ __launch_bounds__(1024) void direct_select_kernel(SelectArgs p) {
    const int k = p.k;
    const int qi = (int)(blockIdx.x / (unsigned)p.segs), sg = (int)(blockIdx.x % (unsigned)p.segs);
    const long long seg_lo = (long long)sg * p.n;
    const int n = (int)(p.n_row - seg_lo < p.n ? p.n_row - seg_lo : p.n);
    const uint8_t* mask = p.mask ? p.mask + seg_lo : nullptr;
    const long long* row_ids = p.row_ids ? p.row_ids + seg_lo : nullptr;
    const long long id_base = p.id_base + seg_lo;
    const size_t obase = (size_t)sg * p.out_stride + (size_t)qi * k;
    float* D = p.D;
    long long* I = p.I;

    __shared__ int hist[1024];
    __shared__ u64 cand[SEL_CAP];
    __shared__ u64 s_lo, s_hi, s_bound;
    __shared__ int s_cnt, s_above, s_done, s_valid;
    const int tid = threadIdx.x, lane = tid & 63;
    const float* sc = p.scores + (size_t)qi * p.n_row + seg_lo;
    if (tid == 0) { s_lo = ~0ull; s_hi = 0ull; s_valid = 0; }
    __syncthreads();
    u64 head = KEY_EMPTY;                                          // this thread's best key
    // the thread's scores (rows tid, tid + 1024, ...) stay in registers for every later pass; a masked row
    // or one past the end is a NaN, which never becomes a key
    float v[SEL_REG];
    const float floor_sc = p.floor ? p.floor[(size_t)qi * p.floor_stride] : -INFINITY;
    int nvalid_w = 0;                                              // keys of this wave that may be in the answer
    {
        u64 mn = ~0ull, mx = 0ull;
#pragma unroll
        for (int u0 = 0; u0 < SEL_REG; u0 += 8) {                  // eight loads per memory latency
#pragma unroll
            for (int u = u0; u < u0 + 8; ++u) v[u] = tid + u * 1024 < n ? sc[tid + u * 1024] : NAN;
            if (mask) {
                unsigned char mk[8];
#pragma unroll
                for (int u = u0; u < u0 + 8; ++u) mk[u - u0] = tid + u * 1024 < n ? mask[tid + u * 1024] : 0;
#pragma unroll
                for (int u = u0; u < u0 + 8; ++u) v[u] = mk[u - u0] ? v[u] : NAN;
            }
#pragma unroll
            for (int u = u0; u < u0 + 8; ++u) {
                if (v[u] < floor_sc) v[u] = NAN;                   // k earlier rows score at least the floor
                const u64 key = make_key(v[u], (uint32_t)(tid + u * 1024));
                if (key != KEY_EMPTY) { mn = key < mn ? key : mn; mx = key > mx ? key : mx; }
            }
        }
        head = mx;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u64 omn = shfl_xor_u64(mn, off), omx = shfl_xor_u64(mx, off);
            mn = omn < mn ? omn : mn;
            mx = omx > mx ? omx : mx;
        }
        if (lane == 0 && mx != 0ull) { atomicMin(&s_lo, mn); atomicMax(&s_hi, mx); }
        if (p.floor) {
#pragma unroll
            for (int u = 0; u < SEL_REG; ++u) nvalid_w += v[u] == v[u] ? 1 : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nvalid_w += __shfl_xor(nvalid_w, off, 64);
            if (lane == 0 && nvalid_w) atomicAdd(&s_valid, nvalid_w);
        }
    }
    const int nheads = __syncthreads_count(head != KEY_EMPTY);
    const u64 glo = s_lo, ghi = s_hi;
    int cnum = 0;
    // behind a floor few keys are left: when they fit the candidate array all of them are candidates, no bound to look for
    const bool all_fit = p.floor && s_valid <= SEL_CAP;
    for (int mode = (nheads >= k && !all_fit ? 0 : 1); mode < 2 && ghi >= glo; ++mode) {
        const bool by_heads = mode == 0;
        __syncthreads();
        if (tid == 0) { s_cnt = 0; s_above = 0; s_done = 0; s_bound = all_fit ? 1ull : KEY_EMPTY; s_lo = glo; s_hi = ghi; }
        __syncthreads();
        u64 lo = glo, hi = ghi;
        // every pass keeps: keys above `hi` are selected for sure (s_above of them), the k-th key lies in [lo, hi]
        for (int pass = 0; pass < 8 && !all_fit; ++pass) {
            const int shift = (hi > lo) ? max(0, 64 - __clzll((long long)(hi - lo)) - 10) : 0;
            hist[tid] = 0;
            __syncthreads();
            if (by_heads) {
                if (head != KEY_EMPTY && head >= lo && head <= hi) atomicAdd(&hist[(int)((head - lo) >> shift)], 1);
            } else {
#pragma unroll
                for (int u = 0; u < SEL_REG; ++u) {
                    const u64 key = make_key(v[u], (uint32_t)(tid + u * 1024));
                    if (key != KEY_EMPTY && key >= lo && key <= hi) atomicAdd(&hist[(int)((key - lo) >> shift)], 1);
                }
            }
            __syncthreads();
            if (tid < 64) {
                const int need = k - s_above;
                int own = 0;
#pragma unroll
                for (int b = 0; b < 16; ++b) own += hist[tid * 16 + b];
                int suf = own;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int up = __shfl_down(suf, off, 64);
                    if (tid + off < 64) suf += up;
                }
                const int higher = suf - own;
                const int total = __shfl(suf, 0, 64);             // keys inside [lo, hi]
                if (total < need) {
                    if (tid == 0) { s_bound = lo; s_done = 1; }   // fewer keys than asked for: take them all
                } else if (higher < need && suf >= need) {
                    int cnt = higher, b = 15;
                    for (; b >= 0; --b) {
                        cnt += hist[tid * 16 + b];
                        if (cnt >= need) break;
                    }
                    const u64 edge = lo + ((u64)(tid * 16 + b) << shift);
                    // heads: any edge with k heads at or above it is a bound (how many keys stand behind it shows
                    // when they are collected); all keys: the survivors must fit the candidate array
                    if (by_heads || s_above + cnt <= SEL_CAP || shift == 0) { s_bound = edge; s_done = 1; }
                    else {
                        // too many keys share the bucket of the k-th: those above the bucket are in, look inside it
                        s_above += cnt - hist[tid * 16 + b];
                        s_lo = edge;
                        s_hi = edge + ((1ull << shift) - 1ull);
                    }
                }
            }
            __syncthreads();
            if (s_done) break;
            lo = s_lo;
            hi = s_hi;
            __syncthreads();
        }
        const u64 bound = s_bound;
        if (bound != KEY_EMPTY) {
#pragma unroll
            for (int u = 0; u < SEL_REG; ++u) {
                const u64 key = make_key(v[u], (uint32_t)(tid + u * 1024));
                if (key != KEY_EMPTY && key >= bound) {
                    const int pos = atomicAdd(&s_cnt, 1);
                    if (pos < SEL_CAP) cand[pos] = key;
                }
            }
        }
        __syncthreads();
        cnum = s_cnt;
        if (cnum <= SEL_CAP) break;                                // settled; otherwise bin every key
        cnum = 0;
    }
    for (int j = tid; j < k; j += 1024) {                     // slots that stay empty
        D[obase + j] = -FLT_MAX;
        I[obase + j] = -1;
        if (p.rows_out) p.rows_out[obase + j] = -1;
    }
    __syncthreads();
    for (int i = tid; i < cnum; i += 1024) {
        const u64 mine = cand[i];
        int above = 0;
#pragma unroll 8
        for (int j = 0; j < cnum; ++j) above += cand[j] > mine ? 1 : 0;
        const float sv = key_score(mine);
        if (above < k && sv >= p.min_score) {
            const long long row = (long long)key_row(mine);
            D[obase + above] = sv;
            I[obase + above] = row_ids ? row_ids[row] : id_base + row;
            if (p.rows_out) p.rows_out[obase + above] = seg_lo + row;
        }
    }
}

int direct_select_launch(rag_index* h, long long nq, int k, float min_score, const uint8_t* mask, const float* scores, float* D,
                         long long* I, hipStream_t s);
int direct_select_rows(long long nq, int n, int k, float min_score, const uint8_t* mask, const long long* row_ids, long long id_base,
                       long long* rows_out, const float* scores, float* D, long long* I, hipStream_t s);
int direct_select_segments(long long nq, long long n_row, long long seg_rows, int k, float min_score, const uint8_t* mask,
                           const long long* row_ids, long long id_base, const float* scores, float* D, long long* I, long long out_stride,
                           const float* floor, long long floor_stride, hipStream_t s);

bool direct_topk_supported(const rag_index* h, long long nq, int k) {
    static const int on = getenv("RAG_DIRECT_TOPK") ? atoi(getenv("RAG_DIRECT_TOPK")) : 1;          // measurement knob
    return on && h->kind == RAG_KIND_FLAT && nq >= 1 && nq <= 4 && h->n >= 1024 && h->n <= SEL_REG_MAX * 1024 &&
           k >= 1 && k <= 1024 && (size_t)4 * h->ld * 4 <= 48 * 1024;
}

size_t direct_topk_workspace(const rag_index* h, long long nq) { return ((size_t)nq * h->n * sizeof(float) + 255) & ~(size_t)255; }

int direct_topk_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask,
                       float* D, long long* I, float* scores, hipStream_t s) {
    const int n = (int)h->n;
    static const int max_waves = std::max(64, getenv("RAG_DIRECT_WAVES") ? atoi(getenv("RAG_DIRECT_WAVES")) : 4096);   // measurement knob
    const long long groups = (n + 3) / 4;                        // 4 rows per wave step
    const long long waves = std::min<long long>(groups, max_waves);
    const int blocks = (int)std::max<long long>(1, (waves + 3) / 4);
    const size_t lds = (size_t)nq * h->ld * sizeof(float);
    if (int rc = profile_begin(h, s)) return rc;
    if (nq == 1) hipLaunchKernelGGL(direct_scores_kernel<1>, dim3(blocks), dim3(256), lds, s, h->X, n, h->ld, qbuf, scores);
    else if (nq == 2) hipLaunchKernelGGL(direct_scores_kernel<2>, dim3(blocks), dim3(256), lds, s, h->X, n, h->ld, qbuf, scores);
    else if (nq == 3) hipLaunchKernelGGL(direct_scores_kernel<3>, dim3(blocks), dim3(256), lds, s, h->X, n, h->ld, qbuf, scores);
    else hipLaunchKernelGGL(direct_scores_kernel<4>, dim3(blocks), dim3(256), lds, s, h->X, n, h->ld, qbuf, scores);
    RAG_HIP_CHECK(hipGetLastError());
    if (int rc = profile_end(h, s)) return rc;
    if (int rc = direct_select_launch(h, nq, k, min_score, mask, scores, D, I, s)) return rc;
    h->last_path = "direct";
    return RAG_OK;
}

// the k best of every query's n scores ([nq][n], n <= 32768): one selection workgroup per query
int direct_select_launch(rag_index* h, long long nq, int k, float min_score, const uint8_t* mask, const float* scores, float* D,
                         long long* I, hipStream_t s) {
    return direct_select_rows(nq, (int)h->n, k, min_score, mask, h->ids_dev, h->id_base, h->merge_rows_out, scores, D, I, s);
}

// the same for any table of n <= 32768 scores per query: row r is reported as row_ids[r], or id_base + r
int direct_select_rows(long long nq, int n, int k, float min_score, const uint8_t* mask, const long long* row_ids, long long id_base,
                       long long* rows_out, const float* scores, float* D, long long* I, hipStream_t s) {
    SelectArgs p = {};
    p.scores = scores;
    p.n = n;
    p.k = k;
    p.mask = mask;
    p.min_score = min_score;
    p.row_ids = row_ids;
    p.id_base = id_base;
    p.D = D;
    p.I = I;
    p.rows_out = rows_out;
    p.segs = 1;
    p.n_row = n;
    p.out_stride = 0;
    if (n <= 8 * 1024) hipLaunchKernelGGL(direct_select_kernel<8>, dim3((unsigned)nq), dim3(1024), 0, s, p);
    else if (n <= 16 * 1024) hipLaunchKernelGGL(direct_select_kernel<16>, dim3((unsigned)nq), dim3(1024), 0, s, p);
    else hipLaunchKernelGGL(direct_select_kernel<SEL_REG_MAX>, dim3((unsigned)nq), dim3(1024), 0, s, p);
    RAG_HIP_CHECK(hipGetLastError());
    return RAG_OK;
}

// a score row of n_row scores per query, taken as segments of seg_rows = 32768 rows (the last one shorter; the caller
// states the segment length its output layout assumes): one workgroup per (query, segment); the list of segment g and
// query q starts at hit g * out_stride + q * k of D and I
int direct_select_segments(long long nq, long long n_row, long long seg_rows, int k, float min_score, const uint8_t* mask,
                           const long long* row_ids, long long id_base, const float* scores, float* D, long long* I, long long out_stride,
                           const float* floor, long long floor_stride, hipStream_t s) {
    const long long seg = (long long)SEL_REG_MAX * 1024;
    const long long segs = (n_row + seg - 1) / seg;
    if (seg_rows != seg || segs < 1 || nq * segs > 0x7fffffffLL) return fail(RAG_ERR_INVALID, "internal: segment selection grid");
    SelectArgs p = {};
    p.scores = scores;
    p.n = (int)seg;
    p.k = k;
    p.mask = mask;
    p.min_score = min_score;
    p.row_ids = row_ids;
    p.id_base = id_base;
    p.D = D;
    p.I = I;
    p.rows_out = nullptr;
    p.segs = (int)segs;
    p.n_row = n_row;
    p.out_stride = out_stride;
    p.floor = floor;
    p.floor_stride = floor_stride;
    hipLaunchKernelGGL(direct_select_kernel<SEL_REG_MAX>, dim3((unsigned)(nq * segs)), dim3(1024), 0, s, p);
    RAG_HIP_CHECK(hipGetLastError());
    return RAG_OK;
}

}  // namespace rag
