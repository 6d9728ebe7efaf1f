// Multi-embedding fan-out fusion on the device (vector_db/search_manager.py:877-932).
//
// The reference searches every document's collection with every query strategy, multiplies each
// hit's score by the strategy's weight, keeps per chunk the best weighted score (strict `>`, so the
// first strategy wins ties), and sorts the union of all documents by that score, stable, best first.
// Here the per-(strategy, document) hit lists are already in HBM (the output of the segment search);
// two kernels turn them into the final list without a trip to the host:
//
//   fanout_best_kernel   one workgroup per document: the candidates of its nq lists (<= 1024) in LDS;
//                        per row the best weighted score and the slot where the row shows up first
//                        (the position a Python dict would give its key); the document's entries
//                        ranked among themselves, the first `limit` written to the document's slab
//   fanout_rank_kernel   slabs merged in groups of up to 4096 entries, repeated until one slab is left;
//                        the last pass writes rows / scores / strategies.  A group is cut to the entries
//                        at or in front of a bound taken from the slabs' own order, and those are ranked
//                        by counting; a workgroup sort is the fall-back
//
// Scores are float32 hit scores widened to fp64 and multiplied by an fp64 weight, as Python does.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "index.h"
#include "sort_entries.h"

namespace rag {

static constexpr int FAN_SORT = 4096;          // entries one ranking workgroup holds (64 KB of LDS)
static constexpr int FAN_SURV = 1024;          // entries the ranking workgroup ranks by counting
static constexpr uint32_t FAN_NONE = 0xFFFFFFFFu;

struct FanoutArgs {
    const float* D;            // [nq][nseg][k] hit scores, best first, unfilled slots anywhere behind
    const long long* R;        // [nq][nseg][k] row of each hit, -1 = unfilled
    const double* weight;      // [nq]
    int nq, k, slab;           // slab = entries kept per document = min(limit, nq * k)
    long long nseg;
    SortEntry* out;            // [nseg][slab]
    double* win_score;         // [nseg][nq*k] by first-appearance slot
    long long* win_row;
    int* win_strategy;
};

__global__ __launch_bounds__(1024) void fanout_best_kernel(FanoutArgs p) {
    const int nthreads = blockDim.x;                                 // one candidate per thread where C <= 1024 allows
    extern __shared__ __align__(16) unsigned char fan_lds[];
    const int C = p.nq * p.k;
    const int CP = (C + 7) & ~7;                                        // arrays padded: the loops below read 8 at a time
    long long* rows = reinterpret_cast<long long*>(fan_lds);           // [CP], padding -2
    double* w = reinterpret_cast<double*>(rows + CP);                   // [CP]
    u64* key = reinterpret_cast<u64*>(w + CP);                          // [CP], 0 = no entry at this slot
    uint32_t* row_lo = reinterpret_cast<uint32_t*>(key + CP);           // [CP] low word of rows: what the common case compares
    uint32_t* key_hi = row_lo + CP;                                     // [CP] high word of key
    __shared__ int n_entries;
    const long long seg = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0) n_entries = 0;
    for (int c = tid; c < CP; c += nthreads) {
        long long r = -2;
        double wc = 0.0;
        if (c < C) {
            const int s = c / p.k, e = c - s * p.k;
            const size_t at = ((size_t)s * p.nseg + seg) * p.k + e;
            r = p.R[at];
            wc = r >= 0 ? __dmul_rn((double)p.D[at], p.weight[s]) : 0.0;
        }
        rows[c] = r;
        row_lo[c] = (uint32_t)r;
        w[c] = wc;
        key[c] = 0;
        key_hi[c] = 0;
    }
    __syncthreads();
    // The slot where a row shows up first owns the row's entry; its score is the best of the row's
    // candidates, the earliest candidate winning equal scores.  Both passes below are C x C compares per
    // document, so a pair costs 32-bit compares on words read eight at a time (broadcast LDS reads); the
    // full row / key is looked at only where the words agree.
    for (int c0 = 0; c0 < C; c0 += nthreads) {
        const int c = c0 + tid;
        bool owner = false;
        double bw = 0.0;
        int bj = 0;
        if (c < C && rows[c] >= 0) {
            const long long r = rows[c];
            const uint32_t rl = (uint32_t)r;
            int first = c;
            bw = w[c];
            bj = c;
            for (int j0 = 0; j0 < CP; j0 += 8) {
                uint32_t x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = row_lo[j0 + u];
                unsigned match = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) match |= (x[u] == rl ? 1u : 0u) << u;
                while (match) {
                    const int j = j0 + __builtin_ctz(match);
                    match &= match - 1;
                    if (rows[j] != r) continue;
                    first = j < first ? j : first;
                    const double wj = w[j];
                    if (j > c && wj > bw) { bw = wj; bj = j; }            // ascending j, strict: the earliest best wins
                }
            }
            owner = first == c;
        }
        if (owner) {
            const u64 kc = f64_ordered(bw);
            key[c] = kc;
            key_hi[c] = (uint32_t)(kc >> 32);
            const size_t slot = (size_t)seg * C + c;
            p.win_score[slot] = bw;
            p.win_row[slot] = rows[c];
            p.win_strategy[slot] = bj / p.k;
        }
        const unsigned long long b = __ballot(owner);
        if ((tid & 63) == 0 && b) atomicAdd(&n_entries, __popcll(b));
    }
    __syncthreads();
    // rank among the document's entries; the first `slab` go out in order.  High words decide unless
    // another entry shares this one's (then the full keys and the slot order are compared).
    for (int c = tid; c < C; c += nthreads) {
        const u64 kc = key[c];
        if (kc == 0) continue;
        const uint32_t hc = (uint32_t)(kc >> 32);
        int rank = 0, same = 0;
        for (int j0 = 0; j0 < CP; j0 += 8) {
            uint32_t x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = key_hi[j0 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                rank += x[u] > hc ? 1 : 0;
                same += x[u] == hc ? 1 : 0;
            }
        }
        if (same > 1) {
            rank = 0;
            for (int j = 0; j < C; ++j) {
                const u64 kj = key[j];
                rank += (kj > kc || (kj == kc && j < c)) ? 1 : 0;
            }
        }
        if (rank < p.slab) {
            SortEntry o;
            o.key = kc;
            o.idx = (uint32_t)((size_t)seg * C + c);
            o.pad = 0;
            p.out[(size_t)seg * p.slab + rank] = o;
        }
    }
    for (int t = n_entries + tid; t < p.slab; t += nthreads) {
        SortEntry o;
        o.key = 0;
        o.idx = FAN_NONE;
        o.pad = 0;
        p.out[(size_t)seg * p.slab + t] = o;
    }
}

struct FanoutRankArgs {
    const SortEntry* in;       // [n_in][in_slab]
    long long n_in;
    int in_slab, group;        // slabs per workgroup
    int np;                    // entries sorted per workgroup (power of two >= group * in_slab)
    int limit;
    SortEntry* out;            // [n_groups][limit] unless this is the last pass
    int last;
    const double* win_score;
    const long long* win_row;
    const int* win_strategy;
    long long* rows_out;       // [limit]
    double* scores_out;
    int* strategy_out;
    int* n_out;
};

__device__ __forceinline__ SortEntry fan_empty() {
    SortEntry x;
    x.key = 0;
    x.idx = FAN_NONE;
    x.pad = 0;
    return x;
}

// what the ranked list becomes: a slab for the next pass, or the caller's arrays
__device__ __forceinline__ void fan_emit(const FanoutRankArgs& p, int t, const SortEntry& x) {
    if (!p.last) {
        p.out[(size_t)blockIdx.x * p.limit + t] = x;
        return;
    }
    const bool ok = x.idx != FAN_NONE;
    p.rows_out[t] = ok ? p.win_row[x.idx] : -1;
    p.scores_out[t] = ok ? p.win_score[x.idx] : 0.0;
    p.strategy_out[t] = ok ? p.win_strategy[x.idx] : -1;
}

// Every input slab is sorted, so the entry at position pos of a slab has pos entries in front of it:
// with m slabs whose entry at pos is at or in front of a bound, m * (pos + 1) entries are.  The bound is
// the m-th of the slabs' entries at pos (ranked by counting); what is at or in front of it -- a few times
// `limit` entries at most on scores without mass ties -- is ranked by counting too.  Anything else (more
// survivors than FAN_SURV, more slabs than threads) sorts the whole group.
__global__ __launch_bounds__(1024) void fanout_rank_kernel(FanoutRankArgs p) {
    extern __shared__ __align__(16) unsigned char fan_lds[];
    SortEntry* e = reinterpret_cast<SortEntry*>(fan_lds);               // fall-back: the whole group
    SortEntry* heads = e;                                               // fast path: [S <= 1024]
    SortEntry* surv = e + 1024;                                         // [FAN_SURV]
    __shared__ SortEntry bound;
    __shared__ int n_surv;
    const int tid = threadIdx.x;
    const long long first = (long long)blockIdx.x * p.group;
    const int S = (int)min((long long)p.group, p.n_in - first);
    const int have = S * p.in_slab;
    const SortEntry* in = p.in + (size_t)first * p.in_slab;
    bool sorted_all = S > 1024;
    if (!sorted_all) {
        int pos = S >= p.limit ? 0 : (p.limit + S - 1) / S - 1;
        pos = pos < p.in_slab ? pos : p.in_slab - 1;
        int m = (p.limit + pos) / (pos + 1);
        m = m < S ? m : S;
        if (tid == 0) {
            bound = fan_empty();
            n_surv = 0;
        }
        if (tid < S) heads[tid] = in[(size_t)tid * p.in_slab + pos];
        __syncthreads();
        if (tid < S) {
            const SortEntry x = heads[tid];
            if (x.idx != FAN_NONE) {
                int rank = 0;
#pragma unroll 4
                for (int j = 0; j < S; ++j) rank += entry_before(heads[j], x) ? 1 : 0;
                if (rank == m - 1) bound = x;        // entries differ in (key, idx): one thread at most
            }
        }
        __syncthreads();
        const SortEntry bnd = bound;
        for (int t0 = 0; t0 < have; t0 += 1024) {
            const int t = t0 + tid;
            SortEntry x = fan_empty();
            if (t < have) x = in[t];
            const bool keep = x.idx != FAN_NONE && !entry_before(bnd, x);    // an empty bound keeps every entry
            const unsigned long long b = __ballot(keep);
            int base = 0;
            if ((tid & 63) == 0 && b) base = atomicAdd(&n_surv, __popcll(b));
            base = __shfl(base, 0);
            if (keep) {
                const int at = base + __popcll(b & ((1ull << (tid & 63)) - 1));
                if (at < FAN_SURV) surv[at] = x;
            }
        }
        __syncthreads();
        const int M = n_surv;
        if (M <= FAN_SURV) {
            for (int t = tid; t < M; t += 1024) {
                const SortEntry x = surv[t];
                int rank = 0;
#pragma unroll 4
                for (int j = 0; j < M; ++j) rank += entry_before(surv[j], x) ? 1 : 0;
                if (rank < p.limit) fan_emit(p, rank, x);
            }
            const int filled = M < p.limit ? M : p.limit;
            for (int t = filled + tid; t < p.limit; t += 1024) fan_emit(p, t, fan_empty());
            if (p.last && tid == 0) *p.n_out = filled;
            return;
        }
        sorted_all = true;
        __syncthreads();
    }
    for (int t = tid; t < p.np; t += 1024) e[t] = t < have ? in[t] : fan_empty();
    sort_entries(e, p.np, tid, 1024);
    for (int t = tid; t < p.limit; t += 1024) {
        const SortEntry x = t < p.np ? e[t] : fan_empty();
        fan_emit(p, t, x);
        if (p.last) {
            // entries come first after the sort: the last of them knows the count
            const bool next_ok = (t + 1 < p.limit && t + 1 < p.np) ? e[t + 1].idx != FAN_NONE : false;
            if (x.idx != FAN_NONE && !next_ok) *p.n_out = t + 1;
            if (t == 0 && x.idx == FAN_NONE) *p.n_out = 0;
        }
    }
}

static size_t fan_al(size_t b) { return (b + 255) & ~(size_t)255; }

static int fan_pow2(long long n) {
    int p = 64;
    while (p < n) p <<= 1;
    return p;
}

struct FanoutPlan {
    int C, slab;
    long long groups0;         // workgroups of the first ranking pass
    size_t o_weight, o_score, o_row, o_strategy, o_a, o_b, o_rows, o_scores, o_strat, o_n, total;
};

static bool fanout_plan(long long nq, long long nseg, int k, int limit, FanoutPlan* pl) {
    if (nq < 1 || nseg < 1 || k < 1 || limit < 1 || limit > 1024 || nq * k > 1024) return false;
    pl->C = (int)(nq * k);
    pl->slab = limit < pl->C ? limit : pl->C;
    if ((long long)pl->C * nseg >= 0x7FFFFFFFLL) return false;       // slots are 32-bit
    const int group0 = FAN_SORT / pl->slab;
    pl->groups0 = (nseg + group0 - 1) / group0;
    size_t o = 0;
    pl->o_weight = o;   o += fan_al((size_t)nq * sizeof(double));
    pl->o_score = o;    o += fan_al((size_t)nseg * pl->C * sizeof(double));
    pl->o_row = o;      o += fan_al((size_t)nseg * pl->C * sizeof(long long));
    pl->o_strategy = o; o += fan_al((size_t)nseg * pl->C * sizeof(int));
    pl->o_a = o;        o += fan_al(std::max((size_t)nseg * pl->slab, (size_t)pl->groups0 * limit) * sizeof(SortEntry));
    pl->o_b = o;        o += fan_al((size_t)pl->groups0 * limit * sizeof(SortEntry));
    pl->o_rows = o;     o += fan_al((size_t)limit * sizeof(long long));
    pl->o_scores = o;   o += fan_al((size_t)limit * sizeof(double));
    pl->o_strat = o;    o += fan_al((size_t)limit * sizeof(int));
    pl->o_n = o;        o += 256;
    pl->total = o;
    return true;
}

// bytes the fusion needs behind the hit lists; 0 = shape not supported
size_t fanout_workspace(long long nq, long long nseg, int k, int limit) {
    FanoutPlan pl;
    return fanout_plan(nq, nseg, k, limit, &pl) ? pl.total : 0;
}

// D / R: device hit lists [nq][nseg][k]; weights_host: nq doubles; ws: fanout_workspace() bytes of
// device memory, 256-byte aligned.  Leaves the ranked list in ws (pointers returned), nothing synchronised.
// weights_dev: the weights where the caller has already brought them to the device (nullptr: copied from weights_host here)
int fanout_fuse_enqueue(const float* D, const long long* R, const double* weights_host, const double* weights_dev, int nq,
                        long long nseg, int k, int limit, char* ws, long long** rows_out, double** scores_out, int** strat_out,
                        int** n_out, hipStream_t s) {
    FanoutPlan pl;
    if (!fanout_plan(nq, nseg, k, limit, &pl)) return fail(RAG_ERR_UNSUPPORTED, "fan-out fusion: shape not supported");
    if (nseg > 0x7FFFFFFFLL) return fail(RAG_ERR_INVALID, "too many segments");
    const double* weight = weights_dev;
    if (!weight) {
        double* w = reinterpret_cast<double*>(ws + pl.o_weight);
        RAG_HIP_CHECK(hipMemcpyAsync(w, weights_host, (size_t)nq * sizeof(double), hipMemcpyHostToDevice, s));
        weight = w;
    }
    FanoutArgs a = {};
    a.D = D;
    a.R = R;
    a.weight = weight;
    a.nq = nq;
    a.k = k;
    a.slab = pl.slab;
    a.nseg = nseg;
    a.out = reinterpret_cast<SortEntry*>(ws + pl.o_a);
    a.win_score = reinterpret_cast<double*>(ws + pl.o_score);
    a.win_row = reinterpret_cast<long long*>(ws + pl.o_row);
    a.win_strategy = reinterpret_cast<int*>(ws + pl.o_strategy);
    const size_t lds = (size_t)((pl.C + 7) & ~7) * (sizeof(long long) + sizeof(double) + sizeof(u64) + 2 * sizeof(uint32_t));
    hipLaunchKernelGGL(fanout_best_kernel, dim3((unsigned)nseg), dim3((unsigned)std::min(1024, (pl.C + 63) & ~63)), lds, s, a);
    RAG_HIP_CHECK(hipGetLastError());

    FanoutRankArgs r = {};
    r.limit = limit;
    r.win_score = a.win_score;
    r.win_row = a.win_row;
    r.win_strategy = a.win_strategy;
    r.rows_out = reinterpret_cast<long long*>(ws + pl.o_rows);
    r.scores_out = reinterpret_cast<double*>(ws + pl.o_scores);
    r.strategy_out = reinterpret_cast<int*>(ws + pl.o_strat);
    r.n_out = reinterpret_cast<int*>(ws + pl.o_n);
    SortEntry* bufs[2] = {reinterpret_cast<SortEntry*>(ws + pl.o_b), reinterpret_cast<SortEntry*>(ws + pl.o_a)};
    const SortEntry* in = a.out;
    long long n_in = nseg;
    int in_slab = pl.slab;
    for (int pass = 0;; ++pass) {
        const int group = FAN_SORT / in_slab;                // >= 4: every pass shrinks the slab count
        const long long groups = (n_in + group - 1) / group;
        r.in = in;
        r.n_in = n_in;
        r.in_slab = in_slab;
        r.group = group;
        r.np = fan_pow2(std::min<long long>(group, n_in) * in_slab);
        r.last = groups == 1;
        r.out = bufs[pass & 1];
        hipLaunchKernelGGL(fanout_rank_kernel, dim3((unsigned)groups), dim3(1024),
                           std::max<size_t>((size_t)r.np, 1024 + FAN_SURV) * sizeof(SortEntry), s, r);
        RAG_HIP_CHECK(hipGetLastError());
        if (r.last) break;
        in = r.out;
        n_in = groups;
        in_slab = limit;
    }
    *rows_out = r.rows_out;
    *scores_out = r.scores_out;
    *strat_out = r.strategy_out;
    *n_out = r.n_out;
    return RAG_OK;
}

}  // namespace rag
