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
//   fanout_rank_kernel   slabs merged in groups of up to 4096 entries by a workgroup sort, repeated
//                        until one slab is left; the last pass writes rows / scores / strategies
//
// Scores are float32 hit scores widened to fp64 and multiplied by an fp64 weight, as Python does.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "index.h"
#include "sort_entries.h"

namespace rag {

static constexpr int FAN_THREADS = 256;
static constexpr int FAN_SORT = 4096;          // entries one ranking workgroup holds (64 KB of LDS)
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

__global__ __launch_bounds__(FAN_THREADS) void fanout_best_kernel(FanoutArgs p) {
    extern __shared__ __align__(16) unsigned char fan_lds[];
    const int C = p.nq * p.k;
    long long* rows = reinterpret_cast<long long*>(fan_lds);           // [C]
    double* w = reinterpret_cast<double*>(rows + C);                    // [C]
    u64* key = reinterpret_cast<u64*>(w + C);                           // [C], 0 = no entry at this slot
    __shared__ int n_entries;
    const long long seg = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0) n_entries = 0;
    for (int c = tid; c < C; c += FAN_THREADS) {
        const int s = c / p.k, e = c - s * p.k;
        const size_t at = ((size_t)s * p.nseg + seg) * p.k + e;
        const long long r = p.R[at];
        rows[c] = r;
        w[c] = r >= 0 ? __dmul_rn((double)p.D[at], p.weight[s]) : 0.0;
        key[c] = 0;
    }
    __syncthreads();
    // the slot where a row shows up first owns the row's entry; its score is the best of the row's
    // candidates, the earliest candidate winning equal scores
    for (int c0 = 0; c0 < C; c0 += FAN_THREADS) {
        const int c = c0 + tid;
        bool owner = false;
        double bw = 0.0;
        int bj = 0;
        if (c < C && rows[c] >= 0) {
            const long long r = rows[c];
            owner = true;
            for (int j = 0; j < c; ++j)
                if (rows[j] == r) { owner = false; break; }
            if (owner) {
                bw = w[c];
                bj = c;
                for (int j = c + 1; j < C; ++j)
                    if (rows[j] == r && w[j] > bw) { bw = w[j]; bj = j; }
            }
        }
        if (owner) {
            key[c] = f64_ordered(bw);
            const size_t slot = (size_t)seg * C + c;
            p.win_score[slot] = bw;
            p.win_row[slot] = rows[c];
            p.win_strategy[slot] = bj / p.k;
        }
        const unsigned long long b = __ballot(owner);
        if ((tid & 63) == 0 && b) atomicAdd(&n_entries, __popcll(b));
    }
    __syncthreads();
    // rank among the document's entries; the first `slab` go out in order
    for (int c = tid; c < C; c += FAN_THREADS) {
        const u64 kc = key[c];
        if (kc == 0) continue;
        int rank = 0;
        for (int j = 0; j < C; ++j) {
            const u64 kj = key[j];
            rank += (kj > kc || (kj == kc && j < c)) ? 1 : 0;
        }
        if (rank < p.slab) {
            SortEntry o;
            o.key = kc;
            o.idx = (uint32_t)((size_t)seg * C + c);
            o.pad = 0;
            p.out[(size_t)seg * p.slab + rank] = o;
        }
    }
    for (int t = n_entries + tid; t < p.slab; t += FAN_THREADS) {
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

__global__ __launch_bounds__(1024) void fanout_rank_kernel(FanoutRankArgs p) {
    extern __shared__ __align__(16) unsigned char fan_lds[];
    SortEntry* e = reinterpret_cast<SortEntry*>(fan_lds);
    const int tid = threadIdx.x;
    const long long first = (long long)blockIdx.x * p.group;
    const long long have = min((long long)p.group, p.n_in - first) * p.in_slab;
    for (int t = tid; t < p.np; t += 1024) {
        SortEntry x;
        if (t < have) {
            x = p.in[(size_t)first * p.in_slab + t];
        } else {
            x.key = 0;
            x.idx = FAN_NONE;
            x.pad = 0;
        }
        e[t] = x;
    }
    sort_entries(e, p.np, tid, 1024);
    if (!p.last) {
        for (int t = tid; t < p.limit; t += 1024) {
            SortEntry x;
            if (t < p.np) {
                x = e[t];
            } else {
                x.key = 0;
                x.idx = FAN_NONE;
                x.pad = 0;
            }
            p.out[(size_t)blockIdx.x * p.limit + t] = x;
        }
        return;
    }
    for (int t = tid; t < p.limit; t += 1024) {
        const uint32_t slot = t < p.np ? e[t].idx : FAN_NONE;
        const bool ok = slot != FAN_NONE;
        p.rows_out[t] = ok ? p.win_row[slot] : -1;
        p.scores_out[t] = ok ? p.win_score[slot] : 0.0;
        p.strategy_out[t] = ok ? p.win_strategy[slot] : -1;
        // entries come first after the sort: the last of them knows the count
        const bool next_ok = (t + 1 < p.limit && t + 1 < p.np) ? e[t + 1].idx != FAN_NONE : false;
        if (ok && !next_ok) *p.n_out = t + 1;
        if (t == 0 && !ok) *p.n_out = 0;
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
int fanout_fuse_enqueue(const float* D, const long long* R, const double* weights_host, int nq, long long nseg, int k, int limit,
                        char* ws, long long** rows_out, double** scores_out, int** strat_out, int** n_out, hipStream_t s) {
    FanoutPlan pl;
    if (!fanout_plan(nq, nseg, k, limit, &pl)) return fail(RAG_ERR_UNSUPPORTED, "fan-out fusion: shape not supported");
    if (nseg > 0x7FFFFFFFLL) return fail(RAG_ERR_INVALID, "too many segments");
    double* weight = reinterpret_cast<double*>(ws + pl.o_weight);
    RAG_HIP_CHECK(hipMemcpyAsync(weight, weights_host, (size_t)nq * sizeof(double), hipMemcpyHostToDevice, s));
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
    const size_t lds = (size_t)pl.C * (sizeof(long long) + sizeof(double) + sizeof(u64));
    hipLaunchKernelGGL(fanout_best_kernel, dim3((unsigned)nseg), dim3(FAN_THREADS), lds, s, a);
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
        hipLaunchKernelGGL(fanout_rank_kernel, dim3((unsigned)groups), dim3(1024), (size_t)r.np * sizeof(SortEntry), s, r);
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
