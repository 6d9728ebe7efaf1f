// Score fusion and greedy MMR selection over a candidate set (a few hundred
// entries), each as ONE kernel launch of one workgroup.  All arithmetic is fp64
// with the reference's operation order and no fused multiply-add in the scoring
// formulas, so fused scores are bit-identical to the Python floats of
// vector_db/storage_manager.py:688-691 and vector_db/search_manager.py:1059-1075
// and MMR orderings match vector_db/mmr_diversifier.py:216-289.
//
// These kernels are latency-bound (microseconds); they exist so that the
// candidate set never has to leave the GPU between search, fusion and MMR.
#include <float.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "index.h"
#include "topk_keys.h"

namespace rag {

constexpr int FUSE_MAX_N = 4096;
constexpr int MMR_MAX_N = 4096;

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
__device__ void sort_entries(SortEntry* e, int n, int tid, int nthreads) {
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

// mode 0: score = a*wa + b*wb                          (content / context fusion)
// mode 1: score = wa*(a/max_a) + wb*(b/max_b)          (semantic / BM25 fusion)
struct FuseArgs {
    const double* a;
    const double* b;
    const uint8_t* has_a;
    const uint8_t* has_b;
    double wa, wb, max_a, max_b, threshold;
    int mode, use_threshold, n, npad, k;
    double* out_score;
    double* out_a;
    double* out_b;
    int* order;
    int* n_out;
};

__global__ __launch_bounds__(256) void fuse_kernel(FuseArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int& s_kept = *reinterpret_cast<int*>(smem);                      // [0..16): header
    SortEntry* e = reinterpret_cast<SortEntry*>(smem + 16);
    const int tid = threadIdx.x;
    if (tid == 0) s_kept = 0;
    __syncthreads();
    for (int i = tid; i < p.npad; i += 256) {
        SortEntry en;
        en.key = 0;            // below every real score
        en.idx = 0xFFFFFFFFu;
        en.pad = 0;
        if (i < p.n) {
            double sa, sb, sc;
            if (p.mode == 0) {
                sa = p.a[i];
                sb = p.b[i];
                sc = __dadd_rn(__dmul_rn(sa, p.wa), __dmul_rn(sb, p.wb));
            } else {
                sa = p.has_a[i] ? p.a[i] / p.max_a : 0.0;
                sb = p.has_b[i] ? p.b[i] / p.max_b : 0.0;
                sc = __dadd_rn(__dmul_rn(p.wa, sa), __dmul_rn(p.wb, sb));
                p.out_a[i] = sa;
                p.out_b[i] = sb;
            }
            p.out_score[i] = sc;
            const bool keep = !p.use_threshold || sc >= p.threshold;
            if (keep && sc == sc) {
                en.key = f64_ordered(sc);
                if (en.key == 0) en.key = 1;
                en.idx = (uint32_t)i;
                atomicAdd(&s_kept, 1);
            }
        }
        e[i] = en;
    }
    sort_entries(e, p.npad, tid, 256);
    const int kept = s_kept;
    const int m = kept < p.k ? kept : p.k;
    for (int i = tid; i < m; i += 256) p.order[i] = (int)e[i].idx;
    if (tid == 0) *p.n_out = m;
}

// ---------------------------------------------------------------------------
// MMR.  One workgroup of 1024 threads (16 waves); rounds separated by barriers.
struct MmrArgs {
    const void* emb;      // [n][d] fp32 or fp64 (host copy) ...
    const float* X;       // ... or rows of an index: X[rows[i]*ld + j]
    const long long* rows;
    int ld;
    int emb_is_f64;
    double* En;           // [n][d] workspace: rows as the similarity uses them
    const double* rel;    // [n]
    double lambda, threshold;
    int metric, n, d, max_results;
    int* out_idx;
    int* n_out;
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Per-launch state of the selection, in the workspace behind the outputs.
struct MmrState {
    double* maxsim;       // [n] largest similarity to any selected row so far
    int* alive;           // [n] 1 = not selected yet
    double* wg_best;      // [G] best marginal relevance of a workgroup's candidates this round
    int* wg_bidx;         // [G] ... and its candidate (0x7FFFFFFF: none alive)
    int* ctrl;            // [0] arrival counter of the round, [1] the row selected last
};

// rows as fp64; cosine: divided by their norm (a zero norm counts as 1).  One wave per row.
__global__ __launch_bounds__(64) void mmr_prepare_kernel(MmrArgs p, MmrState st) {
    const int lane = threadIdx.x, d = p.d;
    const int c = blockIdx.x;
    double ss = 0.0;
    for (int j = lane; j < d; j += 64) {
        double v;
        if (p.X) v = (double)p.X[p.rows[c] * (long long)p.ld + j];
        else if (p.emb_is_f64) v = static_cast<const double*>(p.emb)[(size_t)c * d + j];
        else v = (double)static_cast<const float*>(p.emb)[(size_t)c * d + j];
        p.En[(size_t)c * d + j] = v;
        ss = fma(v, v, ss);
    }
    if (p.metric == RAG_MMR_COSINE) {
        ss = wave_sum_f64(ss);
        double nrm = sqrt(ss);
        if (nrm == 0.0) nrm = 1.0;
        for (int j = lane; j < d; j += 64) p.En[(size_t)c * d + j] = p.En[(size_t)c * d + j] / nrm;
    }
    if (lane == 0) { st.maxsim[c] = -INFINITY; st.alive[c] = 1; }
}

__device__ __forceinline__ bool mmr_better(double ob, int oi, double best, int bidx) {
    // the reference starts from the first remaining candidate and moves on strict '>': the
    // lowest index among equal maxima wins
    return oi != 0x7FFFFFFF && (bidx == 0x7FFFFFFF || ob > best || (ob == best && oi < bidx));
}

// One selection round.  A single workgroup can only pull the candidate rows (n x d fp64, in L2)
// through one CU's memory path -- 200 x 768 took 32 us per round that way -- so a round is a launch
// of n / 8 small workgroups: each wave scores two candidates against the row selected last
// (one fma chain per lane over j = lane, lane + 64, ..., then the xor-shuffle tree), the
// workgroup posts its best marginal relevance, and the workgroup that arrives last reduces the
// posts and records the selection.  Rounds are ordered by the stream.
constexpr int MMR_WAVES = 4, MMR_PER_WAVE = 2, MMR_PER_WG = MMR_WAVES * MMR_PER_WAVE;
__global__ __launch_bounds__(MMR_WAVES * 64) void mmr_round_kernel(MmrArgs p, MmrState st, int r) {
    __shared__ double s_b[MMR_WAVES];
    __shared__ int s_i[MMR_WAVES];
    __shared__ int s_last_wg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.n, d = p.d, G = gridDim.x;
    const int last = r > 0 ? st.ctrl[1] : -1;
    const int c0 = blockIdx.x * MMR_PER_WG + wave * MMR_PER_WAVE;
    bool act[MMR_PER_WAVE];
    const double* ec[MMR_PER_WAVE];
    double acc[MMR_PER_WAVE];
#pragma unroll
    for (int u = 0; u < MMR_PER_WAVE; ++u) {
        const int c = c0 + u;
        act[u] = c < n && st.alive[c] != 0;
        ec[u] = p.En + (size_t)(act[u] ? c : 0) * d;
        acc[u] = 0.0;
    }
    double best = -INFINITY;
    int bidx = 0x7FFFFFFF;
    if (act[0] || act[1]) {
        if (last >= 0) {
            const double* es = p.En + (size_t)last * d;
            if (p.metric == RAG_MMR_EUCLIDEAN) {
#pragma unroll 4
                for (int j = lane; j < d; j += 64) {
                    const double e = es[j];
#pragma unroll
                    for (int u = 0; u < MMR_PER_WAVE; ++u) { const double df = ec[u][j] - e; acc[u] = fma(df, df, acc[u]); }
                }
            } else {
#pragma unroll 4
                for (int j = lane; j < d; j += 64) {
                    const double e = es[j];
#pragma unroll
                    for (int u = 0; u < MMR_PER_WAVE; ++u) acc[u] = fma(ec[u][j], e, acc[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < MMR_PER_WAVE; ++u) {
            const int c = c0 + u;
            double m = 0.0;
            if (last >= 0) {
                double v = wave_sum_f64(acc[u]);
                if (p.metric == RAG_MMR_EUCLIDEAN) v = 1.0 / (1.0 + sqrt(v));
                if (act[u]) {
                    m = st.maxsim[c];
                    if (v > m) { m = v; if (lane == 0) st.maxsim[c] = v; }
                    if (m < p.threshold) m = 0.0;
                }
            }
            if (act[u]) {
                const double v = __dsub_rn(__dmul_rn(p.lambda, p.rel[c]), __dmul_rn(1.0 - p.lambda, m));
                if (bidx == 0x7FFFFFFF) { bidx = c; best = (v > -INFINITY) ? v : -INFINITY; }
                else if (v > best) { best = v; bidx = c; }
            }
        }
    }
    if (lane == 0) { s_b[wave] = best; s_i[wave] = bidx; }
    __syncthreads();
    if (tid == 0) {
        double bb = s_b[0];
        int bi = s_i[0];
        for (int w = 1; w < MMR_WAVES; ++w)
            if (mmr_better(s_b[w], s_i[w], bb, bi)) { bb = s_b[w]; bi = s_i[w]; }
        st.wg_best[blockIdx.x] = bb;
        st.wg_bidx[blockIdx.x] = bi;
        __threadfence();                                   // the post is visible before the arrival
        s_last_wg = (atomicAdd(&st.ctrl[0], 1) == G - 1);
    }
    __syncthreads();
    if (!s_last_wg || wave != 0) return;
    __threadfence();
    best = -INFINITY;
    bidx = 0x7FFFFFFF;
    for (int g = lane; g < G; g += 64) {                   // increasing candidate index along g
        const unsigned long long ob = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(st.wg_best + g),
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int oi = __hip_atomic_load(st.wg_bidx + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mmr_better(__longlong_as_double((long long)ob), oi, best, bidx)) { best = __longlong_as_double((long long)ob); bidx = oi; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bidx, off, 64);
        if (mmr_better(ob, oi, best, bidx)) { best = ob; bidx = oi; }
    }
    if (lane == 0) {
        st.ctrl[0] = 0;
        st.ctrl[1] = bidx;
        if (bidx != 0x7FFFFFFF) {
            st.alive[bidx] = 0;
            p.out_idx[r] = bidx;
            *p.n_out = r + 1;
        }
    }
}

int ivf_reconstruct_rows(rag_index* h, long long n, const int64_t* rows, std::vector<long long>& storage_rows);   // ivf.hip

static thread_local Workspace g_fuse_ws[64];

static int current_device(int* dev) {
    RAG_HIP_CHECK(hipGetDevice(dev));
    return RAG_OK;
}

static int run_fuse(int mode, int64_t n, const double* a, const double* b, const uint8_t* has_a, const uint8_t* has_b,
                    double max_a, double max_b, double wa, double wb, int use_thr, double thr, int k, int32_t* order,
                    double* out_score, double* out_a, double* out_b, int32_t* n_out) {
    if (n < 0 || n > FUSE_MAX_N) return fail(RAG_ERR_INVALID, "fusion handles 0.." + std::to_string(FUSE_MAX_N) + " candidates");
    if (k < 0) return fail(RAG_ERR_INVALID, "k < 0");
    if (!n_out) return fail(RAG_ERR_INVALID, "n_out is NULL");
    *n_out = 0;
    if (n == 0) return RAG_OK;
    if (!a || !b || !order || !out_score) return fail(RAG_ERR_INVALID, "NULL buffer");
    if (mode == 1) {
        if (!has_a || !has_b || !out_a || !out_b) return fail(RAG_ERR_INVALID, "NULL buffer");
        if (max_a == 0.0 || max_b == 0.0) return fail(RAG_ERR_INVALID, "zero normalisation divisor");
    }
    int dev = 0;
    if (int rc = current_device(&dev)) return rc;
    Workspace& ws = g_fuse_ws[dev & 63];
    const size_t nb = (size_t)n * sizeof(double);
    const size_t nbp = (nb + 255) & ~(size_t)255;
    const size_t fb = ((size_t)n + 255) & ~(size_t)255;
    // layout: a | b | out_score | out_a | out_b | has_a | has_b | order | n_out
    const size_t total = 5 * nbp + 2 * fb + (((size_t)n * 4 + 255) & ~(size_t)255) + 256;
    if (int rc = ws.ensure(total)) return rc;
    char* base = static_cast<char*>(ws.ptr);
    double* da = reinterpret_cast<double*>(base);
    double* db = reinterpret_cast<double*>(base + nbp);
    double* ds = reinterpret_cast<double*>(base + 2 * nbp);
    double* dsa = reinterpret_cast<double*>(base + 3 * nbp);
    double* dsb = reinterpret_cast<double*>(base + 4 * nbp);
    uint8_t* dha = reinterpret_cast<uint8_t*>(base + 5 * nbp);
    uint8_t* dhb = dha + fb;
    int* dorder = reinterpret_cast<int*>(dhb + fb);
    int* dnout = reinterpret_cast<int*>(reinterpret_cast<char*>(dorder) + (((size_t)n * 4 + 255) & ~(size_t)255));
    hipStream_t s = nullptr;
    RAG_HIP_CHECK(hipMemcpyAsync(da, a, nb, hipMemcpyHostToDevice, s));
    RAG_HIP_CHECK(hipMemcpyAsync(db, b, nb, hipMemcpyHostToDevice, s));
    if (mode == 1) {
        RAG_HIP_CHECK(hipMemcpyAsync(dha, has_a, (size_t)n, hipMemcpyHostToDevice, s));
        RAG_HIP_CHECK(hipMemcpyAsync(dhb, has_b, (size_t)n, hipMemcpyHostToDevice, s));
    }
    FuseArgs p;
    p.a = da; p.b = db; p.has_a = dha; p.has_b = dhb;
    p.wa = wa; p.wb = wb; p.max_a = max_a; p.max_b = max_b; p.threshold = thr;
    p.mode = mode; p.use_threshold = use_thr; p.n = (int)n; p.npad = pow2_ceil((int)n < 2 ? 2 : (int)n); p.k = k;
    p.out_score = ds; p.out_a = dsa; p.out_b = dsb; p.order = dorder; p.n_out = dnout;
    const size_t lds = 16 + (size_t)p.npad * sizeof(SortEntry);
    if (lds > 48 * 1024)
        RAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fuse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fuse_kernel, dim3(1), dim3(256), lds, s, p);
    RAG_HIP_CHECK(hipGetLastError());
    // results come back in two copies: [order | n_out] (adjacent in the workspace) and the score arrays
    // (adjacent too); the stream is synchronised once
    const size_t ob = ((size_t)n * 4 + 255) & ~(size_t)255;
    std::vector<int32_t> host_order(ob / 4 + 1);
    RAG_HIP_CHECK(hipMemcpyAsync(host_order.data(), dorder, ob + sizeof(int), hipMemcpyDeviceToHost, s));
    if (mode == 1) {
        std::vector<double> host_scores(3 * nbp / sizeof(double));
        RAG_HIP_CHECK(hipMemcpyAsync(host_scores.data(), ds, 3 * nbp, hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipStreamSynchronize(s));
        memcpy(out_score, host_scores.data(), nb);
        memcpy(out_a, host_scores.data() + nbp / sizeof(double), nb);
        memcpy(out_b, host_scores.data() + 2 * nbp / sizeof(double), nb);
    } else {
        RAG_HIP_CHECK(hipMemcpyAsync(out_score, ds, nb, hipMemcpyDeviceToHost, s));
        RAG_HIP_CHECK(hipStreamSynchronize(s));
    }
    const int32_t m = host_order[ob / 4];
    if (m < 0 || (int64_t)m > n) return fail(RAG_ERR_HIP, "fusion kernel returned a bad count");
    if (m > 0) memcpy(order, host_order.data(), (size_t)m * sizeof(int32_t));
    *n_out = m;
    return RAG_OK;
}

static int run_mmr(int device, int64_t n, int d, const void* emb, int emb_is_f64, const float* X, int ld,
                   const long long* rows_host, const double* rel, double lambda, double thr, int metric, int max_results,
                   int32_t* out_idx, int32_t* n_out) {
    if (!n_out) return fail(RAG_ERR_INVALID, "n_out is NULL");
    *n_out = 0;
    if (n < 0 || n > MMR_MAX_N) return fail(RAG_ERR_INVALID, "MMR handles 0.." + std::to_string(MMR_MAX_N) + " candidates");
    if (n == 0 || max_results <= 0) return RAG_OK;
    if (d < 1) return fail(RAG_ERR_INVALID, "d < 1");
    if (!rel || !out_idx) return fail(RAG_ERR_INVALID, "NULL buffer");
    if (metric != RAG_MMR_COSINE && metric != RAG_MMR_DOT && metric != RAG_MMR_EUCLIDEAN) return fail(RAG_ERR_INVALID, "unknown MMR metric");
    if (!(lambda >= 0.0 && lambda <= 1.0)) return fail(RAG_ERR_INVALID, "lambda must be in [0, 1]");
    int prev = 0;
    RAG_HIP_CHECK(hipGetDevice(&prev));
    if (prev != device) RAG_HIP_CHECK(hipSetDevice(device));
    struct Restore { int p, d; ~Restore() { if (p != d) (void)hipSetDevice(p); } } restore{prev, device};
    Workspace& ws = g_fuse_ws[device & 63];
    const size_t esz = emb_is_f64 ? 8 : 4;
    const size_t eb = X ? (((size_t)n * 8 + 255) & ~(size_t)255) : (((size_t)n * d * esz + 255) & ~(size_t)255);
    const size_t enb = ((size_t)n * d * 8 + 255) & ~(size_t)255;
    const size_t rb = ((size_t)n * 8 + 255) & ~(size_t)255;
    const size_t ob = ((size_t)n * 4 + 255) & ~(size_t)255;
    const int G = (int)((n + MMR_PER_WG - 1) / MMR_PER_WG);
    const size_t gb = ((size_t)G * 8 + 255) & ~(size_t)255;
    // [rows or embeddings | En | rel | out_idx | n_out | maxsim | alive | wg_best | wg_bidx | ctrl]
    if (int rc = ws.ensure(eb + enb + rb + ob + 256 + rb + ob + 2 * gb + 256)) return rc;
    char* base = static_cast<char*>(ws.ptr);
    hipStream_t s = nullptr;
    MmrArgs p;
    p.emb = nullptr; p.X = nullptr; p.rows = nullptr; p.ld = ld; p.emb_is_f64 = emb_is_f64;
    if (X) {
        RAG_HIP_CHECK(hipMemcpyAsync(base, rows_host, (size_t)n * 8, hipMemcpyHostToDevice, s));
        p.X = X;
        p.rows = reinterpret_cast<const long long*>(base);
    } else {
        RAG_HIP_CHECK(hipMemcpyAsync(base, emb, (size_t)n * d * esz, hipMemcpyHostToDevice, s));
        p.emb = base;
    }
    p.En = reinterpret_cast<double*>(base + eb);
    double* drel = reinterpret_cast<double*>(base + eb + enb);
    int* dout = reinterpret_cast<int*>(base + eb + enb + rb);
    int* dn = reinterpret_cast<int*>(base + eb + enb + rb + ob);
    char* sbase = base + eb + enb + rb + ob + 256;
    MmrState st;
    st.maxsim = reinterpret_cast<double*>(sbase);
    st.alive = reinterpret_cast<int*>(sbase + rb);
    st.wg_best = reinterpret_cast<double*>(sbase + rb + ob);
    st.wg_bidx = reinterpret_cast<int*>(sbase + rb + ob + gb);
    st.ctrl = reinterpret_cast<int*>(sbase + rb + ob + 2 * gb);
    RAG_HIP_CHECK(hipMemcpyAsync(drel, rel, (size_t)n * 8, hipMemcpyHostToDevice, s));
    RAG_HIP_CHECK(hipMemsetAsync(dn, 0, 256, s));
    RAG_HIP_CHECK(hipMemsetAsync(st.ctrl, 0, 256, s));
    p.rel = drel; p.lambda = lambda; p.threshold = thr; p.metric = metric; p.n = (int)n; p.d = d;
    p.max_results = max_results; p.out_idx = dout; p.n_out = dn;
    hipLaunchKernelGGL(mmr_prepare_kernel, dim3((unsigned)n), dim3(64), 0, s, p, st);
    RAG_HIP_CHECK(hipGetLastError());
    const int rounds = max_results < (int)n ? max_results : (int)n;
    for (int r = 0; r < rounds; ++r)
        hipLaunchKernelGGL(mmr_round_kernel, dim3((unsigned)G), dim3(MMR_WAVES * 64), 0, s, p, st, r);
    RAG_HIP_CHECK(hipGetLastError());
    // [out_idx | n_out] are adjacent in the workspace: one copy, one synchronisation
    std::vector<int32_t> host_out(ob / 4 + 1);
    RAG_HIP_CHECK(hipMemcpyAsync(host_out.data(), dout, ob + sizeof(int), hipMemcpyDeviceToHost, s));
    RAG_HIP_CHECK(hipStreamSynchronize(s));
    const int32_t m = host_out[ob / 4];
    if (m < 0 || (int64_t)m > n) return fail(RAG_ERR_HIP, "MMR kernel returned a bad count");
    if (m > 0) memcpy(out_idx, host_out.data(), (size_t)m * sizeof(int32_t));
    *n_out = m;
    return RAG_OK;
}

}  // namespace rag

using namespace rag;

extern "C" {

int rag_fuse_weighted(int64_t n, const double* a, const double* b, double wa, double wb, int use_threshold,
                      double threshold, int k, int32_t* order, double* out_score, int32_t* n_out) {
    return run_fuse(0, n, a, b, nullptr, nullptr, 1.0, 1.0, wa, wb, use_threshold, threshold, k, order, out_score,
                    nullptr, nullptr, n_out);
}

int rag_fuse_maxnorm(int64_t n, const double* a, const double* b, const uint8_t* has_a, const uint8_t* has_b,
                     double max_a, double max_b, double wa, double wb, int k, int32_t* order, double* out_score,
                     double* out_a, double* out_b, int32_t* n_out) {
    return run_fuse(1, n, a, b, has_a, has_b, max_a, max_b, wa, wb, 0, 0.0, k, order, out_score, out_a, out_b, n_out);
}

int rag_mmr_select(int64_t n, int d, const void* emb, int emb_is_f64, const double* rel, double lambda,
                   double similarity_threshold, int metric, int max_results, int device, int32_t* out_idx,
                   int32_t* n_out) {
    if (n > 0 && !emb) return fail(RAG_ERR_INVALID, "emb is NULL");
    return run_mmr(device, n, d, emb, emb_is_f64, nullptr, 0, nullptr, rel, lambda, similarity_threshold, metric,
                   max_results, out_idx, n_out);
}

int rag_mmr_select_rows(rag_index_t* h, int64_t n, const int64_t* rows, const double* rel, double lambda,
                        double similarity_threshold, int metric, int max_results, int32_t* out_idx, int32_t* n_out) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    RAG_LOCK_INDEX(h);
    if (n > 0 && !rows) return fail(RAG_ERR_INVALID, "rows is NULL");
    for (int64_t i = 0; i < n; ++i)
        if (rows[i] < 0 || rows[i] >= h->n) return fail(RAG_ERR_INVALID, "row " + std::to_string(rows[i]) + " out of range");
    // an IVF index keeps its rows grouped by list: insertion position -> storage position
    std::vector<long long> storage_rows;
    const long long* where = reinterpret_cast<const long long*>(rows);
    if (h->kind == RAG_KIND_IVF && n > 0) {
        if (int rc = ivf_reconstruct_rows(h, n, rows, storage_rows)) return rc;
        where = storage_rows.data();
    }
    return run_mmr(h->device, n, h->dim, nullptr, 0, h->X, h->ld, where, rel, lambda,
                   similarity_threshold, metric, max_results, out_idx, n_out);
}

}  // extern "C"
