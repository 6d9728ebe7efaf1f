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
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "index.h"
#include "sort_entries.h"
#include "topk_keys.h"

namespace rag {

RAG_STAMPS_DEFINE(fuse)

constexpr int FUSE_MAX_N = 4096;
constexpr int MMR_MAX_N = 4096;

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
            }
        }
        e[i] = en;
        // npad is a power of two: a wave's lanes are in this trip together or (npad < 64) lane 0 is among them
        const int cnt = __popcll(__ballot(en.key != 0));
        if ((tid & 63) == 0 && cnt > 0) atomicAdd(&s_kept, cnt);
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
    const long long* pos;  // optional: row position -> storage position (IVF indexes keep rows grouped by list)
    const int* n_dev;      // optional: the candidate count lives on the device (<= n, which then sizes the grids)
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
__device__ __forceinline__ void mmr_prepare_body(const MmrArgs& p, const MmrState& st) {
    const int lane = threadIdx.x, d = p.d;
    const int c = blockIdx.x;
    if (p.n_dev && c >= *p.n_dev) return;
    long long xr = 0;
    if (p.X) {
        xr = p.rows[c];
        if (p.pos) xr = p.pos[xr];
    }
    double ss = 0.0;
    constexpr int PJ = 16;
    if (d <= 64 * PJ) {
        // the row fits the wave's registers (16 values per lane): read once, written once, already divided --
        // no trip through memory between the norm and the division.  Same sum order as the loop below.
        double v[PJ];
#pragma unroll
        for (int t = 0; t < PJ; ++t) {
            const int j = lane + 64 * t;
            v[t] = 0.0;
            if (j < d) {
                if (p.X) v[t] = (double)p.X[xr * (long long)p.ld + j];
                else if (p.emb_is_f64) v[t] = static_cast<const double*>(p.emb)[(size_t)c * d + j];
                else v[t] = (double)static_cast<const float*>(p.emb)[(size_t)c * d + j];
            }
        }
#pragma unroll
        for (int t = 0; t < PJ; ++t)
            if (lane + 64 * t < d) ss = fma(v[t], v[t], ss);
        double nrm = 1.0;
        if (p.metric == RAG_MMR_COSINE) {
            ss = wave_sum_f64(ss);
            nrm = sqrt(ss);
            if (nrm == 0.0) nrm = 1.0;
        }
#pragma unroll
        for (int t = 0; t < PJ; ++t) {
            const int j = lane + 64 * t;
            if (j < d) p.En[(size_t)c * d + j] = p.metric == RAG_MMR_COSINE ? v[t] / nrm : v[t];
        }
        if (lane == 0 && st.maxsim) { st.maxsim[c] = -INFINITY; st.alive[c] = 1; }
        return;
    }
    for (int j = lane; j < d; j += 64) {
        double v;
        if (p.X) v = (double)p.X[xr * (long long)p.ld + j];
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
    if (lane == 0 && st.maxsim) { st.maxsim[c] = -INFINITY; st.alive[c] = 1; }
}

__global__ __launch_bounds__(64) void mmr_prepare_kernel(MmrArgs p, MmrState st) { mmr_prepare_body(p, st); }

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

// ---------------------------------------------------------------------------
// Small candidate sets (n <= MMR_SIM_MAX_N, the reference's top-200): the whole selection in three
// launches instead of one per round.  mmr_simmat_kernel computes the similarity of every candidate
// pair with the arithmetic of mmr_round_kernel (per lane an fma chain over j = lane, lane + 64, ...,
// then the xor-shuffle tree), so S[c][s] is bit for bit the value a round would compute; the
// matrix is symmetric bit for bit (products and squared differences commute).  mmr_select_kernel
// then runs every round in ONE workgroup: a round is a row read of S (the row of the candidate
// selected last) and an argmax over the workgroup.
constexpr int MMR_SIM_MAX_N = 1024;
constexpr int SIM_TC = 8, SIM_TS = 8;      // candidates x selected rows per workgroup
// JL = ceil(d / 64) doubles of a row per lane, compile time (0: any d, rows re-read per pair).  The wave's two candidate rows
// sit in registers and every selected row's loads are independent of the arithmetic, so the eight
// rows of the tile stream through without a dependent-load chain.  Lanes past d hold zeros, which
// leave an fma chain unchanged.
template <int JL>
__device__ __forceinline__ void mmr_simmat_body(const double* En, const int* n_dev, int n_max, int d, int metric, double* S, int ldS) {
    const int n = n_dev ? *n_dev : n_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the matrix is symmetric bit for bit (a product and a squared difference do not depend on the order of
    // their operands, and both orders run the same chain and the same reduction tree): workgroup i computes tile
    // (bx, by), by <= bx, of the lower triangle and stores every value at (c, s) and at (s, c)
    int bx = (int)((sqrtf(8.f * (float)blockIdx.x + 1.f) - 1.f) * 0.5f);
    while ((bx + 1) * (bx + 2) / 2 <= (int)blockIdx.x) ++bx;
    while (bx * (bx + 1) / 2 > (int)blockIdx.x) --bx;
    const int by = (int)blockIdx.x - bx * (bx + 1) / 2;
    const int c0 = bx * SIM_TC + wave * 2;
    const int s0 = by * SIM_TS;
    const bool mirror = bx != by;
    if (c0 >= n || s0 >= n) return;
    const int c1 = c0 + 1 < n ? c0 + 1 : c0;
    const double* e0 = En + (size_t)c0 * d;
    const double* e1 = En + (size_t)c1 * d;
    if constexpr (JL > 0) {
        double r0[JL], r1[JL];
#pragma unroll
        for (int t = 0; t < JL; ++t) {
            const int j = lane + 64 * t;
            r0[t] = j < d ? e0[j] : 0.0;
            r1[t] = j < d ? e1[j] : 0.0;
        }
#pragma unroll 2
        for (int si = 0; si < SIM_TS; ++si) {
            const int sr = s0 + si < n ? s0 + si : n - 1;          // past the end: recompute the last row (not stored)
            const double* es = En + (size_t)sr * d;
            double e[JL];
#pragma unroll
            for (int t = 0; t < JL; ++t) {
                const int j = lane + 64 * t;
                e[t] = j < d ? es[j] : 0.0;
            }
            double a0 = 0.0, a1 = 0.0;
            if (metric == RAG_MMR_EUCLIDEAN) {
#pragma unroll
                for (int t = 0; t < JL; ++t) {
                    const double f0 = r0[t] - e[t], f1 = r1[t] - e[t];
                    a0 = fma(f0, f0, a0);
                    a1 = fma(f1, f1, a1);
                }
            } else {
#pragma unroll
                for (int t = 0; t < JL; ++t) {
                    a0 = fma(r0[t], e[t], a0);
                    a1 = fma(r1[t], e[t], a1);
                }
            }
            a0 = wave_sum_f64(a0);
            a1 = wave_sum_f64(a1);
            if (metric == RAG_MMR_EUCLIDEAN) { a0 = 1.0 / (1.0 + sqrt(a0)); a1 = 1.0 / (1.0 + sqrt(a1)); }
            if (lane == 0 && s0 + si < n) {
                S[(size_t)c0 * ldS + sr] = a0;
                if (mirror) S[(size_t)sr * ldS + c0] = a0;
                if (c0 + 1 < n) {
                    S[(size_t)(c0 + 1) * ldS + sr] = a1;
                    if (mirror) S[(size_t)sr * ldS + c0 + 1] = a1;
                }
            }
        }
    } else {
        for (int si = 0; si < SIM_TS; ++si) {
            const int sr = s0 + si;
            if (sr >= n) break;
            const double* es = En + (size_t)sr * d;
            double a0 = 0.0, a1 = 0.0;
            if (metric == RAG_MMR_EUCLIDEAN) {
#pragma unroll 4
                for (int j = lane; j < d; j += 64) {
                    const double e = es[j];
                    const double f0 = e0[j] - e, f1 = e1[j] - e;
                    a0 = fma(f0, f0, a0);
                    a1 = fma(f1, f1, a1);
                }
            } else {
#pragma unroll 4
                for (int j = lane; j < d; j += 64) {
                    const double e = es[j];
                    a0 = fma(e0[j], e, a0);
                    a1 = fma(e1[j], e, a1);
                }
            }
            a0 = wave_sum_f64(a0);
            a1 = wave_sum_f64(a1);
            if (metric == RAG_MMR_EUCLIDEAN) { a0 = 1.0 / (1.0 + sqrt(a0)); a1 = 1.0 / (1.0 + sqrt(a1)); }
            if (lane == 0) {
                S[(size_t)c0 * ldS + sr] = a0;
                if (mirror) S[(size_t)sr * ldS + c0] = a0;
                if (c0 + 1 < n) {
                    S[(size_t)(c0 + 1) * ldS + sr] = a1;
                    if (mirror) S[(size_t)sr * ldS + c0 + 1] = a1;
                }
            }
        }
    }
}

template <int JL>
__global__ __launch_bounds__(256) void mmr_simmat_kernel(const double* En, const int* n_dev, int n_max, int d, int metric,
                                                         double* S, int ldS) {
    mmr_simmat_body<JL>(En, n_dev, n_max, d, metric, S, ldS);
}

struct MmrSelArgs {
    const double* S;      // [n][ldS] pair similarities
    int ldS;
    const double* rel;    // [n] normalised relevance
    const int* n_dev;     // optional device-side candidate count (<= n_max)
    int n_max;
    double lambda, threshold;
    int max_results;
    int* out_idx;         // [max_results] candidate positions in selection order
    int* n_out;
    // optional epilogue (search -> fusion -> MMR in one enqueue): candidate position -> (id, fused score)
    const long long* cand_rows;
    const double* cand_score;
    const long long* row_ids;
    long long id_base;
    long long* out_ids;
    double* out_scores;
};

__device__ __forceinline__ void mmr_select_body(const MmrSelArgs& p) {
    __shared__ double s_b[16];
    __shared__ int s_i[16];
    __shared__ int s_sel;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.n_dev ? *p.n_dev : p.n_max;
    const int c = tid;
    bool alive = c < n;
    const double rel = alive ? p.rel[c] : 0.0;
    double maxsim = -INFINITY;
    const int rounds = p.max_results < n ? p.max_results : n;
    int sel = -1;
    for (int r = 0; r < rounds; ++r) {
        double best = -INFINITY;
        int bidx = 0x7FFFFFFF;
        if (alive) {
            double m = 0.0;
            if (sel >= 0) {
                const double v = p.S[(size_t)sel * p.ldS + c];
                if (v > maxsim) maxsim = v;
                m = maxsim;
                if (m < p.threshold) m = 0.0;
            }
            const double v = __dsub_rn(__dmul_rn(p.lambda, rel), __dmul_rn(1.0 - p.lambda, m));
            bidx = c;
            best = (v > -INFINITY) ? v : -INFINITY;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bidx, off, 64);
            if (mmr_better(ob, oi, best, bidx)) { best = ob; bidx = oi; }
        }
        if (lane == 0) { s_b[wave] = best; s_i[wave] = bidx; }
        __syncthreads();
        if (wave == 0) {
            best = lane < 16 ? s_b[lane] : -INFINITY;
            bidx = lane < 16 ? s_i[lane] : 0x7FFFFFFF;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bidx, off, 64);
                if (mmr_better(ob, oi, best, bidx)) { best = ob; bidx = oi; }
            }
            if (lane == 0) s_sel = bidx;
        }
        __syncthreads();
        sel = s_sel;
        if (sel == 0x7FFFFFFF) { sel = -1; break; }          // cannot happen while rounds <= n
        if (c == sel) alive = false;
        if (tid == 0) {
            p.out_idx[r] = sel;
            if (p.out_ids) {
                const long long row = p.cand_rows[sel];
                p.out_ids[r] = p.row_ids ? p.row_ids[row] : p.id_base + row;
                p.out_scores[r] = p.cand_score[sel];
            }
        }
    }
    if (tid == 0) *p.n_out = rounds;
}

__global__ __launch_bounds__(1024) void mmr_select_kernel(MmrSelArgs p) { mmr_select_body(p); }

// ---------------------------------------------------------------------------
// Hybrid fusion of a semantic hit list with a keyword (BM25) list, both already on the device
// (vector_db/search_manager.py:1023-1101 and the caller's sort + cut, :969-970), followed by the
// relevance normalisation MMR starts with (vector_db/mmr_diversifier.py:188-200).  Candidates are
// identified by row position.  One workgroup.
//   candidates   the semantic hits in order, then the keyword-only rows in order of first appearance
//   a repeated keyword row keeps the later score (dict semantics, :1036-1039); both maxima run over
//   every entry (:1042-1043); an empty side divides by 1.0 (:1045-1046)
//   a zero maximum is the reference's ZeroDivisionError path (:1098-1101): status = 1 and the
//   semantic list comes back with its raw scores
constexpr int HYB_MAX_SEM = 1024, HYB_MAX_BM = 1024;
struct HybridArgs {
    const float* D;              // [topn] semantic scores, best first
    const long long* rows;       // [topn] row positions, -1 = unfilled (a suffix)
    int topn;
    const long long* bm_rows;    // [nb] row positions of the keyword hits (outside [0, nrows): ignored)
    const double* bm_scores;     // [nb]
    int nb;
    long long nrows;
    double wa, wb;
    int limit;                   // fused list is cut to this many
    long long* f_rows;           // [limit] out: candidates best first
    double* f_score;             // [limit]
    double* f_sem;               // [limit] normalised semantic score (0 where absent)
    double* f_bm;                // [limit]
    unsigned char* f_flags;      // [limit] bit 0 = has semantic, bit 1 = has keyword
    double* rel;                 // [limit] min-max normalised fused score (all equal -> 1.0)
    int* n_fused;
    int* status;                 // 0 fused, 1 zero maximum (semantic list returned)
};

__device__ __forceinline__ void hybrid_fuse_body(const HybridArgs& p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int NC = p.topn + p.nb;                       // candidate slots
    int npad = 2;
    while (npad < NC) npad <<= 1;
    // [0, 64): counters and maxima; all LDS of the kernel is in the dynamic region (16-byte aligned base)
    int& s_nsem = *reinterpret_cast<int*>(smem);
    int& s_nbvalid = *reinterpret_cast<int*>(smem + 4);
    int& s_nonly = *reinterpret_cast<int*>(smem + 8);
    int& s_kept = *reinterpret_cast<int*>(smem + 12);
    unsigned long long& s_maxsem = *reinterpret_cast<unsigned long long*>(smem + 16);     // ordered fp64 bits
    unsigned long long& s_maxbm = *reinterpret_cast<unsigned long long*>(smem + 24);
    SortEntry* e = reinterpret_cast<SortEntry*>(smem + 64);                        // [npad]
    double* c_sem = reinterpret_cast<double*>(e + npad);                           // [NC]
    double* c_bm = c_sem + NC;                                                     // [NC]
    double* c_sc = c_bm + NC;                                                      // [NC]
    long long* c_row = reinterpret_cast<long long*>(c_sc + NC);                    // [NC]
    long long* b_row = c_row + NC;                                                 // [nb] keyword rows (-1: ignored)
    int* bm_first = reinterpret_cast<int*>(b_row + p.nb);                          // [nb]
    int* bm_last = bm_first + p.nb;                                                // [nb]
    int* bm_match = bm_last + p.nb;                                                // [nb]
    int* bm_before = bm_match + p.nb;                                              // [nb]
    unsigned char* c_fl = reinterpret_cast<unsigned char*>(bm_before + p.nb);      // [NC]
    int* order = reinterpret_cast<int*>(smem + ((reinterpret_cast<size_t>(c_fl + NC) - reinterpret_cast<size_t>(smem) + 3) & ~(size_t)3));   // [NC] candidate at each rank
    if (tid == 0) { s_nsem = 0; s_nbvalid = 0; s_nonly = 0; s_kept = 0; s_maxsem = 0ull; s_maxbm = 0ull; }
    RAG_STAMP(0);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    // counts and maxima are reduced inside a wave before they touch their LDS word: atomics of a whole
    // workgroup on one address run one after the other
    auto wave_max_u64 = [](unsigned long long v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = shfl_xor_u64(v, off);
            v = o > v ? o : v;
        }
        return v;
    };
    // semantic side
    for (int i0 = 0; i0 < p.topn; i0 += 1024) {
        const int i = i0 + tid;
        bool hit = false;
        unsigned long long ord = 0ull;
        if (i < p.topn) {
            const long long r = p.rows[i];
            c_fl[i] = 0;
            c_bm[i] = 0.0;
            c_row[i] = r;
            if (r >= 0) {
                const double sv = (double)p.D[i];
                c_sem[i] = sv;
                c_fl[i] = 1;
                hit = true;
                ord = f64_ordered(sv);
            }
        }
        const int cnt = __popcll(__ballot(hit));
        ord = wave_max_u64(ord);
        if (lane == 0 && cnt > 0) { atomicAdd(&s_nsem, cnt); atomicMax(&s_maxsem, ord); }
    }
    // keyword side: rows into LDS (an ignored row becomes -1), maxima over every valid entry.
    // bm_first[j] = lowest j2 with the same row, bm_last[j] = highest, bm_match[j] = lowest semantic hit with it
    for (int j0 = 0; j0 < p.nb; j0 += 1024) {
        const int j = j0 + tid;
        bool valid = false;
        unsigned long long ord = 0ull;
        if (j < p.nb) {
            long long r = p.bm_rows[j];
            if (r >= 0 && r < p.nrows) {
                valid = true;
                ord = f64_ordered(p.bm_scores[j]);
            } else {
                r = -1;
            }
            b_row[j] = r;
            bm_first[j] = j;
            bm_last[j] = j;
            bm_match[j] = 0x7FFFFFFF;
            bm_before[j] = 0;
        }
        const int cnt = __popcll(__ballot(valid));
        ord = wave_max_u64(ord);
        if (lane == 0 && cnt > 0) { atomicAdd(&s_nbvalid, cnt); atomicMax(&s_maxbm, ord); }
    }
    __syncthreads();
    RAG_STAMP(1);
    const int nsem = s_nsem;
    // every (keyword hit, keyword hit) and (keyword hit, semantic hit) pair: a wave takes a keyword hit j, its
    // lanes the other side; what a wave finds about j it reduces itself (no atomics, no divisions)
    auto wave_min_i32 = [](int v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
        return v;
    };
    for (int j = wave; j < p.nb; j += 16) {
        const long long rj = b_row[j];
        if (rj < 0) continue;                                  // wave-uniform
        int first = j, last = j, match = 0x7FFFFFFF;
        for (int j2 = lane; j2 < p.nb; j2 += 64)
            if (j2 != j && b_row[j2] == rj) {
                first = j2 < first ? j2 : first;
                last = j2 > last ? j2 : last;
            }
        for (int i = lane; i < nsem; i += 64)
            if (c_row[i] == rj) match = i < match ? i : match;
        // a keyword hit seldom repeats or coincides with a semantic hit: the defaults stand unless a lane found one
        if (__ballot(first != j || last != j || match != 0x7FFFFFFF) == 0ull) continue;
        first = wave_min_i32(first);
        last = -wave_min_i32(-last);
        match = wave_min_i32(match);
        if (lane == 0) { bm_first[j] = first; bm_last[j] = last; bm_match[j] = match; }
    }
    __syncthreads();
    RAG_STAMP(2);
    // keyword-only rows keep the order of their first appearance: place = keyword-only firsts before j
    for (int j = wave; j < p.nb; j += 16) {
        int before = 0;                                        // wave-uniform: counted with ballots
        for (int j0 = 0; j0 < j; j0 += 64) {
            const int j2 = j0 + lane;
            const bool only = j2 < j && b_row[j2] >= 0 && bm_first[j2] == j2 && bm_match[j2] == 0x7FFFFFFF;
            before += __popcll(__ballot(only));
        }
        if (lane == 0) bm_before[j] = before;
    }
    __syncthreads();
    {
        int only = 0;
        for (int j = tid; j < p.nb; j += 1024) {
            if (b_row[j] < 0 || bm_first[j] != j) continue;
            const double val = p.bm_scores[bm_last[j]];          // a repeated row keeps the later score
            int slot = bm_match[j];
            if (slot == 0x7FFFFFFF) {
                slot = nsem + bm_before[j];
                c_row[slot] = b_row[j];
                c_sem[slot] = 0.0;
                c_fl[slot] = 0;
                ++only;
            }
            c_bm[slot] = val;
            c_fl[slot] |= 2;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) only += __shfl_xor(only, off, 64);
        if (lane == 0 && only > 0) atomicAdd(&s_nonly, only);
    }
    __syncthreads();
    RAG_STAMP(3);
    const int ncand = nsem + s_nonly;
    auto unordered = [](unsigned long long o) { return __longlong_as_double((long long)((o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o)); };
    const double max_sem = nsem > 0 ? unordered(s_maxsem) : 1.0;
    const double max_bm = s_nbvalid > 0 ? unordered(s_maxbm) : 1.0;
    const bool zero_div = (nsem > 0 && max_sem == 0.0) || (s_nbvalid > 0 && max_bm == 0.0);
    const int pool = zero_div ? nsem : ncand;
    // up to 1024 slots the stable descending order comes from counting (below) and only the keys are kept;
    // larger sets are sorted
    const bool by_rank = npad <= 1024;
    u64* ek = reinterpret_cast<u64*>(e);                 // [npad] keys (0: not kept), rank path
    for (int i0 = 0; i0 < npad; i0 += 1024) {
        const int i = i0 + tid;
        SortEntry en;
        en.key = 0;
        en.idx = 0xFFFFFFFFu;
        en.pad = 0;
        if (i < pool) {
            double sc;
            if (zero_div) {
                sc = c_sem[i];
            } else {
                const double sa = (c_fl[i] & 1) ? c_sem[i] / max_sem : 0.0;
                const double sb = (c_fl[i] & 2) ? c_bm[i] / max_bm : 0.0;
                sc = __dadd_rn(__dmul_rn(p.wa, sa), __dmul_rn(p.wb, sb));
                c_sem[i] = sa;
                c_bm[i] = sb;
            }
            c_sc[i] = sc;
            if (sc == sc) {
                en.key = f64_ordered(sc);
                if (en.key == 0) en.key = 1;
                en.idx = (uint32_t)i;
            }
        }
        const int cnt = __popcll(__ballot(en.key != 0));
        if (lane == 0 && cnt > 0) atomicAdd(&s_kept, cnt);
        if (i < npad) {
            if (by_rank) ek[i] = en.key;
            else e[i] = en;
        }
    }
    __syncthreads();
    RAG_STAMP(4);
    const int m = s_kept < p.limit ? s_kept : p.limit;
    if (by_rank) {
        // (score, position) pairs are distinct, so "entries that sort before mine" is a rank; every lane reads
        // the same key at a time (an LDS broadcast) and no barrier separates the steps -- the 45 barrier steps
        // of a 512-entry bitonic sort took a third of this kernel's time
        // work is dealt so that all 16 waves take part: `parts` threads share a candidate, each counts over its
        // slice of the keys (keys above mine, and keys not below: equal scores are rare and settled afterwards)
        int* rank_gt = reinterpret_cast<int*>(ek + npad);      // [npad] behind the keys, inside the entry array
        int* rank_ge = rank_gt + npad;                         // [npad]
        for (int i = tid; i < 2 * npad; i += 1024) rank_gt[i] = 0;
        __syncthreads();
        const int lg = 31 - __clz(npad);                       // npad is a power of two <= 1024
        const int parts = 1024 >> lg;
        {
            const int i = tid & (npad - 1), part = tid >> lg;
            const u64 mine = i < pool ? ek[i] : 0;
            if (mine != 0) {
                const int j_lo = (pool * part) >> (10 - lg), j_hi = (pool * (part + 1)) >> (10 - lg);
                int gt = 0, ge = 0;
#pragma unroll 8
                for (int j = j_lo; j < j_hi; ++j) {
                    const u64 kj = ek[j];
                    gt += kj > mine ? 1 : 0;
                    ge += kj >= mine ? 1 : 0;
                }
                if (parts > 1) { atomicAdd(&rank_gt[i], gt); atomicAdd(&rank_ge[i], ge); }
                else { rank_gt[i] = gt; rank_ge[i] = ge; }
            }
        }
        __syncthreads();
        for (int i = tid; i < pool; i += 1024) {
            const u64 mine = ek[i];
            if (mine == 0) continue;
            int before = rank_gt[i];
            if (rank_ge[i] - before > 1)                       // equal scores: the earlier candidate first
                for (int j = 0; j < i; ++j) before += ek[j] == mine ? 1 : 0;
            if (before < m) order[before] = i;
        }
        __syncthreads();
    } else {
        sort_entries(e, npad, tid, 1024);
        for (int r = tid; r < m; r += 1024) order[r] = (int)e[r].idx;
        __syncthreads();
    }
    RAG_STAMP(5);
    const double hi = m > 0 ? c_sc[order[0]] : 0.0, lo = m > 0 ? c_sc[order[m - 1]] : 0.0;
    for (int r = tid; r < m; r += 1024) {
        const int i = order[r];
        const double sc = c_sc[i];
        p.f_rows[r] = c_row[i];
        p.f_score[r] = sc;
        p.f_sem[r] = c_sem[i];                      // zero maximum: the raw semantic score
        p.f_bm[r] = zero_div ? 0.0 : c_bm[i];
        p.f_flags[r] = zero_div ? 1 : c_fl[i];
        p.rel[r] = hi > lo ? (sc - lo) / (hi - lo) : 1.0;
    }
    if (tid == 0) { *p.n_fused = m; *p.status = zero_div ? 1 : 0; }
    RAG_STAMP(6);
}

__global__ __launch_bounds__(1024) void hybrid_fuse_kernel(HybridArgs p) { hybrid_fuse_body(p); }

// The same four stages for a batch of queries: blockIdx.y names the query, whose arguments (the structs the single-query
// launches take by value) sit in an array in device memory.  The bodies are the single-query ones.
struct TailArgs {
    HybridArgs f;
    MmrArgs p;
    MmrSelArgs sel;
    double* S;            // [n_max][n_max] pair matrix of this query
    int n_max;
};
__global__ __launch_bounds__(1024) void hybrid_fuse_batch_kernel(const TailArgs* t) { hybrid_fuse_body(t[blockIdx.y].f); }
__global__ __launch_bounds__(64) void mmr_prepare_batch_kernel(const TailArgs* t) {
    const MmrState none = {};
    mmr_prepare_body(t[blockIdx.y].p, none);
}
template <int JL>
__global__ __launch_bounds__(256) void mmr_simmat_batch_kernel(const TailArgs* t) {
    const TailArgs& a = t[blockIdx.y];
    mmr_simmat_body<JL>(a.p.En, a.p.n_dev, a.n_max, a.p.d, a.p.metric, a.S, a.n_max);
}
__global__ __launch_bounds__(1024) void mmr_select_batch_kernel(const TailArgs* t) { mmr_select_body(t[blockIdx.y].sel); }

static size_t hybrid_fuse_lds(int topn, int nb) {
    const size_t NC = (size_t)topn + nb;
    size_t npad = 2;
    while (npad < NC) npad <<= 1;
    return npad * sizeof(SortEntry) + NC * (3 * sizeof(double) + sizeof(long long)) + (size_t)nb * (4 * sizeof(int) + sizeof(long long)) + NC +
           NC * sizeof(int) + 4 + 64 + 16;
}


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
    Workspace& ws = thread_workspace(TWS_FUSE, dev);
    const size_t nb = (size_t)n * sizeof(double);
    const size_t nbp = (nb + 255) & ~(size_t)255;
    const size_t fb = ((size_t)n + 255) & ~(size_t)255;
    // layout: inputs a | b | has_a | has_b, then results out_score | out_a | out_b | order | n_out: one copy each way
    // through the thread's page-locked block
    const size_t ob = ((size_t)n * 4 + 255) & ~(size_t)255;
    const size_t in_bytes = 2 * nbp + 2 * fb, out_bytes = 3 * nbp + ob + 256;
    if (int rc = ws.ensure(in_bytes + out_bytes)) return rc;
    char* base = static_cast<char*>(ws.ptr);
    double* da = reinterpret_cast<double*>(base);
    double* db = reinterpret_cast<double*>(base + nbp);
    uint8_t* dha = reinterpret_cast<uint8_t*>(base + 2 * nbp);
    uint8_t* dhb = dha + fb;
    char* out_dev = base + in_bytes;
    double* ds = reinterpret_cast<double*>(out_dev);
    double* dsa = reinterpret_cast<double*>(out_dev + nbp);
    double* dsb = reinterpret_cast<double*>(out_dev + 2 * nbp);
    int* dorder = reinterpret_cast<int*>(out_dev + 3 * nbp);
    int* dnout = reinterpret_cast<int*>(out_dev + 3 * nbp + ob);
    char* pin = nullptr;
    if (int rc = thread_pinned(std::max(in_bytes, out_bytes), &pin)) return rc;
    hipStream_t s = nullptr;
    memcpy(pin, a, nb);
    memcpy(pin + nbp, b, nb);
    if (mode == 1) {
        memcpy(pin + 2 * nbp, has_a, (size_t)n);
        memcpy(pin + 2 * nbp + fb, has_b, (size_t)n);
    }
    RAG_HIP_CHECK(hipMemcpyAsync(base, pin, mode == 1 ? in_bytes : 2 * nbp, hipMemcpyHostToDevice, s));
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
    // (stream order: the input copy has left the block before the results arrive in it)
    RAG_HIP_CHECK(hipMemcpyAsync(pin, out_dev, 3 * nbp + ob + sizeof(int), hipMemcpyDeviceToHost, s));
    RAG_HIP_CHECK(hipStreamSynchronize(s));
    int32_t m = 0;
    memcpy(&m, pin + 3 * nbp + ob, sizeof(int32_t));
    if (m < 0 || (int64_t)m > n) return fail(RAG_ERR_HIP, "fusion kernel returned a bad count");
    memcpy(out_score, pin, nb);
    if (mode == 1) {
        memcpy(out_a, pin + nbp, nb);
        memcpy(out_b, pin + 2 * nbp, nb);
    }
    if (m > 0) memcpy(order, pin + 3 * nbp, (size_t)m * sizeof(int32_t));
    *n_out = m;
    return RAG_OK;
}

static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

static int env_int_m(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// candidate sets up to this size take the pair-matrix path (three launches); larger ones one launch per round
static int mmr_sim_limit() {
    static const int lim = std::min(MMR_SIM_MAX_N, std::max(0, env_int_m("RAG_MMR_SIM_MAX", 512)));
    return lim;
}

// enqueue the row preparation and the pair matrix S[n_max][n_max] of at most n_max candidates (p.En, p.n_dev set by the caller)
static void enqueue_pair_matrix(const MmrArgs& p, int n_max, double* S, hipStream_t s) {
    MmrState none = {};
    hipLaunchKernelGGL(mmr_prepare_kernel, dim3((unsigned)n_max), dim3(64), 0, s, p, none);
    static_assert(SIM_TC == SIM_TS, "square tiles: the lower triangle is enumerated tile by tile");
    const int tiles = (n_max + SIM_TC - 1) / SIM_TC;
    const dim3 grid((unsigned)(tiles * (tiles + 1) / 2));
    const int jl = (p.d + 63) / 64;
    if (jl <= 4) hipLaunchKernelGGL(mmr_simmat_kernel<4>, grid, dim3(256), 0, s, p.En, p.n_dev, n_max, p.d, p.metric, S, n_max);
    else if (jl <= 8) hipLaunchKernelGGL(mmr_simmat_kernel<8>, grid, dim3(256), 0, s, p.En, p.n_dev, n_max, p.d, p.metric, S, n_max);
    else if (jl <= 12) hipLaunchKernelGGL(mmr_simmat_kernel<12>, grid, dim3(256), 0, s, p.En, p.n_dev, n_max, p.d, p.metric, S, n_max);
    else if (jl <= 16) hipLaunchKernelGGL(mmr_simmat_kernel<16>, grid, dim3(256), 0, s, p.En, p.n_dev, n_max, p.d, p.metric, S, n_max);
    else hipLaunchKernelGGL(mmr_simmat_kernel<0>, grid, dim3(256), 0, s, p.En, p.n_dev, n_max, p.d, p.metric, S, n_max);
}

// enqueue prepare + pair matrix + selection for at most n_max candidates (p.En, p.rel, p.n_dev set by the caller)
static int enqueue_mmr_sim(const MmrArgs& p, int n_max, double* S, MmrSelArgs sel, hipStream_t s) {
    enqueue_pair_matrix(p, n_max, S, s);
    sel.S = S;
    sel.ldS = n_max;
    sel.rel = p.rel;
    sel.n_dev = p.n_dev;
    sel.n_max = n_max;
    sel.lambda = p.lambda;
    sel.threshold = p.threshold;
    sel.max_results = p.max_results;
    sel.out_idx = p.out_idx;
    sel.n_out = p.n_out;
    hipLaunchKernelGGL(mmr_select_kernel, dim3(1), dim3(1024), 0, s, sel);
    RAG_HIP_CHECK(hipGetLastError());
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
    Workspace& ws = thread_workspace(TWS_FUSE, device);
    const bool sim = n <= mmr_sim_limit();
    const size_t esz = emb_is_f64 ? 8 : 4;
    const size_t eb = X ? al256((size_t)n * 8) : al256((size_t)n * d * esz);
    const size_t enb = al256((size_t)n * d * 8);
    const size_t rb = al256((size_t)n * 8);
    const size_t ob = al256((size_t)n * 4);
    const int G = (int)((n + MMR_PER_WG - 1) / MMR_PER_WG);
    const size_t gb = al256((size_t)G * 8);
    const size_t simb = sim ? al256((size_t)n * n * 8) : 0;
    // [rows or embeddings | En | rel | out_idx | n_out | maxsim | alive | wg_best | wg_bidx | ctrl | pair matrix]
    if (int rc = ws.ensure(eb + enb + rb + ob + 256 + rb + ob + 2 * gb + 256 + simb)) return rc;
    char* base = static_cast<char*>(ws.ptr);
    hipStream_t s = nullptr;
    MmrArgs p = {};
    p.ld = ld; p.emb_is_f64 = emb_is_f64;
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
    double* S = reinterpret_cast<double*>(sbase + rb + ob + 2 * gb + 256);
    RAG_HIP_CHECK(hipMemcpyAsync(drel, rel, (size_t)n * 8, hipMemcpyHostToDevice, s));
    RAG_HIP_CHECK(hipMemsetAsync(dn, 0, 256, s));
    p.rel = drel; p.lambda = lambda; p.threshold = thr; p.metric = metric; p.n = (int)n; p.d = d;
    p.max_results = max_results; p.out_idx = dout; p.n_out = dn;
    if (sim) {
        if (int rc = enqueue_mmr_sim(p, (int)n, S, MmrSelArgs{}, s)) return rc;
    } else {
        RAG_HIP_CHECK(hipMemsetAsync(st.ctrl, 0, 256, s));
        hipLaunchKernelGGL(mmr_prepare_kernel, dim3((unsigned)n), dim3(64), 0, s, p, st);
        RAG_HIP_CHECK(hipGetLastError());
        const int rounds = max_results < (int)n ? max_results : (int)n;
        for (int r = 0; r < rounds; ++r)
            hipLaunchKernelGGL(mmr_round_kernel, dim3((unsigned)G), dim3(MMR_WAVES * 64), 0, s, p, st, r);
        RAG_HIP_CHECK(hipGetLastError());
    }
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

// ---------------------------------------------------------------------------
// search top-N -> hybrid fusion with a keyword list -> MMR -> final hits, enqueued on one stream
// with no host synchronisation in between (vector_db/search_manager.py:938-970 and :1620-1681).
int search_device_enqueue(rag_index* h, int64_t nq, const float* q_dev, int k, float min_score, int nprobe,
                          const uint8_t* row_mask_dev, float* D_dev, int64_t* I_dev, hipStream_t s);     // api.hip

struct HybridPlan {
    int topn, nb, L, M, d;
    size_t o_q, o_D, o_I, o_rows, o_bmr, o_bms, o_frows, o_fscore, o_fsem, o_fbm, o_rel, o_fflags, o_En, o_S, o_res, total;
    // o_res: [out_ids M*8 | out_scores M*8 | picked M*4 | n_out | n_fused | status] in one block
    size_t res_ids, res_scores, res_picked, res_counts, res_bytes;
};

static HybridPlan hybrid_plan(const rag_index* h, int topn, int nb, int fuse_limit, int max_results) {
    HybridPlan p = {};
    p.topn = topn; p.nb = nb; p.d = h->dim;
    p.L = std::min(fuse_limit, topn + nb);
    p.M = std::min(max_results, p.L);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
    p.o_q = take((size_t)h->dim * 4);
    p.o_D = take((size_t)topn * 4);
    p.o_I = take((size_t)topn * 8);
    p.o_rows = take((size_t)topn * 8);
    p.o_bmr = take((size_t)std::max(nb, 1) * 8);
    p.o_bms = take((size_t)std::max(nb, 1) * 8);
    p.o_frows = take((size_t)p.L * 8);
    p.o_fscore = take((size_t)p.L * 8);
    p.o_fsem = take((size_t)p.L * 8);
    p.o_fbm = take((size_t)p.L * 8);
    p.o_rel = take((size_t)p.L * 8);
    p.o_fflags = take((size_t)p.L);
    p.o_En = take((size_t)p.L * h->dim * 8);
    p.o_S = take((size_t)p.L * p.L * 8);
    p.res_ids = 0;
    p.res_scores = al256((size_t)p.M * 8);
    p.res_picked = p.res_scores + al256((size_t)p.M * 8);
    p.res_counts = p.res_picked + al256((size_t)p.M * 4);
    p.res_bytes = p.res_counts + 256;
    p.o_res = take(p.res_bytes);
    p.total = o;
    return p;
}

static int hybrid_check(rag_index* h, int topn, int64_t nb, int fuse_limit, int max_results, int metric, double lambda,
                        const rag_hybrid_out_t* out) {
    if (!h) return fail(RAG_ERR_INVALID, "index is NULL");
    if (!out || !out->ids || !out->scores || !out->n_out) return fail(RAG_ERR_INVALID, "out->ids / scores / n_out are required");
    if (h->metric == RAG_METRIC_L2) return fail(RAG_ERR_UNSUPPORTED, "the fusion divides similarities by their maximum: not available with the L2 metric");
    if (topn < 1 || topn > HYB_MAX_SEM) return fail(RAG_ERR_INVALID, "topn must be in [1, " + std::to_string(HYB_MAX_SEM) + "]");
    if (nb < 0 || nb > HYB_MAX_BM) return fail(RAG_ERR_INVALID, "at most " + std::to_string(HYB_MAX_BM) + " keyword hits");
    if (fuse_limit < 1 || fuse_limit > MMR_SIM_MAX_N) return fail(RAG_ERR_INVALID, "fuse_limit must be in [1, " + std::to_string(MMR_SIM_MAX_N) + "]");
    if (max_results < 1) return fail(RAG_ERR_INVALID, "max_results must be >= 1");
    if (metric != RAG_MMR_COSINE && metric != RAG_MMR_DOT && metric != RAG_MMR_EUCLIDEAN) return fail(RAG_ERR_INVALID, "unknown MMR metric");
    if (!(lambda >= 0.0 && lambda <= 1.0)) return fail(RAG_ERR_INVALID, "lambda must be in [0, 1]");
    return RAG_OK;
}

static int hybrid_tail(rag_index* h, const HybridPlan& pl, char* w, const float* D, const long long* rows, const long long* bm_rows_dev,
                       const double* bm_scores_dev, double w_sem, double w_bm, double lambda, double thr, int metric,
                       const rag_hybrid_out_t& dst, hipStream_t s);

// everything after the staging of inputs; `w` = the plan's workspace, bm lists already at their device addresses
static int hybrid_enqueue(rag_index* h, const HybridPlan& pl, char* w, const float* q_dev, float min_score, int nprobe,
                          const uint8_t* mask_dev, const long long* bm_rows_dev, const double* bm_scores_dev, double w_sem,
                          double w_bm, double lambda, double thr, int metric, const rag_hybrid_out_t& dst, hipStream_t s) {
    float* D = reinterpret_cast<float*>(w + pl.o_D);
    int64_t* I = reinterpret_cast<int64_t*>(w + pl.o_I);
    long long* rows = reinterpret_cast<long long*>(w + pl.o_rows);
    // 1. semantic top-N; the merge also reports the row position of every hit
    h->merge_rows_out = rows;
    const int rc = search_device_enqueue(h, 1, q_dev, pl.topn, min_score, nprobe, mask_dev, D, I, s);
    h->merge_rows_out = nullptr;
    if (rc) return rc;
    if (h->n == 0) RAG_HIP_CHECK(hipMemsetAsync(rows, 0xFF, (size_t)pl.topn * 8, s));     // no hits: every row -1
    return hybrid_tail(h, pl, w, D, rows, bm_rows_dev, bm_scores_dev, w_sem, w_bm, lambda, thr, metric, dst, s);
}

// the arguments of the fusion and MMR stages of ONE query given its semantic hits (scores D, row positions rows; -1 where a
// slot is empty); `w` = that query's set of the plan's buffers
static TailArgs hybrid_tail_args(rag_index* h, const HybridPlan& pl, char* w, const float* D, const long long* rows,
                                 const long long* bm_rows_dev, const double* bm_scores_dev, double w_sem, double w_bm, double lambda,
                                 double thr, int metric, const rag_hybrid_out_t& dst) {
    TailArgs t = {};
    // fusion + relevance normalisation
    HybridArgs& f = t.f;
    f.D = D; f.rows = rows; f.topn = pl.topn;
    f.bm_rows = bm_rows_dev; f.bm_scores = bm_scores_dev; f.nb = pl.nb; f.nrows = h->n;
    f.wa = w_sem; f.wb = w_bm; f.limit = pl.L;
    f.f_rows = dst.fused_rows ? reinterpret_cast<long long*>(dst.fused_rows) : reinterpret_cast<long long*>(w + pl.o_frows);
    f.f_score = dst.fused_scores ? dst.fused_scores : reinterpret_cast<double*>(w + pl.o_fscore);
    f.f_sem = dst.fused_sem ? dst.fused_sem : reinterpret_cast<double*>(w + pl.o_fsem);
    f.f_bm = dst.fused_bm ? dst.fused_bm : reinterpret_cast<double*>(w + pl.o_fbm);
    f.f_flags = reinterpret_cast<unsigned char*>(w + pl.o_fflags);
    f.rel = reinterpret_cast<double*>(w + pl.o_rel);
    int* counts = reinterpret_cast<int*>(w + pl.o_res + pl.res_counts);       // [n_out, n_fused, status]
    f.n_fused = dst.n_fused ? dst.n_fused : counts + 1;
    f.status = dst.status ? dst.status : counts + 2;
    // MMR over the fused candidates, vectors read from the index rows
    MmrArgs& p = t.p;
    p.X = h->X;
    p.rows = f.f_rows;
    p.pos = h->kind == RAG_KIND_IVF ? h->ivf.pos_dev : nullptr;
    p.ld = h->ld;
    p.n_dev = f.n_fused;
    p.En = reinterpret_cast<double*>(w + pl.o_En);
    p.rel = f.rel;
    p.lambda = lambda; p.threshold = thr; p.metric = metric; p.n = pl.L; p.d = h->dim;
    p.max_results = pl.M;
    p.out_idx = dst.picked ? dst.picked : reinterpret_cast<int*>(w + pl.o_res + pl.res_picked);
    p.n_out = dst.n_out;
    t.S = reinterpret_cast<double*>(w + pl.o_S);
    t.n_max = pl.L;
    MmrSelArgs& sel = t.sel;
    sel.S = t.S;
    sel.ldS = pl.L;
    sel.rel = p.rel;
    sel.n_dev = p.n_dev;
    sel.n_max = pl.L;
    sel.lambda = lambda;
    sel.threshold = thr;
    sel.max_results = pl.M;
    sel.out_idx = p.out_idx;
    sel.n_out = p.n_out;
    sel.cand_rows = f.f_rows;
    sel.cand_score = f.f_score;
    sel.row_ids = h->ids_dev;
    sel.id_base = h->id_base;
    sel.out_ids = reinterpret_cast<long long*>(dst.ids);
    sel.out_scores = dst.scores;
    return t;
}

static int hybrid_fuse_attr(rag_index* h, size_t lds) {
    static bool attr_set[64] = {};
    if (lds > 48 * 1024 && !attr_set[h->device & 63]) {
        RAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(hybrid_fuse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(hybrid_fuse_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[h->device & 63] = true;
    }
    return RAG_OK;
}

// fusion and MMR of ONE query: four small launches
static int hybrid_tail(rag_index* h, const HybridPlan& pl, char* w, const float* D, const long long* rows, const long long* bm_rows_dev,
                       const double* bm_scores_dev, double w_sem, double w_bm, double lambda, double thr, int metric,
                       const rag_hybrid_out_t& dst, hipStream_t s) {
    const TailArgs t = hybrid_tail_args(h, pl, w, D, rows, bm_rows_dev, bm_scores_dev, w_sem, w_bm, lambda, thr, metric, dst);
    const size_t lds = hybrid_fuse_lds(pl.topn, pl.nb);
    if (int rc = hybrid_fuse_attr(h, lds)) return rc;
    hipLaunchKernelGGL(hybrid_fuse_kernel, dim3(1), dim3(1024), lds, s, t.f);
    RAG_HIP_CHECK(hipGetLastError());
    return enqueue_mmr_sim(t.p, pl.L, t.S, t.sel, s);
}

// the same four stages for nq queries whose arguments sit in device memory (args_dev[0 .. nq)): one launch per stage
static int hybrid_tail_batch(rag_index* h, const HybridPlan& pl, const TailArgs* args_dev, int nq, hipStream_t s) {
    const size_t lds = hybrid_fuse_lds(pl.topn, pl.nb);
    if (int rc = hybrid_fuse_attr(h, lds)) return rc;
    const unsigned y = (unsigned)nq;
    hipLaunchKernelGGL(hybrid_fuse_batch_kernel, dim3(1, y), dim3(1024), lds, s, args_dev);
    hipLaunchKernelGGL(mmr_prepare_batch_kernel, dim3((unsigned)pl.L, y), dim3(64), 0, s, args_dev);
    const int tiles = (pl.L + SIM_TC - 1) / SIM_TC;
    const dim3 grid((unsigned)(tiles * (tiles + 1) / 2), y);
    const int jl = (h->dim + 63) / 64;
    if (jl <= 4) hipLaunchKernelGGL(mmr_simmat_batch_kernel<4>, grid, dim3(256), 0, s, args_dev);
    else if (jl <= 8) hipLaunchKernelGGL(mmr_simmat_batch_kernel<8>, grid, dim3(256), 0, s, args_dev);
    else if (jl <= 12) hipLaunchKernelGGL(mmr_simmat_batch_kernel<12>, grid, dim3(256), 0, s, args_dev);
    else if (jl <= 16) hipLaunchKernelGGL(mmr_simmat_batch_kernel<16>, grid, dim3(256), 0, s, args_dev);
    else hipLaunchKernelGGL(mmr_simmat_batch_kernel<0>, grid, dim3(256), 0, s, args_dev);
    hipLaunchKernelGGL(mmr_select_batch_kernel, dim3(1, y), dim3(1024), 0, s, args_dev);
    RAG_HIP_CHECK(hipGetLastError());
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

int rag_similarity_matrix(int64_t n, int d, const void* emb, int emb_is_f64, int metric, int device, double* S_out) {
    if (n < 0 || n > MMR_MAX_N) return fail(RAG_ERR_INVALID, "the similarity matrix handles 0.." + std::to_string(MMR_MAX_N) + " rows");
    if (n == 0) return RAG_OK;
    if (d < 1) return fail(RAG_ERR_INVALID, "d < 1");
    if (!emb || !S_out) return fail(RAG_ERR_INVALID, "NULL buffer");
    if (metric != RAG_MMR_COSINE && metric != RAG_MMR_DOT && metric != RAG_MMR_EUCLIDEAN) return fail(RAG_ERR_INVALID, "unknown MMR metric");
    int prev = 0;
    RAG_HIP_CHECK(hipGetDevice(&prev));
    if (prev != device) RAG_HIP_CHECK(hipSetDevice(device));
    struct Restore { int p, d; ~Restore() { if (p != d) (void)hipSetDevice(p); } } restore{prev, device};
    Workspace& ws = thread_workspace(TWS_FUSE, device);
    const size_t esz = emb_is_f64 ? 8 : 4;
    const size_t eb = al256((size_t)n * d * esz), enb = al256((size_t)n * d * 8), sb = al256((size_t)n * n * 8);
    // [embeddings | rows as the similarity uses them | pair matrix]
    if (int rc = ws.ensure(eb + enb + sb)) return rc;
    char* base = static_cast<char*>(ws.ptr);
    hipStream_t s = nullptr;
    RAG_HIP_CHECK(hipMemcpyAsync(base, emb, (size_t)n * d * esz, hipMemcpyHostToDevice, s));
    MmrArgs p = {};
    p.emb = base; p.emb_is_f64 = emb_is_f64; p.En = reinterpret_cast<double*>(base + eb);
    p.metric = metric; p.n = (int)n; p.d = d;
    double* S = reinterpret_cast<double*>(base + eb + enb);
    enqueue_pair_matrix(p, (int)n, S, s);
    RAG_HIP_CHECK(hipGetLastError());
    RAG_HIP_CHECK(hipMemcpyAsync(S_out, S, (size_t)n * n * 8, hipMemcpyDeviceToHost, s));
    RAG_HIP_CHECK(hipStreamSynchronize(s));
    return RAG_OK;
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

int rag_index_hybrid_mmr_device(rag_index_t* h, const float* q_dev, int topn, float min_score, int nprobe,
                                const uint8_t* row_mask_dev, int64_t nb, const int64_t* bm25_rows_dev,
                                const double* bm25_scores_dev, double w_sem, double w_bm25, int fuse_limit, double mmr_lambda,
                                double mmr_threshold, int mmr_metric, int max_results, const rag_hybrid_out_t* out_dev,
                                void* stream) {
    if (int rc = hybrid_check(h, topn, nb, fuse_limit, max_results, mmr_metric, mmr_lambda, out_dev)) return rc;
    if (!q_dev || (nb > 0 && (!bm25_rows_dev || !bm25_scores_dev))) return fail(RAG_ERR_INVALID, "NULL buffer");
    if (min_score != min_score) return fail(RAG_ERR_INVALID, "min_score is NaN");
    RAG_LOCK_INDEX(h);
    int prev = 0;
    RAG_HIP_CHECK(hipGetDevice(&prev));
    if (prev != h->device) RAG_HIP_CHECK(hipSetDevice(h->device));
    struct Restore { int p, d; ~Restore() { if (p != d) (void)hipSetDevice(p); } } restore{prev, h->device};
    const HybridPlan pl = hybrid_plan(h, topn, (int)nb, fuse_limit, max_results);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = ws_acquire(h, s)) return rc;
    if (int rc = h->ws3.ensure(pl.total)) return rc;
    if (int rc = hybrid_enqueue(h, pl, static_cast<char*>(h->ws3.ptr), q_dev, min_score, nprobe, row_mask_dev,
                                reinterpret_cast<const long long*>(bm25_rows_dev), bm25_scores_dev, w_sem, w_bm25, mmr_lambda,
                                mmr_threshold, mmr_metric, *out_dev, s))
        return rc;
    return ws_release(h, s);
}

int rag_index_hybrid_mmr_batch_device(rag_index_t* h, int64_t nq, const float* q_dev, int topn, float min_score, int nprobe,
                                      const uint8_t* row_mask_dev, int64_t nb, const int64_t* bm25_rows_dev,
                                      const double* bm25_scores_dev, double w_sem, double w_bm25, int fuse_limit, double mmr_lambda,
                                      double mmr_threshold, int mmr_metric, int max_results, int64_t* ids_dev, double* scores_dev,
                                      int32_t* n_out_dev, void* stream) {
    rag_hybrid_out_t probe = {};
    probe.ids = ids_dev; probe.scores = scores_dev; probe.n_out = n_out_dev;
    if (int rc = hybrid_check(h, topn, nb, fuse_limit, max_results, mmr_metric, mmr_lambda, &probe)) return rc;
    if (nq < 0) return fail(RAG_ERR_INVALID, "nq < 0");
    if (nq == 0) return RAG_OK;
    if (!q_dev || (nb > 0 && (!bm25_rows_dev || !bm25_scores_dev))) return fail(RAG_ERR_INVALID, "NULL buffer");
    if (min_score != min_score) return fail(RAG_ERR_INVALID, "min_score is NaN");
    RAG_LOCK_INDEX(h);
    int prev = 0;
    RAG_HIP_CHECK(hipGetDevice(&prev));
    if (prev != h->device) RAG_HIP_CHECK(hipSetDevice(h->device));
    struct Restore { int p, d; ~Restore() { if (p != d) (void)hipSetDevice(p); } } restore{prev, h->device};
    const HybridPlan pl = hybrid_plan(h, topn, (int)nb, fuse_limit, max_results);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = ws_acquire(h, s)) return rc;
    // fusion and MMR run for a chunk of queries at a time (one launch per stage, blockIdx.y = query), every query of a chunk
    // with its own set of the plan's buffers; the per-query arguments hold addresses only, nothing the search produces
    const size_t tail_bytes = al256(pl.total);
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nq, 512), (int64_t)(((size_t)2 << 30) / tail_bytes)));
    const size_t ab = al256((size_t)nq * sizeof(TailArgs));
    const size_t db = al256((size_t)nq * topn * sizeof(float)), ib = al256((size_t)nq * topn * sizeof(long long));
    if (int rc = h->ws3.ensure((size_t)chunk * tail_bytes + ab + db + ib)) return rc;
    char* w = static_cast<char*>(h->ws3.ptr);
    TailArgs* args_dev = reinterpret_cast<TailArgs*>(w + (size_t)chunk * tail_bytes);
    float* Dall = reinterpret_cast<float*>(w + (size_t)chunk * tail_bytes + ab);
    long long* Rall = reinterpret_cast<long long*>(w + (size_t)chunk * tail_bytes + ab + db);
    const int M = pl.M;
    // 1. the semantic top-N of all queries as ONE search.  It leaves row positions (the id table and base are set aside for
    //    it: the index is locked and they are read when a launch is built); ids are named when the hits are final
    long long* const ids = h->ids_dev;
    const long long base = h->id_base;
    h->ids_dev = nullptr;
    h->id_base = 0;
    const int rc = search_device_enqueue(h, nq, q_dev, topn, min_score, nprobe, row_mask_dev, Dall, reinterpret_cast<int64_t*>(Rall), s);
    h->ids_dev = ids;
    h->id_base = base;
    if (rc) return rc;
    // 2. the per-query kernel arguments, built only now: an IVF search may have re-packed the rows (a new X, a new
    //    insertion -> storage map), and the arguments hold those addresses
    {
        // pinned staging that belongs to the index; the event says when the last copy out of it has been made
        if (h->tail_args_cap < (size_t)nq * sizeof(TailArgs)) {
            if (h->tail_args_event) RAG_HIP_CHECK(hipEventSynchronize(h->tail_args_event));
            if (h->tail_args_host) RAG_HIP_CHECK(hipHostFree(h->tail_args_host));
            h->tail_args_host = nullptr;
            h->tail_args_cap = 0;
            RAG_HIP_CHECK(hipHostMalloc(&h->tail_args_host, (size_t)nq * sizeof(TailArgs), hipHostMallocDefault));
            h->tail_args_cap = (size_t)nq * sizeof(TailArgs);
        }
        if (!h->tail_args_event) RAG_HIP_CHECK(hipEventCreateWithFlags(&h->tail_args_event, hipEventDisableTiming));
        else RAG_HIP_CHECK(hipEventSynchronize(h->tail_args_event));
        TailArgs* host = static_cast<TailArgs*>(h->tail_args_host);
        for (int64_t i = 0; i < nq; ++i) {
            rag_hybrid_out_t dst = {};
            dst.ids = ids_dev + (size_t)i * M;
            dst.scores = scores_dev + (size_t)i * M;
            dst.n_out = n_out_dev + i;
            host[i] = hybrid_tail_args(h, pl, w + (size_t)(i % chunk) * tail_bytes, Dall + (size_t)i * topn, Rall + (size_t)i * topn,
                                       nb > 0 ? reinterpret_cast<const long long*>(bm25_rows_dev) + (size_t)i * nb : nullptr,
                                       nb > 0 ? bm25_scores_dev + (size_t)i * nb : nullptr, w_sem, w_bm25, mmr_lambda, mmr_threshold,
                                       mmr_metric, dst);
        }
        RAG_HIP_CHECK(hipMemcpyAsync(args_dev, host, (size_t)nq * sizeof(TailArgs), hipMemcpyHostToDevice, s));
        RAG_HIP_CHECK(hipEventRecord(h->tail_args_event, s));
    }
    // 3. fusion and MMR behind it on the same stream, a chunk of queries per launch
    RAG_HIP_CHECK(hipMemsetAsync(ids_dev, 0xFF, (size_t)nq * M * sizeof(int64_t), s));        // slots past a query's count: -1
    RAG_HIP_CHECK(hipMemsetAsync(n_out_dev, 0, (size_t)nq * sizeof(int32_t), s));
    for (int64_t q0 = 0; q0 < nq; q0 += chunk)
        if (int rc2 = hybrid_tail_batch(h, pl, args_dev + q0, (int)std::min<int64_t>(chunk, nq - q0), s)) return rc2;
    return ws_release(h, s);
}

int rag_index_hybrid_mmr(rag_index_t* h, const float* q, int topn, float min_score, int nprobe, const uint8_t* row_mask,
                         int64_t nb, const int64_t* bm25_rows, const double* bm25_scores, double w_sem, double w_bm25,
                         int fuse_limit, double mmr_lambda, double mmr_threshold, int mmr_metric, int max_results,
                         const rag_hybrid_out_t* out) {
    if (int rc = hybrid_check(h, topn, nb, fuse_limit, max_results, mmr_metric, mmr_lambda, out)) return rc;
    if (!q || (nb > 0 && (!bm25_rows || !bm25_scores))) return fail(RAG_ERR_INVALID, "NULL buffer");
    if (min_score != min_score) return fail(RAG_ERR_INVALID, "min_score is NaN");
    RAG_LOCK_INDEX(h);
    int prev = 0;
    RAG_HIP_CHECK(hipGetDevice(&prev));
    if (prev != h->device) RAG_HIP_CHECK(hipSetDevice(h->device));
    struct Restore { int p, d; ~Restore() { if (p != d) (void)hipSetDevice(p); } } restore{prev, h->device};
    const HybridPlan pl = hybrid_plan(h, topn, (int)nb, fuse_limit, max_results);
    hipStream_t s = h->stream;
    if (int rc = ws_acquire(h, s)) return rc;
    const size_t mb = row_mask ? al256((size_t)h->n) : 0;
    if (int rc = h->ws3.ensure(pl.total + mb)) return rc;
    char* w = static_cast<char*>(h->ws3.ptr);
    float* q_dev = reinterpret_cast<float*>(w + pl.o_q);
    long long* bmr = reinterpret_cast<long long*>(w + pl.o_bmr);
    double* bms = reinterpret_cast<double*>(w + pl.o_bms);
    uint8_t* m_dev = row_mask ? reinterpret_cast<uint8_t*>(w + pl.total) : nullptr;
    RAG_HIP_CHECK(hipMemcpyAsync(q_dev, q, (size_t)h->dim * 4, hipMemcpyHostToDevice, s));
    if (nb > 0) {
        RAG_HIP_CHECK(hipMemcpyAsync(bmr, bm25_rows, (size_t)nb * 8, hipMemcpyHostToDevice, s));
        RAG_HIP_CHECK(hipMemcpyAsync(bms, bm25_scores, (size_t)nb * 8, hipMemcpyHostToDevice, s));
    }
    if (row_mask && h->n > 0) RAG_HIP_CHECK(hipMemcpyAsync(m_dev, row_mask, (size_t)h->n, hipMemcpyHostToDevice, s));
    // results land in the workspace (one block for the final hits) and are copied out behind the last kernel
    char* res = w + pl.o_res;
    rag_hybrid_out_t dst = {};
    dst.ids = reinterpret_cast<int64_t*>(res + pl.res_ids);
    dst.scores = reinterpret_cast<double*>(res + pl.res_scores);
    dst.picked = reinterpret_cast<int32_t*>(res + pl.res_picked);
    dst.n_out = reinterpret_cast<int32_t*>(res + pl.res_counts);
    if (int rc = hybrid_enqueue(h, pl, w, q_dev, min_score, nprobe, m_dev, bmr, bms, w_sem, w_bm25, mmr_lambda, mmr_threshold,
                                mmr_metric, dst, s))
        return rc;
    std::vector<unsigned char> host(pl.res_bytes);
    RAG_HIP_CHECK(hipMemcpyAsync(host.data(), res, pl.res_bytes, hipMemcpyDeviceToHost, s));
    const bool want_fused = out->fused_rows || out->fused_scores || out->fused_sem || out->fused_bm;
    std::vector<unsigned char> fused;
    const size_t fused_bytes = pl.o_rel - pl.o_frows;         // f_rows | f_score | f_sem | f_bm are adjacent
    if (want_fused) {
        fused.resize(fused_bytes);
        RAG_HIP_CHECK(hipMemcpyAsync(fused.data(), w + pl.o_frows, fused_bytes, hipMemcpyDeviceToHost, s));
    }
    RAG_HIP_CHECK(hipStreamSynchronize(s));
    h->ws_pending = false;
    const int32_t* counts = reinterpret_cast<const int32_t*>(host.data() + pl.res_counts);
    const int32_t m = counts[0], nf = counts[1], st = counts[2];
    if (m < 0 || m > pl.M || nf < 0 || nf > pl.L) return fail(RAG_ERR_HIP, "hybrid pipeline returned a bad count");
    memcpy(out->ids, host.data() + pl.res_ids, (size_t)m * 8);
    memcpy(out->scores, host.data() + pl.res_scores, (size_t)m * 8);
    if (out->picked) memcpy(out->picked, host.data() + pl.res_picked, (size_t)m * 4);
    *out->n_out = m;
    if (out->n_fused) *out->n_fused = nf;
    if (out->status) *out->status = st;
    if (want_fused) {
        if (out->fused_rows) memcpy(out->fused_rows, fused.data() + (pl.o_frows - pl.o_frows), (size_t)nf * 8);
        if (out->fused_scores) memcpy(out->fused_scores, fused.data() + (pl.o_fscore - pl.o_frows), (size_t)nf * 8);
        if (out->fused_sem) memcpy(out->fused_sem, fused.data() + (pl.o_fsem - pl.o_frows), (size_t)nf * 8);
        if (out->fused_bm) memcpy(out->fused_bm, fused.data() + (pl.o_fbm - pl.o_frows), (size_t)nf * 8);
    }
    return RAG_OK;
}

}  // extern "C"
