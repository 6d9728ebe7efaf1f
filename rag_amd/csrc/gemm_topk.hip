// Batched exhaustive search: S = X * Q^T on the fp32 matrix cores, fused with
// top-k selection so that no score is ever written to memory (gfx950).
//
// Roofline: fp32 MFMA (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD).  Algorithmic
// work per launch = 2 * nq * n * ld flops.  Scores are bit-for-bit a k-ordered
// fp32 fma chain, the same arithmetic a CPU inner product performs.
//
// Tiling.  A workgroup (4 waves) owns a block of QT queries and a contiguous
// range of corpus rows, which it walks in tiles of RT rows.  Per tile it runs
// the K loop in steps of 32 through LDS (double buffered, one barrier per step;
// the next step's global loads are in flight while the current step computes).
// Three shapes, chosen by the batch size:
//      RT x QT     waves (rows x queries)   per wave            used for
//      256 x 256   4 x 2 (8 waves)          64 x 128, 2x4 MFMA  nq >= 1024
//      128 x 128   2 x 2                    64 x 64, 2x2 MFMA   64 < nq < 1024
//      128 x 64    2 x 2                    64 x 32, 2x1 MFMA   32 < nq <= 64
//      256 x 32    4 x 1                    64 x 32, 2x1 MFMA   5 <= nq <= 32
// The narrow shapes spend no matrix-core time on padding queries: a batch of 32
// costs about one HBM pass over the corpus instead of a 128-query tile's work.
// Corpus rows sit on the MFMA's row index (registers), queries on its column
// index (lanes): a lane owns NJ queries and sees 2 x 32 rows of each.
//
// A lane reads k = 8*kb + 4*h + s (h = lane >> 5, s = 0..3) of its row as one
// ds_read_b128 and feeds element s to MFMA step s; both operands use the same k
// order, so the sum over (h, s) covers the 8 k's of the block exactly once.
//
// Staging.  global_load_lds_dwordx4 writes the tiles straight into LDS (no
// staging registers, no ds_write).  The LDS image is dense [rows][8 x 16 B]; the
// 16-byte chunk c of row r sits at position c ^ ((r >> 1) & 7), applied on the
// SOURCE address of the DMA and again on the fragment read, which makes the reads
// of any 16 rows that differ mod 16 hit 16 different bank groups.
//
// Selection.  Per query the workgroup keeps a threshold (LDS) and a candidate
// buffer of TRIG + RT keys (global memory, private to the workgroup, L2 resident).  A
// score enters the buffer only if it beats the threshold, which after the first
// few tiles is rare: the epilogue is a max tree + one ballot per wave, and only
// a wave with a hit walks its accumulators.  When a query's buffer could
// overflow on the next tile (count > TRIG) one wave sorts it in LDS, keeps k
// and tightens the threshold.  At the end every query's k best keys go to `part`,
// and merge_keys_kernel merges the row-range partials.
//
// Optional bf16 first pass (B16 = true).  The same kernel runs on bf16 copies of the corpus
// and the queries with v_mfma_f32_32x32x16_bf16 (one MFMA per 16-byte fragment instead of
// four) and keeps, per query, every row whose approximate score lies within a margin of the
// k-th best one.  margin = 2 * delta with delta >= |approximate - exact| for every row
// (delta = c * max row norm * query norm), so the exact top-k is a subset of the kept rows;
// rescore_select_kernel re-scores them in fp32 and selects.  A buffer that cannot hold the
// kept rows (many near ties) raises a flag and the batch is answered by the fp32 kernel.
//
// Placement.  blockIdx -> (query block, row range) is chosen so that the
// workgroups of one XCD (blockIdx % 8) share few query blocks: their query tiles
// stay in that XCD's L2 while the corpus streams through.
#include <float.h>

#include <algorithm>

#include "index.h"
#include "ivf_lists.h"
#include "topk_keys.h"

namespace rag {

int merge_cap(long long per_query, int k);

static int env_int_g(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int GK = 32;           // K step
constexpr int GEMM_MAX_K = 64;

// A query's candidate buffer is compacted (sorted, cut to k, threshold tightened) once it holds
// more than TRIG keys at the end of a tile; a tile adds at most RT, so it holds TRIG + RT.
// Narrow shapes see few rows per workgroup (the corpus is split 512 ways for a single query
// block), so they tighten early and often; the square shape streams long ranges and rarely compacts.
// The narrow shapes trigger at 64 keys for k <= 32 and at 128 above (a trigger below 2k would
// compact after every tile); their buffers are sized for the larger one.
__host__ __device__ constexpr int gemm_trig_max(int RT, int QT) { return QT >= 128 ? 256 : 128; }
// Wide shapes (QT >= 128): the threshold of a query only moves when its buffer is compacted, so a
// late trigger leaves it stale -- with 256 the k-th best of the first 256 rows stood for the next 6000
// rows and every tile pushed a dozen candidates per wave.  Triggering at about 3k keeps the threshold
// within a factor of ~3 of the current one; the sort of a compaction then covers 64 or 128 keys.
// The bf16 pass carries every key within the margin of the k-th best across a compaction (a few times k
// on unstructured data), so it triggers later.
// k above 64 (the wide shapes only, up to GEMM_BIG_MAX_K): the trigger is 256 keys, which has to stay well above k -- a buffer
// cut to k keys that is past its trigger again after a push would compact at every tile.  The buffers hold 256 + RT keys.
__host__ __device__ constexpr int gemm_trig(int RT, int QT, int k, bool b16 = false) {
    return QT >= 128 ? (b16 ? (k <= 16 ? 128 : RT) : (k <= 16 ? 32 : (k <= 32 ? 64 : (k <= 64 ? 128 : 256))))
                     : (b16 ? 128 : (k <= 32 ? 64 : 128));
}
__host__ __device__ constexpr int gemm_capq(int RT, int QT) { return gemm_trig_max(RT, QT) + RT; }

// Timing-only ablations for diagnostic builds (outputs are wrong when non-zero):
// bit 0 skip the tile epilogue, bit 1 load global tiles only once, bit 2 barrier only twice,
// bit 10 every load re-reads the first K step of the first tile (load issue cost without memory latency),
// bit 11 the bf16 pass issues 16x16x32 MFMAs on its fragments (same cycles; the clock the smaller shape holds).
#ifndef RAG_GEMM_DIAG
#define RAG_GEMM_DIAG 0
#endif
constexpr int DIAG = RAG_GEMM_DIAG;

struct GemmArgs {
    const float* X;
    const float* Q;
    const uint8_t* mask;
    const uint8_t* mask2;  // a second mask (soft deletes and the caller's filter, not AND-ed first)
    int mask_aliased;      // 1: the masks count rows as row_alias does (insertion order), not storage order
    u64* cand;       // [grid][QT][TRIG + RT]
    u64* part;       // [nq][S][outc]
    const float* margin;   // bf16 pass: per query, how far below the k-th approximate score a row is kept
    int* overflow;         // bf16 pass: set when a buffer cannot hold the kept rows
    const int* run_if;     // when not NULL the kernel runs only if *run_if != 0
    int outc;              // keys per (query, row range) in part: k, or PREF_OUTC for the bf16 pass
    long long n;
    long long ntiles;
    int nq;
    int ld;
    int k;
    int nqb;         // query blocks
    int nqb_pad;     // query blocks rounded up for the XCD-friendly decode
    int S;           // row ranges
    int gq;          // placement: query-block groups across the 8 XCDs (0: plain b % nqb_pad decode)
    int trig;        // a query's buffer is compacted once it holds more keys than this (gemm_trig)
    // list-major IVF scan (TASKS): a work item = up to QT queries that probe one list
    const int* total_tasks;        // device: number of work items
    const int* task_list;          // [tasks] list of the work item
    const int* task_begin;         // [nlist] first work item of a list
    const int* pair_begin;         // [nlist] first (query, probe) pair of a list
    const int* pair_cnt;           // [nlist] pairs of a list
    const int* pair_q;             // [pairs] query of a pair, grouped by list
    const int* pair_slot;          // [pairs] where the pair's k keys go in part (slot * k)
    const long long* list_begin;   // [nlist + 1] rows of a list (storage order)
    const long long* row_alias;    // storage position -> row number reported in keys (NULL: identity)
    // the best k-th score any row range has reached so far, per query (ordered-float bits, 0 = none
    // yet).  Every workgroup of a query raises it when it compacts and reads it once per tile: a row
    // scoring below another range's k-th best cannot be in the answer.  The bf16 pass publishes its
    // k-th best approximate score minus the margin, below which no range needs to keep a row (and
    // which also lies below the exact k-th best, so the fp32 fallback may start from it).
    unsigned int* gthr;
    // STORE variant: every score goes to scores[query * n + row] and nothing is selected
    float* scores;
};

// cache policy bits of the corpus loads in the narrow shapes (2 = non-temporal)
#ifndef RAG_B16_SCHED
#define RAG_B16_SCHED 1
#endif
// which kernels hand K steps over in front of the last block of MFMAs: bit 0 fp32, bit 1 bf16 pass,
// bit 2 also the narrow shapes
#ifndef RAG_GEMM_HANDOVER
#define RAG_GEMM_HANDOVER 1
#endif
#ifndef RAG_GEMM_NARROW_AUX
#define RAG_GEMM_NARROW_AUX 2
#endif
// rare path of the tile epilogue.  Inlined on purpose.  As an out-of-line call (-DRAG_PUSH_OUTLINE,
// a diagnostic build) the generated code is sound and gives the same results on every GEMM-path
// test: the callee touches v0..v13 and s0..s9 of the caller-saved set, honours EXEC, and finds
// s_cnt / s_flag through the dynamic-LDS offset table under the kernel id in s15.  What the call
// costs is private memory: `const GemmArgs&` then needs an address, so each kernel copies its
// argument block into scratch at entry and the callee reads mask / row_alias / trig back through
// flat loads, and the accumulators and thresholds that live across calls made under a partial EXEC
// mask are spilled around the K loop and inside the divergent epilogue -- 208..352 bytes of scratch
// per lane and 256 VGPRs in every instantiation, for a path a handful of lanes per tile take.
// Inlined, the kernels use no scratch, and the build fails if that ever changes.
#ifdef RAG_PUSH_OUTLINE
#define RAG_PUSH_ATTR __noinline__
#else
#define RAG_PUSH_ATTR __forceinline__
#endif
template <int RT, int QT>
__device__ RAG_PUSH_ATTR void push_candidate(const GemmArgs& a, u64* cand, int* s_cnt, int* s_flag, int ql,
                                             long long row, long long row_hi, float sc) {
    constexpr int CAPQ = gemm_capq(RT, QT);
    if (row >= row_hi) return;
    const long long shown = a.row_alias ? a.row_alias[row] : row;       // the row number keys (and aliased masks) use
    const long long mr = a.mask_aliased ? shown : row;
    if ((!a.mask || a.mask[mr]) && (!a.mask2 || a.mask2[mr])) {
        const int pos = atomicAdd(&s_cnt[ql], 1);
        cand[(size_t)ql * CAPQ + pos] = make_key(sc, (uint32_t)shown);
        if (pos + 1 > a.trig) *s_flag = 1;       // compact before the next tile can overflow it
    }
}

// bf16 pass: keys a (query, row range) may hand to the re-scoring step -- 128, or 64 when one query
// block is split over more than 128 row ranges (each range then sees few rows and keeps few); for k > 16
// twice that (the keys within the margin of the k-th best number about 2k to 3k on unstructured data)
// (32, 64 for k > 10, 128 for k > 16 in the narrow 256 x 32 shape, whose 512 ranges of a query see 1/512 of the corpus each and are
// cut at the best threshold any range has published)
static int pref_outc(int S, int QT = 256, int k = 10) { return QT <= 64 ? (k <= 10 ? 32 : k <= 16 ? 64 : 128) : (k > 16 ? (S <= 64 ? 256 : 128) : (S <= 128 ? 128 : 64)); }

// number of leading keys of a descending list whose score is above `thr` (wave-wide, n <= 512)
__device__ __forceinline__ int count_above(const u64* sorted, int n, float thr, int lane) {
    int kept = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool above = i < n && sorted[i] != KEY_EMPTY && key_score(sorted[i]) > thr;
        kept += __popcll(__ballot(above));
    }
    return kept;
}

// one wave per query: sort the candidate buffer of every query that could overflow on the next
// tile, keep the k best (fp32) or every key within the margin of the k-th best (bf16 pass),
// tighten the threshold
template <int RT, int QT, int NW, bool B16>
__device__ __forceinline__ void compact_queries(u64* scratch, u64* cand, int* s_cnt, float* s_thr, const float* s_margin,
                                                int* overflow, unsigned int* gthr, const int* qmap, int k, int TRIG, int wave, int lane) {
    // gthr[qmap ? qmap[ql] : ql] is the query's shared threshold (work items of the list-major scan map their
    // columns to queries through qmap)
    constexpr int CAPQ = gemm_capq(RT, QT);
    for (int ql = wave; ql < QT; ql += NW) {
        const int c = s_cnt[ql];
        if (c <= TRIG) continue;            // wave-uniform
        const int n2 = pow2_ceil(c > k ? c : k);
        for (int i = lane; i < n2; i += 64) scratch[i] = (i < c) ? cand[(size_t)ql * CAPQ + i] : KEY_EMPTY;
        bitonic_sort_desc_lds(scratch, n2, lane, 64, SyncWave());
        int keep = c < k ? c : k;
        float thr = c >= k ? key_score(scratch[k - 1]) : -INFINITY;
        if constexpr (B16) {
            if (c >= k) {
                thr -= s_margin[ql];
                keep = count_above(scratch, c, thr, lane);
                if (keep > TRIG) {           // more near ties than the buffer can carry over
                    if (lane == 0) *overflow = 1;
                    keep = TRIG;
                }
            }
        }
        for (int i = lane; i < keep; i += 64) cand[(size_t)ql * CAPQ + i] = scratch[i];
        if (lane == 0) {
            s_cnt[ql] = keep;
            if (c >= k) {
                s_thr[ql] = thr;
                if (gthr) atomicMax(&gthr[qmap ? qmap[ql] : ql], f32_ordered(thr));
            }
        }
        SyncWave()();
    }
}

// Tile epilogue.  A lane owns NJ queries and 2 x 16 scores of each.  First the cheap test "does
// one of my 16-score groups beat its query's threshold" (a max tree per group, one ballot per
// group); the wave walks only the groups in which some lane has a hit -- a wave meets about one
// candidate per tile for most of a row range, so it walks 16 accumulators, not all 32 * NJ.
// rbase = first corpus row of this wave's 64, qbase = first query (within the block) of this
// wave's NJ * 32.
template <int RT, int QT, int NJ>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& a, f16v (&acc)[2][NJ], u64* cand, float* s_thr, int* s_cnt,
                                              int* s_flag, int m_valid, long long rbase, long long row_hi, int qbase, int l31, int h,
                                              const float (&tg)[NJ]) {
    float thr[NJ];
    bool hit[2][NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        const int ql = qbase + nj * 32 + l31;
        thr[nj] = s_thr[ql];                                // +inf for the padding queries of the block
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[mi][nj][r]);
            hit[mi][nj] = m > thr[nj] && m >= tg[nj];      // a tie with another range's k-th best may still win on row number
        }
    }
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        const int ql = qbase + nj * 32 + l31;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (__ballot(hit[mi][nj]) == 0ull) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sc = acc[mi][nj][r];
                if (sc > thr[nj] && sc >= tg[nj])
                    push_candidate<RT, QT>(a, cand, s_cnt, s_flag, ql, rbase + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, row_hi, sc);
            }
        }
    }
}

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int RT, int QT, int NW, bool B16, bool TASKS = false, bool STORE = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_topk_kernel(GemmArgs a) {
    static_assert(!TASKS || (!B16 && QT == 32), "work items of the list-major scan use the 256 x 32 fp32 shape");
    static_assert(!STORE || (!B16 && !TASKS), "the score-writing variant is the plain fp32 kernel without its selection");
    constexpr int NT = NW * 64;              // threads
    constexpr int WM = RT / 64;              // waves along the corpus rows
    constexpr int WQ = NW / WM;              // waves along the queries
    constexpr int NJ = QT / WQ / 32;         // 32-query MFMA columns per wave
    constexpr int CAPQ = gemm_capq(RT, QT);
    constexpr int CAPQ2 = CAPQ <= 256 ? 256 : 512;      // sort size bound (power of two)
    constexpr int APW = RT / (8 * NW);       // 1-KiB DMA pieces (8 rows x 128 B) per wave and K step, corpus
    constexpr int BPW = QT / (8 * NW);       //   ... queries
    constexpr int TILEA = RT * GK;           // floats per corpus stage
    constexpr int TILEB = QT * GK;
    // cache policy of the corpus loads: a single query block reads every row once (non-temporal,
    // bit 1); with many query blocks the rows are re-read by other workgroups and should stay cached
    constexpr int A_AUX = (QT >= 128) ? 0 : RAG_GEMM_NARROW_AUX;
    // hand-over of K steps in front of a step's last block of MFMAs (see the K loop)
    constexpr bool HO = !TASKS && ((RAG_GEMM_HANDOVER & (B16 ? 2 : 1)) != 0) && (QT >= 128 || (RAG_GEMM_HANDOVER & 4) != 0);
    static_assert(WM * WQ == NW && NJ >= 1 && WM * 64 == RT && WQ * NJ * 32 == QT && APW >= 1 && BPW >= 1, "tile shape");
    static_assert(NW * CAPQ2 * 8 <= TILEA * 4, "compaction scratch must fit in one idle corpus stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* As = reinterpret_cast<float*>(smem);                       // [2][TILEA] corpus rows
    float* Bs = As + 2 * TILEA;                                        // [2][TILEB] queries
    float* s_thr = Bs + 2 * TILEB;                                     // [QT]
    int* s_cnt = reinterpret_cast<int*>(s_thr + QT);                   // [QT]
    int* s_flag = s_cnt + QT;                                          // [1] (+3 pad)
    float* s_margin = reinterpret_cast<float*>(s_flag + 4);            // [QT], bf16 pass only

    if (a.run_if && *a.run_if == 0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WQ;        // which 64 corpus rows of the tile
    const int wq = wave % WQ;        // which NJ*32 queries of the block
    const int l31 = lane & 31;
    const int h = lane >> 5;

    const int b = blockIdx.x;
    const int k = a.k;
    // a K step is 128 bytes of every row: 32 floats, or 64 bf16
    const int KT = B16 ? a.ld / (2 * GK) : a.ld / GK;
    u64* cand = a.cand + (size_t)b * QT * CAPQ;
    const int ntasks = TASKS ? *a.total_tasks : 1;
  for (int task = TASKS ? b : 0; task < ntasks; task += TASKS ? (int)gridDim.x : 1) {
    // ---- the work item: which queries, which rows, where the keys go ------------------
    int q0 = 0, sp = 0, m_valid, p0 = 0;
    long long row_lo = 0, row_hi = a.n, t_begin, t_end;
    if constexpr (TASKS) {
        const int lst = a.task_list[task];
        const int g = task - a.task_begin[lst];
        p0 = a.pair_begin[lst] + g * QT;
        m_valid = a.pair_cnt[lst] - g * QT;
        m_valid = m_valid < QT ? m_valid : QT;
        row_lo = a.list_begin[lst];
        row_hi = a.list_begin[lst + 1];
        t_begin = 0;
        t_end = (row_hi - row_lo + RT - 1) / RT;
    } else {
        int qb;
        if (a.gq > 0) {
            // XCD x = b % 8 owns query-block group x % gq and row-range group x / gq; inside the XCD
            // consecutive workgroups take different query blocks of the same row range
            const int xcd = b & 7, j = b >> 3;
            const int m = a.nqb_pad / a.gq, r = a.S / (8 / a.gq);
            qb = (xcd % a.gq) * m + (j % m);
            sp = (xcd / a.gq) * r + (j / m);
        } else {
            qb = b % a.nqb_pad;
            sp = b / a.nqb_pad;
        }
        if (qb >= a.nqb) return;
        q0 = qb * QT;
        m_valid = a.nq - q0 < QT ? a.nq - q0 : QT;
        t_begin = a.ntiles * sp / a.S;
        t_end = a.ntiles * (sp + 1) / a.S;
    }

    for (int i = tid; i < QT; i += NT) {
        s_thr[i] = i < m_valid ? -INFINITY : INFINITY;      // padding queries keep nothing
        s_cnt[i] = 0;
        if constexpr (B16) s_margin[i] = (q0 + i) < a.nq ? a.margin[q0 + i] : 0.f;
    }
    if (tid == 0) *s_flag = 0;
    __syncthreads();

    const f4* X4 = reinterpret_cast<const f4*>(a.X);
    const f4* Q4 = reinterpret_cast<const f4*>(a.Q);
    const int ld4 = B16 ? a.ld / 8 : a.ld / 4;       // 16-byte chunks per row

    // ---- loader ------------------------------------------------------------------------
    // wave instruction i fills tile rows (APW*wave + i)*8 .. +7 (queries: BPW), lane -> (row = lane>>3,
    // position = lane&7); the chunk fetched for a position is position ^ ((row>>1)&7).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    int arow[APW], achunk[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        arow[i] = (wv * APW + i) * 8 + (lane >> 3);
        achunk[i] = (lane & 7) ^ ((arow[i] >> 1) & 7);
    }
    const f4* qp[BPW];
    const f4* xp[APW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int brow = (wv * BPW + i) * 8 + (lane >> 3);
        int qr;
        if constexpr (TASKS) {
            qr = a.pair_q[p0 + (brow < m_valid ? brow : m_valid - 1)];
        } else {
            qr = q0 + brow;
            qr = qr < a.nq ? qr : a.nq - 1;
        }
        qp[i] = Q4 + (long long)qr * ld4 + ((lane & 7) ^ ((brow >> 1) & 7));
    }
    // A work item's last tile is ragged (a list of 780 rows has a fourth 256-row tile with 12 of them).  Its waves
    // each load and multiply their own 64 rows of a tile (WQ == 1), so a wave whose rows all lie past the end of the
    // list skips its loads, fragment reads and MFMAs for that tile and only takes the barriers.
    static_assert(!TASKS || (WQ == 1 && APW * 8 == 64), "a wave of a work item loads the 64 tile rows it multiplies");
    bool a_live = true;              // the tile xp points at has rows for this wave
    auto set_tile = [&](long long t) {
        if constexpr (TASKS) a_live = row_lo + t * RT + wv * 64 < row_hi;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            long long r = row_lo + t * RT + arow[i];
            r = r < row_hi ? r : row_hi - 1;
            xp[i] = X4 + r * ld4 + achunk[i];
        }
    };
    // request the tiles of K step kt (of the tile xp points at) for LDS buffer `buf`
    auto issue_load = [&](int kt, int buf) {
        unsigned char* Ab = reinterpret_cast<unsigned char*>(As + buf * TILEA) + wv * (APW * 1024);
        unsigned char* Bb = reinterpret_cast<unsigned char*>(Bs + buf * TILEB) + wv * (BPW * 1024);
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            if (!TASKS || a_live)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xp[i] + kt * (GK / 4)),
                                                 (__attribute__((address_space(3))) void*)(Ab + i * 1024), 16, 0, A_AUX);
            if (i < BPW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qp[i < BPW ? i : 0] + kt * (GK / 4)),
                                                 (__attribute__((address_space(3))) void*)(Bb + i * 1024), 16, 0, 0);
        }
    };
    // fragment read offsets (floats) of this lane inside an operand tile, per 8-wide k block
    int foff[GK / 8];
#pragma unroll
    for (int kb = 0; kb < GK / 8; ++kb) foff[kb] = l31 * GK + (((kb * 2 + h) ^ ((l31 >> 1) & 7)) << 2);

    f16v acc[2][NJ];
    int buf = 0;
    if (t_begin < t_end) {
        set_tile(t_begin);
        issue_load(0, 0);
    }
    for (long long t = t_begin; t < t_end; ++t) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;
        // the other row ranges' progress on this lane's queries: requested at the start of the last
        // K step, used after it (the latency hides behind that step's MFMAs)
        float tg[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) tg[nj] = -INFINITY;
        // fragments of k block kb+1 are requested while the MFMAs of block kb issue (two register
        // sets, kb parity).  Issue order, fixed with sched_group_barrier: per block [MFMAs, 1 read]
        // until the next block's reads are out, then the rest of the block's MFMAs -- every read has
        // most of a block to land.
        constexpr int READS = 2 + NJ, MFMAS = 8 * NJ, NKB = GK / 8;
        f4 fa[2][2], fb[2][NJ];
        const bool cur_live = !TASKS || (row_lo + t * RT + wv * 64 < row_hi);       // wave-uniform
        for (int kt = 0; kt < KT; ++kt, buf ^= 1) {
            const float* Ab = As + buf * TILEA + (wm * 64) * GK;
            const float* Bb = Bs + buf * TILEB + (wq * NJ * 32) * GK;
            if (!HO || kt == 0) {
                if (!(DIAG & 4) || (t == t_begin && kt < 2))
                    __syncthreads();     // stage `buf` is complete (the LDS-DMA in flight is drained here too)
                // a candidate buffer filled up during the previous tile's epilogue: every wave has
                // passed the barrier above, so the flag is stable and stage buf^1 is idle
                if (kt == 0 && *s_flag != 0) {
                    compact_queries<RT, QT, NW, B16>(reinterpret_cast<u64*>(As + (buf ^ 1) * TILEA) + wave * CAPQ2, cand, s_cnt, s_thr, s_margin,
                                                     a.overflow, a.gthr ? (TASKS ? a.gthr : a.gthr + q0) : nullptr, TASKS ? a.pair_q + p0 : nullptr, k, a.trig,
                                                     wave, lane);
                    __syncthreads();
                    if (tid == 0) *s_flag = 0;
                }
                if (!(DIAG & 2) || (t == t_begin && kt == 0)) {
                    if (DIAG & 1024) issue_load(0, buf ^ 1);       // same instructions, cache-resident source
                    else if (kt + 1 < KT) issue_load(kt + 1, buf ^ 1);
                    else if (t + 1 < t_end) { set_tile(t + 1); issue_load(0, buf ^ 1); }
                }
                if constexpr (TASKS) {
                    if (!cur_live) continue;          // nothing of this tile belongs to the wave
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) fa[0][mi] = *reinterpret_cast<const f4*>(Ab + mi * 32 * GK + foff[0]);
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj) fb[0][nj] = *reinterpret_cast<const f4*>(Bb + nj * 32 * GK + foff[0]);
                if (!B16 || RAG_B16_SCHED) __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);
            }
            if (kt == KT - 1 && a.gthr) {
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj) {
                    const int ql = wq * NJ * 32 + nj * 32 + l31;
                    if (ql < m_valid) {
                        // a work item of the list-major scan finds its column's query through pair_q: every list a
                        // query probes competes for the same k places, so the lists share the query's threshold
                        const int qg = TASKS ? a.pair_q[p0 + ql] : q0 + ql;
                        const unsigned int o = __hip_atomic_load(&a.gthr[qg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (o != 0u) tg[nj] = ordered_f32(o);
                    }
                }
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const int cur = kb & 1, nxt = cur ^ 1;
                bool reads = kb + 1 < NKB;
                if (kb + 1 < NKB) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) fa[nxt][mi] = *reinterpret_cast<const f4*>(Ab + mi * 32 * GK + foff[kb + 1]);
#pragma unroll
                    for (int nj = 0; nj < NJ; ++nj) fb[nxt][nj] = *reinterpret_cast<const f4*>(Bb + nj * 32 * GK + foff[kb + 1]);
                } else if (HO && kt + 1 < KT) {
                    // hand-over to the next K step in front of this step's last block of MFMAs: the
                    // fragments of that block are in registers, so after the barrier stage `buf` is free
                    // for the step after next, and the first fragments of the next step (stage buf^1,
                    // complete at this barrier) are read while those MFMAs run
                    __syncthreads();
                    if (!(DIAG & 2)) {
                        if (kt + 2 < KT) issue_load(kt + 2, buf);
                        else if (t + 1 < t_end) { set_tile(t + 1); issue_load(0, buf); }
                    }
                    const float* An = As + (buf ^ 1) * TILEA + (wm * 64) * GK;
                    const float* Bn = Bs + (buf ^ 1) * TILEB + (wq * NJ * 32) * GK;
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) fa[nxt][mi] = *reinterpret_cast<const f4*>(An + mi * 32 * GK + foff[0]);
#pragma unroll
                    for (int nj = 0; nj < NJ; ++nj) fb[nxt][nj] = *reinterpret_cast<const f4*>(Bn + nj * 32 * GK + foff[0]);
                    reads = true;
                }
                if constexpr (B16) {
                    // a 16-byte fragment is 8 bf16 = the whole K extent of one 32x32x16 MFMA
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int nj = 0; nj < NJ; ++nj)
                        {
                            if (DIAG & 2048) {
                                // timing only (answers invalid): the same fragments and cycles as 16x16x32 MFMAs, two per pair and
                                // k block on alternating quarters of the accumulator -- does the smaller shape hold a higher clock?
                                typedef float f4v __attribute__((ext_vector_type(4)));
                                f16v& c = acc[mi][nj];
#pragma unroll
                                for (int h2 = 0; h2 < 2; ++h2) {
                                    const int qd = 2 * (kb & 1) + h2;
                                    f4v part = {c[4 * qd], c[4 * qd + 1], c[4 * qd + 2], c[4 * qd + 3]};
                                    part = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fa[cur][mi]),
                                                                                   __builtin_bit_cast(bf8, fb[cur][nj]), part, 0, 0, 0);
                                    c[4 * qd] = part[0]; c[4 * qd + 1] = part[1]; c[4 * qd + 2] = part[2]; c[4 * qd + 3] = part[3];
                                }
                            } else {
                                acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[cur][mi]),
                                                                                      __builtin_bit_cast(bf8, fb[cur][nj]), acc[mi][nj], 0, 0, 0);
                            }
                        }
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                            for (int nj = 0; nj < NJ; ++nj)
                                acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][mi][s], fb[cur][nj][s], acc[mi][nj], 0, 0, 0);
                }
                if constexpr (B16 && RAG_B16_SCHED) {
                    // 2*NJ MFMAs per block: one read behind each of the first READS MFMAs
                    if (reads) {
#pragma unroll
                        for (int j = 0; j < READS; ++j) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ - READS, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ, 0);
                    }
                }
                if constexpr (!B16) {
                    if (reads) {
#pragma unroll
                        for (int j = 0; j < READS; ++j) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, MFMAS - 2 * READS, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, MFMAS, 0);
                    }
                }
            }
        }
        if (DIAG & 1) {
            float keep = 0.f;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) keep += acc[mi][nj][r];
            if (keep == 123.456f) s_thr[0] = keep;      // keeps the accumulators live
            continue;
        }
        if constexpr (STORE) {
            // a lane holds, per query column, runs of 4 consecutive rows: one 16-byte store each where the row count allows
            const long long rbase = row_lo + t * RT + wm * 64;
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj) {
                const int ql = wq * NJ * 32 + nj * 32 + l31;
                if (ql >= m_valid) continue;
                float* out = a.scores + (size_t)(q0 + ql) * (size_t)a.n;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const long long row = rbase + mi * 32 + 8 * g + 4 * h;
                        if ((a.n & 3) == 0 && row + 3 < row_hi) {
                            *reinterpret_cast<f4*>(out + row) = (f4){acc[mi][nj][4 * g], acc[mi][nj][4 * g + 1], acc[mi][nj][4 * g + 2], acc[mi][nj][4 * g + 3]};
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (row + e < row_hi) out[row + e] = acc[mi][nj][4 * g + e];
                        }
                    }
            }
            continue;
        }
        if constexpr (TASKS) {
            if (!cur_live) continue;
        }
        tile_epilogue<RT, QT, NJ>(a, acc, cand, s_thr, s_cnt, s_flag, m_valid, row_lo + t * RT + wm * 64, row_hi, wq * NJ * 32, l31, h, tg);
    }

    // ---- final: every query's k best keys of this row range -> part ---------------------
    __syncthreads();
    if constexpr (!STORE) {
        u64* scratch = reinterpret_cast<u64*>(As) + wave * CAPQ2;
        for (int ql = wave; ql < QT; ql += NW) {
            if (ql >= m_valid) continue;
            const int c = s_cnt[ql];
            const int n2 = pow2_ceil(c > k ? c : k);
            for (int i = lane; i < n2; i += 64) scratch[i] = (i < c) ? cand[(size_t)ql * CAPQ + i] : KEY_EMPTY;
            bitonic_sort_desc_lds(scratch, n2, lane, 64, SyncWave());
            u64* out = TASKS ? a.part + (size_t)a.pair_slot[p0 + ql] * k
                             : a.part + ((size_t)(q0 + ql) * a.S + sp) * a.outc;
            if constexpr (B16) {
                // kept: above the range's own k-th best minus the margin and not below what any
                // range has published (its k-th best minus the margin)
                int keep = c;
                float cut = c >= k ? key_score(scratch[k - 1]) - s_margin[ql] : -INFINITY;
                if (a.gthr) {
                    const unsigned int o = __hip_atomic_load(&a.gthr[q0 + ql], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (o != 0u) cut = fmaxf(cut, ordered_f32(o));
                }
                if (cut > -INFINITY) keep = count_above(scratch, c, cut, lane);
                if (keep > a.outc) {
                    if (lane == 0) *a.overflow = 1;
                    keep = a.outc;
                }
                for (int i = lane; i < a.outc; i += 64) out[i] = i < keep ? scratch[i] : KEY_EMPTY;
            } else {
                for (int i = lane; i < k; i += 64) out[i] = scratch[i];
            }
            SyncWave()();
        }
    }
    if constexpr (TASKS) __syncthreads();      // the next work item re-initialises the LDS state
  }
}

// ---------------------------------------------------------------------------------------------
// Second step of the bf16 pass, one workgroup per query: gather the keys the row ranges kept,
// find the k-th best approximate score, re-score in fp32 every row within the margin of it,
// select the k best exact scores.
struct RescoreArgs {
    const u64* part;        // [nq][S * outc]
    int outc;               // keys per row range (a descending run)
    int per_query;          // S * outc
    const float* X;         // fp32 rows
    const float* Q;         // fp32 padded queries
    const float* margin;
    int ld;
    int* overflow;
    MergeArgs fin;          // k, min_score, id mapping, D, I
};
constexpr int RESC_CAP = 8192;      // keys of one query that fit the selection buffer (64 KB of LDS)

__device__ __forceinline__ void finalise_scored(const MergeArgs& a, long long q, int j, u64 key) {
    float s = -3.402823466e+38f;
    long long id = -1;
    if (key != KEY_EMPTY) {
        const float sc = key_score(key);
        if (sc >= a.min_score) {
            const uint32_t row = key_row(key);
            s = sc;
            id = a.row_ids ? a.row_ids[row] : a.id_base + (long long)row;
        }
    }
    a.D[(size_t)q * a.k + j] = s;
    a.I[(size_t)q * a.k + j] = id;
    if (a.rows_out) a.rows_out[(size_t)q * a.k + j] = s < -3.0e38f ? -1 : (long long)key_row(key);
}

__global__ __launch_bounds__(256) void rescore_select_kernel(RescoreArgs r) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
    u64* buf = reinterpret_cast<u64*>(rs_smem);                        // [RESC_CAP]
    int& s_n = *reinterpret_cast<int*>(buf + RESC_CAP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long q = blockIdx.x;
    const int k = r.fin.k;
    const u64* in = r.part + (size_t)q * r.per_query;
    // every row range handed over a descending run of `outc` keys: the k-th largest run head is at
    // most the k-th best approximate score, so nothing below (that head - margin) can matter --
    // with many short ranges most of what they kept falls below it and never enters the buffer
    u64 floor_key = KEY_EMPTY;
    const int runs = r.outc > 0 ? (int)(r.per_query / r.outc) : 0;
    if (runs >= k && runs <= RESC_CAP) {
        const int h2 = pow2_ceil(runs);
        for (int i = tid; i < h2; i += 256) buf[i] = i < runs ? in[(size_t)i * r.outc] : KEY_EMPTY;
        bitonic_sort_desc_lds(buf, h2, tid, 256, SyncBlock());
        const u64 hk = buf[k - 1];
        __syncthreads();
        if (hk != KEY_EMPTY) floor_key = make_key(key_score(hk) - r.margin[q], 0xFFFFFFFFu);   // lowest key of that score
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int i = tid; i < r.per_query; i += 256) {
        const u64 key = in[i];
        if (key != KEY_EMPTY && key >= floor_key) {
            const int pos = atomicAdd(&s_n, 1);
            if (pos < RESC_CAP) buf[pos] = key;
        }
    }
    __syncthreads();
    int n = s_n;
    if (n > RESC_CAP) {
        if (tid == 0) *r.overflow = 1;
        n = RESC_CAP;
    }
    int n2 = pow2_ceil(n > 1 ? n : 1);
    for (int i = n + tid; i < n2; i += 256) buf[i] = KEY_EMPTY;
    bitonic_sort_desc_lds(buf, n2, tid, 256, SyncBlock());
    // rows to re-score: everything above (k-th approximate score - margin); all of them when fewer than k
    const float thr = n >= k ? key_score(buf[k - 1]) - r.margin[q] : -INFINITY;
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256)
        if (key_score(buf[i]) > thr) atomicAdd(&s_n, 1);        // a prefix of the sorted list
    __syncthreads();
    const int m = s_n;
    // exact fp32 inner products, one wave per row
    const f4* q4 = reinterpret_cast<const f4*>(r.Q + (size_t)q * r.ld);
    const int ld4 = r.ld / 4;
    for (int i = wave; i < m; i += 4) {
        const uint32_t row = key_row(buf[i]);
        const f4* x4 = reinterpret_cast<const f4*>(r.X + (size_t)row * r.ld);
        float acc = 0.f;
        for (int c = lane; c < ld4; c += 64) {
            const f4 x = x4[c], y = q4[c];
            acc = fmaf(x.x, y.x, acc);
            acc = fmaf(x.y, y.y, acc);
            acc = fmaf(x.z, y.z, acc);
            acc = fmaf(x.w, y.w, acc);
        }
        acc = wave_sum_to_lane63(acc);
        if (lane == 63) buf[i] = make_key(acc, row);
    }
    __syncthreads();
    n2 = pow2_ceil(m > 1 ? m : 1);
    for (int i = m + tid; i < n2; i += 256) buf[i] = KEY_EMPTY;
    bitonic_sort_desc_lds(buf, n2, tid, 256, SyncBlock());
    for (int j = tid; j < k; j += 256) finalise_scored(r.fin, q, j, j < m ? buf[j] : KEY_EMPTY);
}

// fp32 rows -> bf16 (round to nearest even); the largest row norm seen so far is kept as the
// bit pattern of a non-negative float, which orders like an integer
__global__ void rows_to_bf16_kernel(const float* X, long long r0, long long r1, int ld, __bf16* Xb, int* maxnorm_bits) {
    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    const int ld4 = ld / 4;                       // rows are padded to a multiple of 4 floats
    for (long long r = r0 + wave; r < r1; r += nw) {
        const f4* src = reinterpret_cast<const f4*>(X + r * ld);
        bf4* dst = reinterpret_cast<bf4*>(Xb + r * ld);
        float ss = 0.f;
        for (int i = lane; i < ld4; i += 64) {
            const f4 v = src[i];
            ss = fmaf(v.x, v.x, ss);
            ss = fmaf(v.y, v.y, ss);
            ss = fmaf(v.z, v.z, ss);
            ss = fmaf(v.w, v.w, ss);
            bf4 o;
            o.x = (__bf16)v.x;
            o.y = (__bf16)v.y;
            o.z = (__bf16)v.z;
            o.w = (__bf16)v.w;
            dst[i] = o;
        }
        ss = wave_sum_to_lane63(ss);
        if (lane == 63) atomicMax(maxnorm_bits, __float_as_int(sqrtf(ss)));
    }
}

// padded fp32 queries -> bf16 copies and the per-query margin 2 * delta,
// delta = pref_err(ld) * (largest row norm) * (query norm)
// |approximate - exact| <= pref_err(ld) * |x| * |q|: both operands rounded to bf16 (relative 2^-9 each:
// 2^-8 + 2^-18 on the product, which is exact in fp32) plus the worst case of ld fp32 additions
// (ld * 2^-24, with 1 % on top for the rounded operands).  768 floats: 0.00396; the bound grows with the
// row width instead of running out of slack at a few thousand floats.
__host__ __device__ inline float pref_err(int ld) { return 0.00390625f + 3.8147e-6f + 1.01f * (float)ld * 5.9604645e-8f; }
__global__ void queries_to_bf16_kernel(const float* Q, int ld, const int* maxnorm_bits, __bf16* Qb, float* margin) {
    const long long q = blockIdx.x;
    const int lane = threadIdx.x;            // one wave per query
    const float* src = Q + q * ld;
    float ss = 0.f;
    for (int i = lane; i < ld; i += 64) {
        const float v = src[i];
        ss = fmaf(v, v, ss);
        Qb[q * ld + i] = (__bf16)v;
    }
    ss = wave_sum_to_lane63(ss);
    if (lane == 63) margin[q] = 2.f * pref_err(ld) * __int_as_float(*maxnorm_bits) * sqrtf(ss) * 1.0001f + 1e-30f;
}

// The seeded threshold words hold a lower bound B of the exact k-th best score.  The bf16 pass keeps a row when its
// approximate score is within the margin (2 delta) of the k-th best APPROXIMATE score, which is at least B - delta (the k
// rows that score B or more exactly score B - delta or more approximately): the word becomes B - 3 delta.
__global__ void pref_seed_adjust_kernel(unsigned int* gthr, const float* margin, long long nq) {
    const long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const unsigned int b = gthr[q];
    if (b != 0u) gthr[q] = f32_ordered(ordered_f32(b) - 1.5001f * margin[q]);
}

// bf16 first pass in the narrow 256 x 32 shape: one pass over half the bytes for up to 32 queries (k <= 32: the
// keys within the margin of a range's k-th best -- about 2k on unstructured data -- have to fit the 128 keys a
// range hands over and the 128-key trigger of its candidate buffers, or the batch goes through the fp32 kernel
// as well); the corpus has to give the single query block its row ranges
static bool narrow_bf16_ok(const rag_index* h, int k) {
    static const int narrow16 = env_int_g("RAG_GEMM_NARROW_BF16", 1);
    static const int narrow_kmax = env_int_g("RAG_GEMM_NARROW_BF16_KMAX", 32);       // measurement only
    return h->prefilter && narrow16 && k <= narrow_kmax && h->n >= 262144 && (h->ld % (2 * GK)) == 0;
}

// from 5 queries; with the bf16 first pass in the narrow shape from 2 (half a corpus pass instead of a whole one)
// The wide shapes (128 x 128, 256 x 256) carry up to 200 keys per query (candidate buffers of 256 + RT keys, the trigger at 256):
// they can keep the selection of a request up to k = 200 -- the reference's largest, limit * 2 with limit <= 100 -- in the kernel
// instead of sending a score matrix through HBM.  With thresholds that start open this pays only from 2048 queries in the
// 256 x 256 shape, where a workgroup's row range is long enough for them to settle (10M x 768, k = 200: 4096 queries 589 ms by the
// segment walk, 514 ms here; 512 queries 78.5 against 84.1, 1024 queries the same either way).
//
// Seeded thresholds: what makes k = 200 expensive inside the kernel is the start of a row range, where a query's threshold is
// loose and every tile pushes and sorts hundreds of keys.  On a corpus of 2^20+ rows the first few segments (of 32768 rows) go
// through the score-writing kernel first; a lower bound of their k-th best score per query (gemm_seed_thresholds, below) is a
// lower bound of the final one and becomes the starting value of the query's shared threshold word, so the
// selecting kernel then walks the whole corpus pushing only the handful of rows per range that beat it.  That makes the in-kernel
// selection pay from 65 queries (the 128 x 128 and 256 x 256 shapes) instead of 2048.
constexpr int GEMM_BIG_MAX_K = 200;
static bool gemm_big_shape(const rag_index* h, long long nq);
// The same step pays at k <= 64, most where a workgroup's row range is short -- a short corpus under a large batch, or a batch of a
// few dozen queries whose single query block is split 256 or 512 ways: the first tile of a range alone pushes and sorts a key per
// row and query.  There 4096 to 16384 leading rows seed the thresholds (k = 10; 100k x 384: 4096 queries 3.31 -> 2.86 ms,
// 256 queries 0.60 -> 0.34, 64 queries 0.28 -> 0.16; 1M x 768: 64 queries 1.24 -> 0.99, 512 queries 6.64 -> 6.36; 10M x 768:
// 64 queries 9.50 -> 9.19, 256 queries 30.0 -> 28.9, 4096 queries 452.3 -> 450.2).
constexpr long long SEED_SEG_ROWS = 32768;
static long long gemm_seed_rows(const rag_index* h, long long nq, int k);        // 0: the thresholds start open
static bool gemm_seeded(const rag_index* h, long long nq, int k) { return gemm_seed_rows(h, nq, k) > 0; }
static size_t gemm_seed_workspace(const rag_index* h, long long nq, int k);
static int gemm_seed_thresholds(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask, char* w,
                                unsigned int* gthr, hipStream_t s);
bool gemm_topk_supported(const rag_index* h, long long nq, int k) {
    static const int min_nq = env_int_g("RAG_GEMM_MIN_NQ", 5);
    static const int big_max_k = std::min(GEMM_BIG_MAX_K, env_int_g("RAG_GEMM_BIG_MAX_K", GEMM_BIG_MAX_K));      // 64 switches it off
    if (!(h->kind == RAG_KIND_FLAT && (h->ld % GK) == 0 && h->n >= 1)) return false;
    static const int big_k_min_nq = env_int_g("RAG_GEMM_BIG_K_MIN_NQ", 2048);
    if (k > GEMM_MAX_K && !(k <= big_max_k && !h->prefilter && (gemm_seeded(h, nq, k) || (nq >= big_k_min_nq && gemm_big_shape(h, nq))))) return false;
    return nq >= min_nq || (nq >= 2 && narrow_bf16_ok(h, k));
}

struct GemmPlan {
    int RT, QT, NW;      // tile shape, waves per workgroup
    int nqb, nqb_pad, S, gq;
    long long ntiles;
    size_t cand_bytes, part_bytes;
    bool pref;                                   // bf16 first pass + fp32 re-scoring
    size_t qb_bytes, margin_bytes, part16_bytes; // its extra workspace (then 256 bytes for the flag)
    size_t gthr_bytes;                           // shared per-query thresholds of the fp32 pass
};

// tile shape, (query blocks, padded query blocks, row ranges) and workspace for a batch of nq queries
static GemmPlan gemm_plan(const rag_index* h, long long nq, int k) {
    static const int target_wgs_env = env_int_g("RAG_GEMM_WGS", 0);
    static const int force_qt = env_int_g("RAG_GEMM_QT", 0);
    static const int big_min_nq = env_int_g("RAG_GEMM_BIG_MIN_NQ", 1024);
    GemmPlan p;
    // with the bf16 first pass on, the 256 x 256 shape (the one it runs in) already pays from 33 queries
    // (on a corpus long enough to give a single query block its 128+ row ranges)
    const int big_from = (h->prefilter && h->n >= 262144) ? std::min(big_min_nq, 33) : big_min_nq;
    // below that, the 256 x 256 shape still wins 2-3 % over 128 x 128 wherever it pads the batch no further
    // (129..256, 385..512, ... queries: 256 queries 30.6 -> 30.0 ms, 512: 59.6 -> 58.3, 768: 89.3 -> 86.9)
    // (on a corpus of 1M+ rows: a shorter one cannot give the fewer, larger workgroups 8 tiles each on every CU)
    const bool same_pad = nq > 128 && h->n >= (1 << 20) && (nq + 255) / 256 * 256 == (nq + 127) / 128 * 128;
    // 33..64 queries with the bf16 first pass: the 128 x 64 shape, whose query tile is a third of a K step's
    // loads instead of half of them
    static const int mid16 = env_int_g("RAG_GEMM_MID_BF16", 1);
    static const int mid16_maxq = env_int_g("RAG_GEMM_MID_BF16_MAXQ", 64);        // measurement only
    const bool mid_b16 = mid16 && nq > 32 && nq <= mid16_maxq && narrow_bf16_ok(h, k);
    // a short corpus under a large batch leaves a 256 x 256 workgroup a few dozen tiles, all of them walked while its
    // thresholds are still loose; the 128 x 128 shape (twice the workgroups, a quarter of the tile) warms up on fewer
    // rows (100k x 384 under 4096 queries: 3.46 -> 3.29 ms)
    const bool short_corpus = !h->prefilter && h->n < (1 << 19) && h->n > 32768;      // (up to 32768 rows such batches write a score matrix: no thresholds)
    p.QT = force_qt ? force_qt : mid_b16 ? 64 : (((nq >= big_from || same_pad) && !short_corpus) ? 256 : nq <= 32 ? 32 : nq <= 64 ? 64 : 128);
    p.RT = (p.QT == 32 || p.QT == 256) ? 256 : 128;
    p.NW = p.QT == 256 ? 8 : 4;
    const int target_wgs = target_wgs_env > 0 ? target_wgs_env : (p.NW == 8 ? 256 : 512);
    p.nqb = (int)((nq + p.QT - 1) / p.QT);
    // a row range should hold at least 8 tiles, or splitting costs more than it spreads -- unless the grid would
    // then leave compute units without a workgroup (a short corpus under a large batch: the 16384 centroids of an
    // IVF index under 4096 queries are 64 tiles for 16 query blocks), where ranges go down to one tile
    static const int min_tiles_short = std::max(1, env_int_g("RAG_GEMM_MIN_TILES_SHORT", 1));      // measurement knob
    p.ntiles = (h->n + p.RT - 1) / p.RT;
    long long max_s = (p.ntiles + 7) / 8;
    if (max_s * p.nqb < target_wgs) max_s = std::max(max_s, (p.ntiles + min_tiles_short - 1) / min_tiles_short);
    if (max_s < 1) max_s = 1;
    // Query blocks x row ranges = at most target_wgs workgroups (one round on the chip).  Placement over the XCDs (gq > 0:
    // XCD x owns query-block group x % gq and row-range group x / gq) needs the block count padded to a multiple of gq
    // and the range count a multiple of 8 / gq; padding blocks exit at once, so a padded grid leaves compute units idle
    // (9 blocks padded to 16 x 16 ranges: 144 of 256 workgroups, and 2304 queries took as long as 4096).  Of the
    // placements gq = 8, 4, 2, 1 and the plain one (gq = 0, no padding) the one with the most working workgroups is
    // taken, the widest split on a tie.
    static const int gq_env = env_int_g("RAG_GEMM_GQ", -1);
    long long best_active = -1;
    p.nqb_pad = p.nqb;
    p.S = 1;
    p.gq = 0;
    for (int gq = 8; gq >= 0; gq = gq > 1 ? gq >> 1 : gq - 1) {
        if (gq_env >= 0 && gq != gq_env) continue;
        const long long pad = gq > 0 ? (p.nqb + gq - 1) / gq * gq : p.nqb;
        long long s = std::min<long long>(target_wgs / pad, max_s);
        if (gq > 0) s -= s % (8 / gq);
        if (s < 1) continue;
        const long long active = (long long)p.nqb * s;
        if (active > best_active) {
            best_active = active;
            p.nqb_pad = (int)pad;
            p.S = (int)s;
            p.gq = gq;
        }
    }
    if (best_active < 0) {           // more query blocks than workgroups in a round (or a forced placement that does not fit)
        p.nqb_pad = p.nqb;
        p.S = 1;
        p.gq = 0;
    }
    p.cand_bytes = (size_t)p.nqb_pad * p.S * p.QT * gemm_capq(p.RT, p.QT) * sizeof(u64);
    p.part_bytes = ((size_t)nq * p.S * k * sizeof(u64) + 255) & ~(size_t)255;
    p.gthr_bytes = ((size_t)nq * sizeof(unsigned int) + 255) & ~(size_t)255;
    // the bf16 first pass pays off where the batch is MFMA-bound: the 256 x 256 shape
    // ... and, as one pass over half the bytes, the narrow shape of up to 32 queries
    const bool shape_ok = p.QT == 256 || ((p.QT == 32 || p.QT == 64) && narrow_bf16_ok(h, k) && p.S > 1);
    // k <= 32: above, the keys within the margin of the k-th best outgrow the buffers somewhere in nearly every
    // batch, and one overflow sends the whole batch through the fp32 kernel as well
    p.pref = h->prefilter && shape_ok && k <= 32 && (h->ld % (2 * GK)) == 0 && (long long)p.S * pref_outc(p.S, p.QT, k) <= 65536;
    p.qb_bytes = p.pref ? (((size_t)nq * h->ld * 2 + 255) & ~(size_t)255) : 0;
    p.margin_bytes = p.pref ? (((size_t)nq * sizeof(float) + 255) & ~(size_t)255) : 0;
    p.part16_bytes = p.pref ? (size_t)nq * p.S * pref_outc(p.S, p.QT, k) * sizeof(u64) : 0;
    return p;
}

// does a batch of nq queries take the 256 x 256 shape (what gemm_plan decides for the fp32 path without the first pass)
static bool gemm_big_shape(const rag_index* h, long long nq) {
    static const int force_qt = env_int_g("RAG_GEMM_QT", 0);
    static const int big_min_nq = env_int_g("RAG_GEMM_BIG_MIN_NQ", 1024);
    if (force_qt) return force_qt == 256;
    const bool same_pad = nq > 128 && h->n >= (1 << 20) && (nq + 255) / 256 * 256 == (nq + 127) / 128 * 128;
    return (nq >= big_min_nq || same_pad) && !(!h->prefilter && h->n < (1 << 19) && h->n > 32768);
}

static long long gemm_seed_rows(const rag_index* h, long long nq, int k) {
    static const int on = env_int_g("RAG_GEMM_SEED", 1);            // measurement knobs
    static const long long seed_segs_env = env_int_g("RAG_GEMM_SEED_SEGS", 0);            // > 0: that many segments whatever the shape
    static const long long rows_env = env_int_g("RAG_GEMM_SEED_ROWS", -1);                // k <= 64: 0 off, > 0 that many rows
    static const int min_nq = env_int_g("RAG_GEMM_SEED_MIN_NQ", 5);
    if (!on) return 0;
    if (k > GEMM_MAX_K) {
        if (!(k <= GEMM_BIG_MAX_K && h->n >= (1 << 20) && nq > 64) || h->prefilter) return 0;      // 65+ queries: a wide shape
        // How many segments: a row range of R rows meets about R k / (seeding rows) rows above the seeded bound, and a
        // query's buffer in a range should stay under its trigger (256 keys), so the seeding rows are R k / 128, in whole
        // segments, at most 8 and at most an eighth of the corpus -- they are walked twice.  10M x 768, k = 200: 256 queries
        // (ranges of 39k rows) 4 segments 30.0 ms, 8: 30.2; 1024 queries (156k rows) 4: 119.8, 8: 118.2, 16: 120.7;
        // 3M rows, 1024 queries (47k) 1: 39.3, 2: 36.4, 4: 36.8, 8: 38.2; 1.25M rows, 256 queries (4.9k) 1: 4.80, 8: 5.41.
        const GemmPlan p = gemm_plan(h, nq, k);
        const long long range_rows = (h->n + p.S - 1) / p.S;
        long long segs = (range_rows * k / 128 + SEED_SEG_ROWS - 1) / SEED_SEG_ROWS;
        segs = std::max<long long>(1, std::min<long long>(std::min<long long>(8, segs), h->n / (8 * SEED_SEG_ROWS)));
        if (seed_segs_env > 0) segs = seed_segs_env;
        return segs * SEED_SEG_ROWS;
    }
    // k <= 64: the same rule with this k's trigger (32 / 64 / 128 keys), between 4096 and 16384 rows and at most an eighth
    // of the corpus; every batch this kernel takes (5+ queries) on more rows than the score-matrix path takes (1M x 96 under
    // 32 queries 0.37 -> 0.16 ms; 10M x 768, k = 64, 32 queries 6.41 -> 5.82)
    if (rows_env == 0 || h->n <= SEED_SEG_ROWS || nq < min_nq) return 0;
    const GemmPlan p = gemm_plan(h, nq, k);
    // The bf16 first pass takes the seed only in its 256 x 256 shape (10M x 768: 256 queries 6.67 -> 5.36 ms, 1024 queries
    // 19.2 -> 17.4).  In the narrow shapes a range hands over at most 32..128 keys and relies on the thresholds other ranges
    // publish when they compact; a seeded bound (3 delta below the sample's k-th best) is tight enough to stop the compactions
    // and too loose for that hand-over, so the overflow flag went up and the fp32 kernel ran as well (8 queries 2.4 -> 7.8 ms).
    if (p.pref && p.QT != 256) return 0;
    const long long range_rows = (h->n + p.S - 1) / p.S;
    long long rows = range_rows * k / std::max(1, gemm_trig(p.RT, p.QT, k) / 2);
    rows = std::max<long long>(4096, std::min<long long>(16384, rows));
    rows = std::min<long long>(rows, h->n / 8) / 256 * 256;
    if (rows_env > 0) rows = std::min<long long>(rows_env, SEED_SEG_ROWS);
    return rows;
}

// bytes of workspace the gemm path needs after the padded queries
size_t gemm_topk_workspace(const rag_index* h, long long nq, int k) {
    const GemmPlan p = gemm_plan(h, nq, k);
    return p.cand_bytes + p.part_bytes + p.gthr_bytes + p.qb_bytes + p.margin_bytes + p.part16_bytes + (p.pref ? 256 : 0) +
           (gemm_seeded(h, nq, k) ? gemm_seed_workspace(h, nq, k) : 0);
}

template <int RT, int QT, int NW, bool B16 = false>
static int launch_gemm(const GemmArgs& a, int grid, int device, hipStream_t s) {
    const size_t lds = (size_t)(2 * (RT + QT) * GK) * sizeof(float) + QT * 8 + 16 + (B16 ? QT * 4 : 0);
    static bool attr_set[64] = {};          // per device: the attribute belongs to the device's code object
    if (!attr_set[device & 63]) {
        RAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_topk_kernel<RT, QT, NW, B16>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[device & 63] = true;
    }
    hipLaunchKernelGGL((gemm_topk_kernel<RT, QT, NW, B16>), dim3(grid), dim3(NW * 64), lds, s, a);
    RAG_HIP_CHECK(hipGetLastError());
    return RAG_OK;
}

// bring the bf16 copy of the rows up to date (rows are only ever appended)
static int prefilter_sync_rows(rag_index* h, hipStream_t s) {
    if (h->xb_cap < h->n || !h->Xb) {
        if (h->Xb) RAG_HIP_CHECK(hipFree(h->Xb));
        h->Xb = nullptr;
        h->xb_rows = 0;
        h->xb_cap = std::max(h->cap, h->n);
        RAG_HIP_CHECK(hipMalloc((void**)&h->Xb, (size_t)h->xb_cap * h->ld * sizeof(uint16_t)));
    }
    if (!h->xb_maxnorm) {
        RAG_HIP_CHECK(hipMalloc((void**)&h->xb_maxnorm, 256));
        RAG_HIP_CHECK(hipMemsetAsync(h->xb_maxnorm, 0, 256, s));
    }
    if (h->xb_rows < h->n) {
        long long blocks = (h->n - h->xb_rows + 3) / 4;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, h->X, h->xb_rows, h->n, h->ld,
                           reinterpret_cast<__bf16*>(h->Xb), h->xb_maxnorm);
        RAG_HIP_CHECK(hipGetLastError());
        h->xb_rows = h->n;
    }
    return RAG_OK;
}

int gemm_topk_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask,
                     float* D, long long* I, hipStream_t s) {
    const GemmPlan p = gemm_plan(h, nq, k);
    GemmArgs a = {};
    a.X = h->X;
    a.mask = mask;
    a.n = h->n;
    a.ntiles = p.ntiles;
    a.nq = (int)nq;
    a.ld = h->ld;
    a.k = k;
    a.nqb = p.nqb;
    a.nqb_pad = p.nqb_pad;
    a.S = p.S;
    a.gq = p.gq;
    a.outc = k;
    static const int trig_env = env_int_g("RAG_GEMM_TRIG", 0);          // measurement only
    const int plan_rt = (p.QT == 256 || p.QT == 32) ? 256 : 128;
    a.trig = gemm_trig(plan_rt, p.QT, k);
    if (trig_env > 0 && p.QT >= 128) a.trig = std::min(trig_env, plan_rt);
    const int grid = p.nqb_pad * p.S;
    // workspace: [queries | candidate buffers | partial keys | bf16 queries | margins | kept keys | flag]
    const size_t qbytes = ((size_t)nq * h->ld * sizeof(float) + 255) & ~(size_t)255;
    if (h->ws.bytes < qbytes + gemm_topk_workspace(h, nq, k))
        return fail(RAG_ERR_INVALID, "internal: workspace not sized for the gemm path");
    char* w = static_cast<char*>(h->ws.ptr) + qbytes;
    a.Q = qbuf;
    a.cand = reinterpret_cast<u64*>(w);
    a.part = reinterpret_cast<u64*>(w + p.cand_bytes);
    static const int share_thr = env_int_g("RAG_GEMM_SHARE_THR", 1);
    const bool seeded = gemm_seeded(h, nq, k);
    if ((share_thr && p.S > 1) || seeded) {
        a.gthr = reinterpret_cast<unsigned int*>(w + p.cand_bytes + p.part_bytes);
        RAG_HIP_CHECK(hipMemsetAsync(a.gthr, 0, (size_t)nq * sizeof(unsigned int), s));
    }
    if (seeded) {       // (its score matrix lies behind everything the bf16 first pass keeps in the workspace)
        char* seed_ws = w + p.cand_bytes + p.part_bytes + p.gthr_bytes + p.qb_bytes + p.margin_bytes + p.part16_bytes + (p.pref ? 256 : 0);
        if (int rc = gemm_seed_thresholds(h, nq, qbuf, k, min_score, mask, seed_ws, a.gthr, s)) return rc;
    }
    MergeArgs m = {};
    m.keys = a.part;
    m.per_query = (long long)a.S * k;
    m.k = k;
    m.cap = merge_cap(m.per_query, k);
    m.min_score = min_score;
    m.row_ids = h->ids_dev;
    m.id_base = h->id_base;
    m.rows_out = h->merge_rows_out;
    m.D = D;
    m.I = I;

    if (p.pref) {
        // bf16 pass over bf16 copies, fp32 re-scoring of the kept rows, and -- only if a buffer
        // overflowed -- the fp32 kernel below, which then redoes the whole batch
        if (int rc = prefilter_sync_rows(h, s)) return rc;
        __bf16* qb = reinterpret_cast<__bf16*>(w + p.cand_bytes + p.part_bytes + p.gthr_bytes);
        float* margin = reinterpret_cast<float*>(w + p.cand_bytes + p.part_bytes + p.gthr_bytes + p.qb_bytes);
        u64* part16 = reinterpret_cast<u64*>(w + p.cand_bytes + p.part_bytes + p.gthr_bytes + p.qb_bytes + p.margin_bytes);
        int* flag = reinterpret_cast<int*>(w + p.cand_bytes + p.part_bytes + p.gthr_bytes + p.qb_bytes + p.margin_bytes + p.part16_bytes);
        RAG_HIP_CHECK(hipMemsetAsync(flag, 0, 256, s));
        hipLaunchKernelGGL(queries_to_bf16_kernel, dim3((unsigned)nq), dim3(64), 0, s, qbuf, h->ld, h->xb_maxnorm, qb, margin);
        RAG_HIP_CHECK(hipGetLastError());
        if (seeded) {
            hipLaunchKernelGGL(pref_seed_adjust_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, a.gthr, margin, nq);
            RAG_HIP_CHECK(hipGetLastError());
        }
        GemmArgs b = a;
        b.X = reinterpret_cast<const float*>(h->Xb);
        b.Q = reinterpret_cast<const float*>(qb);
        b.part = part16;
        b.margin = margin;
        b.overflow = flag;
        b.outc = pref_outc(p.S, p.QT, k);
        b.trig = gemm_trig(256, p.QT, k, true);
        if (int rc = profile_begin(h, s)) return rc;
        if (p.QT == 32) { if (int rc = launch_gemm<256, 32, 4, true>(b, grid, h->device, s)) return rc; }
        else if (p.QT == 64) { if (int rc = launch_gemm<128, 64, 4, true>(b, grid, h->device, s)) return rc; }
        else if (int rc = launch_gemm<256, 256, 8, true>(b, grid, h->device, s)) return rc;
        if (int rc = profile_end(h, s)) return rc;
        RescoreArgs r = {};
        r.part = part16;
        r.per_query = p.S * pref_outc(p.S, p.QT, k);
        r.outc = pref_outc(p.S, p.QT, k);
        r.X = h->X;
        r.Q = qbuf;
        r.margin = margin;
        r.ld = h->ld;
        r.overflow = flag;
        r.fin = m;
        {
            constexpr size_t rlds = (size_t)RESC_CAP * sizeof(u64) + 16;
            static bool rattr[64] = {};
            if (!rattr[h->device & 63]) {
                RAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rescore_select_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
                rattr[h->device & 63] = true;
            }
            hipLaunchKernelGGL(rescore_select_kernel, dim3((unsigned)nq), dim3(256), rlds, s, r);
        }
        RAG_HIP_CHECK(hipGetLastError());
        a.run_if = flag;
        m.run_if = flag;
        if (p.QT == 32) { if (int rc = launch_gemm<256, 32, 4>(a, grid, h->device, s)) return rc; }
        else if (p.QT == 64) { if (int rc = launch_gemm<128, 64, 4>(a, grid, h->device, s)) return rc; }
        else if (int rc = launch_gemm<256, 256, 8>(a, grid, h->device, s)) return rc;
        RAG_HIP_CHECK(launch_merge_keys(m, nq, s));
        h->last_path = "gemm-bf16";
        return RAG_OK;
    }

    if (int rc = profile_begin(h, s)) return rc;
    int rc;
    if (p.QT == 32) rc = launch_gemm<256, 32, 4>(a, grid, h->device, s);
    else if (p.QT == 64) rc = launch_gemm<128, 64, 4>(a, grid, h->device, s);
    else if (p.QT == 256) rc = launch_gemm<256, 256, 8>(a, grid, h->device, s);
    else rc = launch_gemm<128, 128, 4>(a, grid, h->device, s);
    if (rc) return rc;
    if (int rc2 = profile_end(h, s)) return rc2;
    RAG_HIP_CHECK(launch_merge_keys(m, nq, s));
    h->last_path = (seeded && k > GEMM_MAX_K) ? "gemm-seeded" : "gemm";     // ("gemm-seeded": the seeding is what keeps k > 64 in this kernel)
    return RAG_OK;
}

// ---------------------------------------------------------------------------------------------
// A large batch over a short corpus (the coarse step of an IVF search: 16384 centroids under 4096
// queries, top-64): a workgroup's row range is a few tiles, far too few rows for a threshold to
// build up, so the in-kernel selection pushes a large share of every tile (2.1 ms for 0.87 ms of
// arithmetic at that shape).  Here the matrix-core kernel writes every score (nq x n fp32: 268 MB at
// that shape, a few tens of microseconds of HBM traffic) and one workgroup per query selects its k
// best with the selection kernel of the few-query path (direct_topk.hip).  Same MFMA chain per
// (row, query) as every other shape, same order rule.
int direct_select_launch(rag_index* h, long long nq, int k, float min_score, const uint8_t* mask, const float* scores, float* D,
                         long long* I, hipStream_t s);      // direct_topk.hip
constexpr long long SCORES_MAX_ROWS = 32768;                 // what one selection workgroup holds in registers

// the shape alone (an IVF index asks this of its quantizer, whose bf16 first pass serves the assignment of new rows only)
bool gemm_scores_shape_ok(const rag_index* h, long long nq, int k) {
    static const int min_nq = env_int_g("RAG_GEMM_SCORES_MIN_NQ", 128);        // 0 switches the path off (measurement knob)
    return min_nq > 0 && h->kind == RAG_KIND_FLAT && (h->ld % GK) == 0 && h->n >= 1024 && h->n <= SCORES_MAX_ROWS &&
           nq >= min_nq && k >= 1 && k <= 1024 && (size_t)nq * (size_t)h->n * sizeof(float) <= ((size_t)2 << 30);
}

bool gemm_scores_supported(const rag_index* h, long long nq, int k) { return !h->prefilter && gemm_scores_shape_ok(h, nq, k); }

size_t gemm_scores_workspace(const rag_index* h, long long nq) { return ((size_t)nq * (size_t)h->n * sizeof(float) + 255) & ~(size_t)255; }

template <int RT, int QT, int NW>
static int launch_gemm_store(const GemmArgs& a, int grid, int device, hipStream_t s) {
    const size_t lds = (size_t)(2 * (RT + QT) * GK) * sizeof(float) + QT * 8 + 16;
    static bool attr_set[64] = {};
    if (!attr_set[device & 63]) {
        RAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_topk_kernel<RT, QT, NW, false, false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[device & 63] = true;
    }
    hipLaunchKernelGGL((gemm_topk_kernel<RT, QT, NW, false, false, true>), dim3(grid), dim3(NW * 64), lds, s, a);
    RAG_HIP_CHECK(hipGetLastError());
    return RAG_OK;
}

int gemm_scores_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask, float* D,
                       long long* I, hipStream_t s) {
    GemmPlan p = gemm_plan(h, nq, k);
    if (p.QT != 256) {                       // 128 .. 1023 queries: the 128 x 128 shape
        p.QT = 128; p.RT = 128; p.NW = 4;
        p.nqb = (int)((nq + 127) / 128);
        p.nqb_pad = p.nqb;                    // plain placement (gq = 0): nothing to pad for
        p.ntiles = (h->n + 127) / 128;
        // at most 512 workgroups (two per compute unit): one more would run alone after the others, for as long again
        p.S = (int)std::max<long long>(1, std::min<long long>(p.ntiles, 512 / p.nqb_pad));
        p.gq = 0;
    }
    const size_t qbytes = ((size_t)nq * h->ld * sizeof(float) + 255) & ~(size_t)255;
    if (h->ws.bytes < qbytes + gemm_scores_workspace(h, nq)) return fail(RAG_ERR_INVALID, "internal: workspace not sized for the score matrix");
    GemmArgs a = {};
    a.X = h->X;
    a.Q = qbuf;
    a.n = h->n;
    a.ntiles = p.ntiles;
    a.nq = (int)nq;
    a.ld = h->ld;
    a.k = k;
    a.nqb = p.nqb;
    a.nqb_pad = p.nqb_pad;
    a.S = p.S;
    a.gq = p.gq;
    a.outc = k;
    a.trig = gemm_trig(p.RT, p.QT, k);
    a.scores = reinterpret_cast<float*>(static_cast<char*>(h->ws.ptr) + qbytes);
    const int grid = p.nqb_pad * p.S;
    if (int rc = profile_begin(h, s)) return rc;
    int rc = p.QT == 256 ? launch_gemm_store<256, 256, 8>(a, grid, h->device, s) : launch_gemm_store<128, 128, 4>(a, grid, h->device, s);
    if (rc) return rc;
    if (int rc2 = profile_end(h, s)) return rc2;
    if (int rc3 = direct_select_launch(h, nq, k, min_score, mask, a.scores, D, I, s)) return rc3;
    h->last_path = "gemm-scores";
    return RAG_OK;
}

// ---------------------------------------------------------------------------------------------
// k above what the selecting kernel carries (64), 2 or more queries, any corpus: the in-kernel candidate buffers do not
// scale to hundreds of keys per query, and the fallback used to be one full scan of the corpus PER QUERY (64 queries,
// k = 100, 10M rows: 64 x 4.6 ms).  Path "gemm-segments" walks the corpus in segments of 32768 rows: the score-writing
// kernel above fills the [queries x segment] score matrix, direct_select_kernel leaves every query's k best of the
// segment (mask, threshold and ids applied there), and once all segments are done the k-way device merge of
// rag_merge_topk_device reduces the per-segment lists -- ties to the lower slot, i.e. the lower row.  Queries go in
// chunks so that the lists of a chunk stay a few hundred MB.  Cost: the matrix-core pass over the corpus plus a few per cent.
int direct_select_rows(long long nq, int n, int k, float min_score, const uint8_t* mask, const long long* row_ids, long long id_base,
                       long long* rows_out, const float* scores, float* D, long long* I, hipStream_t s);     // direct_topk.hip
int merge_lists_device(int nlists, long long nq, int k, const float* D_dev, long long d_list_stride, const long long* I_dev,
                       long long i_list_stride, float* D_out, long long* I_out, u64* keys, hipStream_t s);    // api.hip

// queries per chunk: as many as keep a chunk's per-segment lists and merge keys (20 bytes per hit and segment) within
// about 6 GB -- more queries per launch give every workgroup more tiles of a segment -- between 256 and 4096
static long long seg_query_chunk(long long nseg, int k) {
    static const long long budget = (long long)env_int_g("RAG_GEMM_SEGMENTS_BUDGET_MB", 6144) << 20;      // measurement knob
    long long qc = budget / std::max<long long>(1, nseg * k * 20);
    long long p2 = 256;
    while (p2 * 2 <= qc && p2 < 4096) p2 *= 2;
    return p2;
}

bool gemm_segments_supported(const rag_index* h, long long nq, int k) {
    static const int min_nq = env_int_g("RAG_GEMM_SEGMENTS_MIN_NQ", 2);         // 0 switches the path off (measurement knob)
    return min_nq > 0 && h->kind == RAG_KIND_FLAT && !h->prefilter && (h->ld % GK) == 0 && h->n >= 1024 && nq >= min_nq &&
           k > GEMM_MAX_K && k <= 1024 && h->n <= ((long long)1 << 31) - SCORES_MAX_ROWS;
}

// Segments go to the kernels in groups: one score-writing launch fills the rows of `group` consecutive segments
// ([queries][group * 32768] scores) and one selection launch runs a workgroup per (query, segment) of the group.  A
// small chunk of queries takes large groups, so that both launches have a few hundred workgroups, and the narrow
// query shapes (256 x 32 up to 32 queries, 128 x 64 up to 64), which read the corpus once without multiplying padding
// columns.  Larger chunks take groups too (8192 (query, segment) pairs per launch, at most 32 segments): a workgroup of
// the score launch then walks several tiles, which evens out batch sizes whose query-block count does not divide the
// grid (768 queries: 3 blocks x 85 row ranges over the 128 tiles of one segment cost what 4 blocks cost).
bool narrow_scores_supported(const rag_index* h, long long nq);                                      // narrow_topk.hip
int narrow_scores_launch(const rag_index* h, const float* X, long long rows, long long nq, const float* qbuf, float* scores,
                         hipStream_t s);
struct SegPlan {
    long long nseg, qc, group;
    size_t scores_bytes, d_bytes, i_bytes, keys_bytes;
};
static SegPlan seg_plan(const rag_index* h, long long nq, int k) {
    static const long long max_group = std::max(1, env_int_g("RAG_GEMM_SEGMENTS_GROUP", 32));        // measurement knob
    SegPlan p;
    p.nseg = (h->n + SCORES_MAX_ROWS - 1) / SCORES_MAX_ROWS;
    p.qc = std::min<long long>(nq, seg_query_chunk(p.nseg, k));
    static const long long group_pairs = std::max(1, env_int_g("RAG_GEMM_SEGMENTS_PAIRS", 8192));      // (query, segment) pairs per launch: a score matrix of at most 1 GB
    p.group = std::max<long long>(1, std::min<long long>(std::min<long long>(max_group, p.nseg), group_pairs / p.qc));
    // up to 32 queries on the register-streaming kernel: every launch stages the query block again, so the groups are as
    // long as a score matrix of 512 MB allows (8 queries: the 10M rows of a corpus in one launch)
    static const long long max_group_narrow = std::max(1, env_int_g("RAG_GEMM_SEGMENTS_GROUP_NARROW", 512));      // measurement knob
    if (narrow_scores_supported(h, p.qc)) p.group = std::max<long long>(1, std::min<long long>(std::min<long long>(max_group_narrow, p.nseg), 4096 / p.qc));
    const long long group_rows = std::min<long long>(h->n, p.group * SCORES_MAX_ROWS);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    p.scores_bytes = al((size_t)p.qc * group_rows * sizeof(float));
    // lists of one group of segments behind the running answer (list 0)
    p.d_bytes = al((size_t)(p.group + 1) * p.qc * k * sizeof(float));
    p.i_bytes = al((size_t)(p.group + 1) * p.qc * k * sizeof(long long));
    p.keys_bytes = al((size_t)(p.group + 1) * p.qc * k * sizeof(u64));
    return p;
}
size_t gemm_segments_workspace(const rag_index* h, long long nq, int k) {
    const SegPlan p = seg_plan(h, nq, k);
    return p.scores_bytes + p.d_bytes + p.i_bytes + p.keys_bytes;
}

int direct_select_segments(long long nq, long long n_row, long long seg_rows, int k, float min_score, const uint8_t* mask,
                           const long long* row_ids, long long id_base, const float* scores, float* D, long long* I, long long out_stride,
                           const float* floor, long long floor_stride, hipStream_t s);   // direct_topk.hip
void launch_fill_results(float* D, long long* I, long long n, hipStream_t s);                // api.hip

int gemm_segments_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask, float* D,
                         long long* I, hipStream_t s) {
    const SegPlan p = seg_plan(h, nq, k);
    const size_t qbytes = ((size_t)nq * h->ld * sizeof(float) + 255) & ~(size_t)255;
    if (h->ws.bytes < qbytes + gemm_segments_workspace(h, nq, k)) return fail(RAG_ERR_INVALID, "internal: workspace not sized for the segment walk");
    char* w = static_cast<char*>(h->ws.ptr) + qbytes;
    float* scores = reinterpret_cast<float*>(w);
    float* Dseg = reinterpret_cast<float*>(w + p.scores_bytes);
    long long* Iseg = reinterpret_cast<long long*>(w + p.scores_bytes + p.d_bytes);
    u64* keys = reinterpret_cast<u64*>(w + p.scores_bytes + p.d_bytes + p.i_bytes);
    if (int rc = profile_begin(h, s)) return rc;
    for (long long q0 = 0; q0 < nq; q0 += p.qc) {
        const long long qc = std::min(p.qc, nq - q0);
        // the last chunk of a batch may be shorter than the chunk the group was planned for: its shape follows its own size
        const int QT = qc >= 512 ? 256 : qc > 64 ? 128 : qc > 32 ? 64 : 32;
        const int RT = QT == 32 ? 256 : QT == 64 ? 128 : QT;
        // The running answer of the chunk is list 0 of Dseg / Iseg; a group's segments leave their lists behind it, and one
        // merge per group folds them in (ties to the lower list = the lower row).  Its k-th score is the floor of the next
        // group's selection: after g groups of G segments a segment holds about k / (g G) rows above it instead of k, so
        // its selection ranks a handful of keys directly instead of searching a bound among 32768.
        const size_t lstride = (size_t)qc * k;
        float* Dout = D + (size_t)q0 * k;
        long long* Iout = I + (size_t)q0 * k;
        launch_fill_results(Dseg, Iseg, (long long)lstride, s);
        auto fold = [&](long long sg, long long segs_here) -> int {
            if (int rc = merge_lists_device((int)segs_here + 1, qc, k, Dseg, (long long)lstride, Iseg, (long long)lstride, Dout, Iout, keys, s)) return rc;
            if (sg + segs_here < p.nseg) {       // the answer so far goes back to list 0
                RAG_HIP_CHECK(hipMemcpyAsync(Dseg, Dout, lstride * sizeof(float), hipMemcpyDeviceToDevice, s));
                RAG_HIP_CHECK(hipMemcpyAsync(Iseg, Iout, lstride * sizeof(long long), hipMemcpyDeviceToDevice, s));
            }
            return RAG_OK;
        };
        static const int use_floor = env_int_g("RAG_GEMM_SEGMENTS_FLOOR", 1);        // measurement knob
        for (long long sg = 0; sg < p.nseg; sg += p.group) {
            const long long lo = sg * SCORES_MAX_ROWS, rows = std::min<long long>(p.group * SCORES_MAX_ROWS, h->n - lo);
            const long long segs_here = (rows + SCORES_MAX_ROWS - 1) / SCORES_MAX_ROWS;
            // (a floor only where min_score cannot empty list 0: the floor is the k-th kept score, kept means >= min_score)
            const float* floor = (use_floor && sg > 0) ? Dseg + (k - 1) : nullptr;
            if (narrow_scores_supported(h, qc)) {
                // up to 32 queries: the register-streaming kernel reads the rows at the HBM rate (narrow_topk.hip)
                if (int rc = narrow_scores_launch(h, h->X + (size_t)lo * h->ld, rows, qc, qbuf + (size_t)q0 * h->ld, scores, s)) return rc;
                if (int rc2 = direct_select_segments(qc, rows, SCORES_MAX_ROWS, k, min_score, mask ? mask + lo : nullptr,
                                                     h->ids_dev ? h->ids_dev + lo : nullptr, h->id_base + lo, scores,
                                                     Dseg + lstride, Iseg + lstride, (long long)lstride, floor, k, s))
                    return rc2;
                if (int rc3 = fold(sg, segs_here)) return rc3;
                continue;
            }
            GemmArgs a = {};
            a.X = h->X + (size_t)lo * h->ld;
            a.Q = qbuf + (size_t)q0 * h->ld;
            a.n = rows;
            a.ntiles = (rows + RT - 1) / RT;
            a.nq = (int)qc;
            a.ld = h->ld;
            a.k = k;
            a.nqb = (int)((qc + QT - 1) / QT);
            a.nqb_pad = a.nqb;
            // a grid of at most one (256 x 256) or two workgroups per compute unit: one more would run alone after the others
            a.S = (int)std::max<long long>(1, std::min<long long>(a.ntiles, (QT == 256 ? 256 : 512) / a.nqb_pad));
            a.gq = 0;
            a.outc = k;
            a.trig = gemm_trig(RT, QT, GEMM_MAX_K);
            a.scores = scores;
            const int grid = a.nqb_pad * a.S;
            int rc;
            if (QT == 256) rc = launch_gemm_store<256, 256, 8>(a, grid, h->device, s);
            else if (QT == 128) rc = launch_gemm_store<128, 128, 4>(a, grid, h->device, s);
            else if (QT == 64) rc = launch_gemm_store<128, 64, 4>(a, grid, h->device, s);
            else rc = launch_gemm_store<256, 32, 4>(a, grid, h->device, s);
            if (rc) return rc;
            if (int rc2 = direct_select_segments(qc, rows, SCORES_MAX_ROWS, k, min_score, mask ? mask + lo : nullptr,
                                                 h->ids_dev ? h->ids_dev + lo : nullptr, h->id_base + lo, scores, Dseg + lstride, Iseg + lstride,
                                                 (long long)lstride, floor, k, s))
                return rc2;
            if (int rc3 = fold(sg, segs_here)) return rc3;
        }
    }
    if (int rc = profile_end(h, s)) return rc;
    h->last_path = "gemm-segments";
    return RAG_OK;
}

// ---------------------------------------------------------------------------------------------
// Starting thresholds for the selecting kernel (gemm_seed_rows): the first rows of the corpus go through the score-writing
// kernel, a chunk of queries at a time, and one workgroup per query takes a lower bound of the k-th best score among them: every
// thread keeps the best score of its share of the rows (mask and min_score applied: only rows that may be reported count), and the
// k-th largest of those per-thread maxima has k distinct rows at or above it.  With many more threads than k that is close to
// the k-th best itself (k = 10 over 256 threads: rank 10.2 on average; k = 200 over 1024 threads: rank 223).  Where fewer than k
// threads saw a row the word stays 0 = "no threshold yet".
struct SeedPlan {
    long long qc, rows;
    size_t scores_bytes;
};
static SeedPlan seed_plan(const rag_index* h, long long nq, int k) {
    static_assert(SEED_SEG_ROWS == SCORES_MAX_ROWS, "seeding rows are counted in segments of the selection kernel");
    SeedPlan p;
    p.qc = std::min<long long>(nq, 4096);
    // a score matrix of at most 2 GB
    const long long want = gemm_seed_rows(h, nq, k);
    const long long cap_rows = std::max<long long>(SCORES_MAX_ROWS, ((long long)2 << 30) / (p.qc * (long long)sizeof(float)) / SCORES_MAX_ROWS * SCORES_MAX_ROWS);
    p.rows = std::min<long long>(h->n, std::min(want, cap_rows));
    p.scores_bytes = ((size_t)p.qc * p.rows * sizeof(float) + 255) & ~(size_t)255;
    return p;
}
static size_t gemm_seed_workspace(const rag_index* h, long long nq, int k) { return seed_plan(h, nq, k).scores_bytes; }

template <int NT>
__global__ __launch_bounds__(NT) void seed_heads_kernel(const float* scores, long long rows, const uint8_t* mask, float min_score, int k,
                                                        unsigned int* gthr) {
    __shared__ __attribute__((aligned(16))) unsigned int heads[NT];
    const int tid = threadIdx.x;
    const float* sc = scores + (size_t)blockIdx.x * rows;
    unsigned int best = 0u;
#pragma unroll 8
    for (long long r = tid; r < rows; r += NT) {
        const float v = sc[r];
        if (v >= min_score && (!mask || mask[r])) {
            const unsigned int o = f32_ordered(v);
            best = o > best ? o : best;
        }
    }
    heads[tid] = best;
    __syncthreads();
    if (best == 0u) return;
    int above = 0;                       // maxima that rank before this one (equal ones: the lower thread first)
    for (int j = 0; j < NT; j += 4) {
        const uint4 hj = *reinterpret_cast<const uint4*>(&heads[j]);
        above += (hj.x > best || (hj.x == best && j + 0 < tid)) ? 1 : 0;
        above += (hj.y > best || (hj.y == best && j + 1 < tid)) ? 1 : 0;
        above += (hj.z > best || (hj.z == best && j + 2 < tid)) ? 1 : 0;
        above += (hj.w > best || (hj.w == best && j + 3 < tid)) ? 1 : 0;
    }
    if (above == k - 1) gthr[blockIdx.x] = best;
}

// the bound alone, for any table of [nq][rows] scores
static int seed_thresholds_from_scores(const float* scores, long long rows, long long nq, const uint8_t* mask, float min_score, int k,
                                unsigned int* gthr, hipStream_t s) {
    if (rows <= 16384 && k <= 64)
        hipLaunchKernelGGL(seed_heads_kernel<256>, dim3((unsigned)nq), dim3(256), 0, s, scores, rows, mask, min_score, k, gthr);
    else
        hipLaunchKernelGGL(seed_heads_kernel<1024>, dim3((unsigned)nq), dim3(1024), 0, s, scores, rows, mask, min_score, k, gthr);
    RAG_HIP_CHECK(hipGetLastError());
    return RAG_OK;
}

// gthr: zeroed by the caller
static int gemm_seed_thresholds(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask, char* w,
                                unsigned int* gthr, hipStream_t s) {
    const SeedPlan p = seed_plan(h, nq, k);
    float* scores = reinterpret_cast<float*>(w);
    for (long long q0 = 0; q0 < nq; q0 += p.qc) {
        const long long qc = std::min(p.qc, nq - q0);
        const int QT = qc >= 512 ? 256 : qc > 64 ? 128 : qc > 32 ? 64 : 32;
        const int RT = QT == 32 ? 256 : QT == 64 ? 128 : QT;
        GemmArgs a = {};
        a.X = h->X;
        a.Q = qbuf + (size_t)q0 * h->ld;
        a.n = p.rows;
        a.ntiles = (p.rows + RT - 1) / RT;
        a.nq = (int)qc;
        a.ld = h->ld;
        a.k = k;
        a.nqb = (int)((qc + QT - 1) / QT);
        a.nqb_pad = a.nqb;
        a.S = (int)std::max<long long>(1, std::min<long long>(a.ntiles, (QT == 256 ? 256 : 512) / a.nqb_pad));
        a.gq = 0;
        a.outc = k;
        a.trig = gemm_trig(RT, QT, GEMM_MAX_K);
        a.scores = scores;
        const int grid = a.nqb_pad * a.S;
        int rc;
        if (QT == 256) rc = launch_gemm_store<256, 256, 8>(a, grid, h->device, s);
        else if (QT == 128) rc = launch_gemm_store<128, 128, 4>(a, grid, h->device, s);
        else if (QT == 64) rc = launch_gemm_store<128, 64, 4>(a, grid, h->device, s);
        else rc = launch_gemm_store<256, 32, 4>(a, grid, h->device, s);
        if (rc) return rc;
        if (int rc2 = seed_thresholds_from_scores(scores, p.rows, qc, mask, min_score, k, gthr + q0, s)) return rc2;
    }
    return RAG_OK;
}

// ---------------------------------------------------------------------------------------------
// List-major IVF scan for large batches.  After the coarse step every query has nprobe lists; with
// many queries a list is probed by several of them, and scanning per query would read it once for
// each.  Here the (query, probe) pairs are grouped by list on the device, a work item is "up to 32
// queries of one list", and the 256 x 32 MFMA shape scores the list's rows against them: a list
// is read once per 32 probing queries.  Every pair's k keys go to its own slot (query * nprobe +
// probe), and the usual merge reduces a query's nprobe slots.
__global__ void lm_count_kernel(const long long* lists, long long npairs, int* cnt) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npairs; i += (long long)gridDim.x * blockDim.x) {
        const long long l = lists[i];
        if (l >= 0) atomicAdd(&cnt[l], 1);
    }
}
// one workgroup: exclusive scans of the pair counts and of the work-item counts, work-item table
__global__ __launch_bounds__(1024) void lm_plan_kernel(const int* cnt, int nlist, const long long* list_begin, int* pair_begin,
                                                       int* task_begin, int* task_list, int* total_tasks) {
    __shared__ int sp[1024], st[1024];
    __shared__ int carry_p, carry_t;
    const int tid = threadIdx.x;
    if (tid == 0) { carry_p = 0; carry_t = 0; }
    __syncthreads();
    for (int base = 0; base < nlist; base += 1024) {
        const int l = base + tid;
        int c = 0, t = 0;
        if (l < nlist) {
            c = cnt[l];
            t = (list_begin[l + 1] > list_begin[l]) ? (c + 31) / 32 : 0;     // an empty list needs no work
        }
        sp[tid] = c;
        st[tid] = t;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {          // inclusive scan
            const int ap = tid >= off ? sp[tid - off] : 0, at = tid >= off ? st[tid - off] : 0;
            __syncthreads();
            sp[tid] += ap;
            st[tid] += at;
            __syncthreads();
        }
        const int pb = carry_p + sp[tid] - c, tb = carry_t + st[tid] - t;
        if (l < nlist) {
            pair_begin[l] = pb;
            task_begin[l] = tb;
            for (int g = 0; g < t; ++g) task_list[tb + g] = l;
        }
        __syncthreads();
        if (tid == 1023) { carry_p += sp[1023]; carry_t += st[1023]; }
        __syncthreads();
    }
    if (tid == 0) *total_tasks = carry_t;
}
__global__ void lm_fill_kernel(const long long* lists, long long npairs, int nprobe, const int* pair_begin, int* fill,
                               int* pair_q, int* pair_slot) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npairs; i += (long long)gridDim.x * blockDim.x) {
        const long long l = lists[i];
        if (l < 0) continue;
        const int pos = pair_begin[l] + atomicAdd(&fill[l], 1);
        pair_q[pos] = (int)(i / nprobe);
        pair_slot[pos] = (int)i;
    }
}

constexpr int LM_GRID = 512;      // persistent workgroups of the list-major scan (2 per CU)

// Starting thresholds of the list scan.  A (query, list) pair begins with an open threshold, so its first rows are all pushed
// and sorted -- 64 times per query, although nearly every hit comes from the query's nearest lists.  One workgroup per query
// scores the first rows of the query's first list and publishes a lower bound of their k-th best score in the query's shared
// threshold word, which every work item reads when it starts.  The bound must hold for the scores the scan kernel computes (on
// the matrix cores, in another summation order): two fp32 sums of the same `ld` products differ by at most
// 2 ld 2^-24 sum|x_i q_i|, which is subtracted.  Rows a mask removes do not count.
constexpr int IVF_SEED_MAX = 64;
__global__ __launch_bounds__(256) void ivf_seed_kernel(const float* X, const float* Q, const long long* coarse_lists, int nprobe,
                                                       const long long* list_begin, const long long* row_alias, const uint8_t* mask,
                                                       const uint8_t* mask2, int ld, int k, int rows_max, unsigned int* gthr) {
    __shared__ unsigned int sc[IVF_SEED_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long q = blockIdx.x;
    const long long l = coarse_lists[q * nprobe];
    if (l < 0) return;
    const long long lo = list_begin[l];
    const long long len = list_begin[l + 1] - lo;
    const int n = (int)(len < rows_max ? len : rows_max);
    if (n < k) return;
    const f4* q4 = reinterpret_cast<const f4*>(Q + (size_t)q * ld);
    const int ld4 = ld / 4;
    // 16 lanes per row, 16 rows per pass of the workgroup: the loads of a lane are independent of each other
    const int sub = lane & 15;
    for (int r = wave * 4 + (lane >> 4); r < ((n + 15) & ~15); r += 16) {
        const long long row = lo + (r < n ? r : n - 1);
        const f4* x4 = reinterpret_cast<const f4*>(X + (size_t)row * ld);
        float dot = 0.f, mag = 0.f;
#pragma unroll 4
        for (int c = sub; c < ld4; c += 16) {
            const f4 x = x4[c], w = q4[c];
            dot = fmaf(x[0], w[0], dot); mag += fabsf(x[0] * w[0]);
            dot = fmaf(x[1], w[1], dot); mag += fabsf(x[1] * w[1]);
            dot = fmaf(x[2], w[2], dot); mag += fabsf(x[2] * w[2]);
            dot = fmaf(x[3], w[3], dot); mag += fabsf(x[3] * w[3]);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            dot += __shfl_xor(dot, off, 64);
            mag += __shfl_xor(mag, off, 64);
        }
        if (sub == 0 && r < n) {
            const long long shown = row_alias ? row_alias[row] : row;        // masks count rows in insertion order
            const bool ok = (!mask || mask[shown]) && (!mask2 || mask2[shown]);
            const float v = dot - 1.5e-7f * (float)ld * mag;
            sc[r] = (ok && v == v) ? f32_ordered(v) : 0u;
        }
    }
    __syncthreads();
    if (tid < n) {
        const unsigned int mine = sc[tid];
        if (mine != 0u) {
            int above = 0;
            for (int j = 0; j < n; ++j) above += (sc[j] > mine || (sc[j] == mine && j < tid)) ? 1 : 0;
            if (above == k - 1) gthr[q] = mine;
        }
    }
}

static size_t lm_align(size_t b) { return (b + 255) & ~(size_t)255; }
// candidate buffers of whichever of the two list-scan kernels runs
static size_t lm_cand_bytes() { return std::max((size_t)LM_GRID * 32 * gemm_capq(256, 32) * sizeof(u64), lm_align(ivf_lists_cand_bytes())); }

bool ivf_listmajor_supported(const rag_index* h, long long nq, int k, int nprobe) {
    static const int on = env_int_g("RAG_IVF_LISTMAJOR", 1);
    // worth it once a list is probed by more than one query on average: (query, probe) pairs per list,
    // in halves (measurement knob; default 1 pair per list: 256 queries x 64 probes over 16384 lists 9.2 -> 8.9 ms, half of that is slower)
    static const int min_pairs_x2 = std::max(1, env_int_g("RAG_IVF_LISTMAJOR_PAIRS_X2", 2));
    const long long npairs = nq * (long long)nprobe;
    return on && k <= GEMM_MAX_K && (h->ld % GK) == 0 && 2 * npairs >= (long long)min_pairs_x2 * h->ivf.nlist && npairs < (1LL << 30);
}

// bytes after the padded queries: [4 x nlist ints | total | work-item table | pair_q | pair_slot | cand | part]
size_t ivf_listmajor_workspace(const rag_index* h, long long nq, int k, int nprobe) {
    const size_t nlist = (size_t)h->ivf.nlist, npairs = (size_t)nq * nprobe;
    return 4 * lm_align(nlist * 4) + 256 + lm_align((std::min(nlist, npairs) + npairs / 32 + 64) * 4) + 2 * lm_align(npairs * 4) +
           lm_cand_bytes() + lm_align(npairs * k * sizeof(u64)) + lm_align((size_t)nq * 4);
}

int ivf_listmajor_scan(rag_index* h, long long nq, const float* qbuf, int k, float min_score, int nprobe,
                       const long long* coarse_lists, const uint8_t* mask, const uint8_t* mask2, float* D, long long* I, hipStream_t s) {
    const IvfState& v = h->ivf;
    const size_t nlist = (size_t)v.nlist, npairs = (size_t)nq * nprobe;
    const size_t qbytes = ((size_t)nq * h->ld * sizeof(float) + 255) & ~(size_t)255;
    if (h->ws.bytes < qbytes + ivf_listmajor_workspace(h, nq, k, nprobe))
        return fail(RAG_ERR_INVALID, "internal: workspace not sized for the list-major scan");
    char* w = static_cast<char*>(h->ws.ptr) + qbytes;
    int* cnt = reinterpret_cast<int*>(w);                w += lm_align(nlist * 4);
    int* fill = reinterpret_cast<int*>(w);               w += lm_align(nlist * 4);
    int* pair_begin = reinterpret_cast<int*>(w);         w += lm_align(nlist * 4);
    int* task_begin = reinterpret_cast<int*>(w);         w += lm_align(nlist * 4);
    int* total = reinterpret_cast<int*>(w);              w += 256;
    int* task_list = reinterpret_cast<int*>(w);          w += lm_align((std::min(nlist, npairs) + npairs / 32 + 64) * 4);
    int* pair_q = reinterpret_cast<int*>(w);             w += lm_align(npairs * 4);
    int* pair_slot = reinterpret_cast<int*>(w);          w += lm_align(npairs * 4);
    u64* cand = reinterpret_cast<u64*>(w);               w += lm_cand_bytes();
    u64* part = reinterpret_cast<u64*>(w);               w += lm_align(npairs * k * sizeof(u64));
    unsigned int* gthr = reinterpret_cast<unsigned int*>(w);
    static const int share_thr = env_int_g("RAG_IVF_SHARE_THR", 1);                // measurement knob
    RAG_HIP_CHECK(hipMemsetAsync(gthr, 0, (size_t)nq * 4, s));
    RAG_HIP_CHECK(hipMemsetAsync(cnt, 0, 2 * lm_align(nlist * 4), s));          // cnt and fill
    RAG_HIP_CHECK(hipMemsetAsync(total, 0, 256, s));                            // [0] work items (lm_plan_kernel), [1] the work queue's head
    RAG_HIP_CHECK(hipMemsetAsync(part, 0, npairs * k * sizeof(u64), s));        // pairs without rows stay empty
    const int pb = (int)std::min<size_t>((npairs + 255) / 256, 4096);
    hipLaunchKernelGGL(lm_count_kernel, dim3(pb), dim3(256), 0, s, coarse_lists, (long long)npairs, cnt);
    hipLaunchKernelGGL(lm_plan_kernel, dim3(1), dim3(1024), 0, s, cnt, (int)nlist, v.list_begin_dev, pair_begin, task_begin, task_list, total);
    hipLaunchKernelGGL(lm_fill_kernel, dim3(pb), dim3(256), 0, s, coarse_lists, (long long)npairs, nprobe, pair_begin, fill, pair_q, pair_slot);
    RAG_HIP_CHECK(hipGetLastError());
    MergeArgs m = {};
    m.keys = part;
    m.per_query = (long long)nprobe * k;
    m.k = k;
    m.cap = merge_cap(m.per_query, k);
    m.min_score = min_score;
    m.row_ids = h->ids_dev;
    m.id_base = h->id_base;
    m.rows_out = h->merge_rows_out;
    m.D = D;
    m.I = I;
    // starting thresholds (either scan kernel reads a query's shared word when a work item starts / once per tile)
    static const int seed_rows = std::min(IVF_SEED_MAX, env_int_g("RAG_IVF_SEED_ROWS", 32));      // 0 switches the seeding off (measurement knob)
    if (seed_rows > 0 && share_thr && k <= IVF_SEED_MAX) {
        hipLaunchKernelGGL(ivf_seed_kernel, dim3((unsigned)nq), dim3(256), 0, s, h->X, qbuf, coarse_lists, nprobe, v.list_begin_dev, v.orig_dev,
                           mask, mask2, h->ld, k, std::max(seed_rows, std::min(IVF_SEED_MAX, 2 * k)), gthr);
        RAG_HIP_CHECK(hipGetLastError());
    }
    if (ivf_lists_kernel_ok(h, k)) {
        // the register-streaming kernel (ivf_lists.hip): 16-query column blocks, work items from a device queue
        ListScanArgs ls = {};
        ls.X = h->X;
        ls.Q = qbuf;
        ls.mask = mask;            // both masks count rows in insertion order, like the keys
        ls.mask2 = mask2;
        ls.cand = cand;
        ls.part = part;
        ls.gthr = gthr;
        ls.counters = total;
        ls.task_list = task_list;
        ls.task_begin = task_begin;
        ls.pair_begin = pair_begin;
        ls.pair_cnt = cnt;
        ls.pair_q = pair_q;
        ls.pair_slot = pair_slot;
        ls.list_begin = v.list_begin_dev;
        ls.row_alias = v.orig_dev;
        ls.ld = h->ld;
        ls.k = k;
        if (int rc = profile_begin(h, s)) return rc;
        if (int rc = ivf_lists_launch(ls, s)) return rc;
        if (int rc = profile_end(h, s)) return rc;
        RAG_HIP_CHECK(launch_merge_keys(m, nq, s));
        return RAG_OK;
    }
    GemmArgs a = {};
    a.X = h->X;
    a.Q = qbuf;
    a.gthr = share_thr ? gthr : nullptr;
    a.mask = mask;                 // both masks count rows in insertion order, like the keys
    a.mask2 = mask2;
    a.mask_aliased = 1;
    a.cand = cand;
    a.part = part;
    a.n = h->n;
    a.nq = (int)nq;
    a.ld = h->ld;
    a.k = k;
    a.outc = k;
    a.trig = gemm_trig(256, 32, k);
    a.total_tasks = total;
    a.task_list = task_list;
    a.task_begin = task_begin;
    a.pair_begin = pair_begin;
    a.pair_cnt = cnt;
    a.pair_q = pair_q;
    a.pair_slot = pair_slot;
    a.list_begin = v.list_begin_dev;
    a.row_alias = v.orig_dev;
    if (int rc = profile_begin(h, s)) return rc;
    {
        constexpr size_t lds = (size_t)(2 * (256 + 32) * GK) * sizeof(float) + 32 * 8 + 16;
        static bool attr_set[64] = {};
        if (!attr_set[h->device & 63]) {
            RAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_topk_kernel<256, 32, 4, false, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set[h->device & 63] = true;
        }
        hipLaunchKernelGGL((gemm_topk_kernel<256, 32, 4, false, true>), dim3(LM_GRID), dim3(256), lds, s, a);
        RAG_HIP_CHECK(hipGetLastError());
    }
    if (int rc = profile_end(h, s)) return rc;
    RAG_HIP_CHECK(launch_merge_keys(m, nq, s));
    return RAG_OK;
}

// ---------------------------------------------------------------------------------------------
// bf16 first pass for ONE query: the scan kernel over the bf16 copy of the rows (half the HBM
// bytes) collects the PRE1 best approximate scores; verify_rescore_kernel re-scores those rows in
// fp32 and accepts the k best exact ones only if the k-th of them beats (worst kept approximate
// score + delta), delta >= |approximate - exact| for every row: then no row outside the kept set
// can reach the top k.  Otherwise it raises the flag and the fp32 scan enqueued behind it
// (run_if = flag) answers.  The query stays fp32, so delta = pref_err1(ld) * max row norm * |q|.
constexpr int PRE1_MAX = 1024;
// rows kept by the approximate scan for a request of k: 64 (register lists) up to k = 16, 1024
// (LDS lists) up to k = 256
static int pre1_for(int k) { return k <= 16 ? 64 : 1024; }
// rows rounded to bf16 (2^-9), the query exact, ld fp32 additions in the worst case
__host__ __device__ inline float pref_err1(int ld) { return 0.001953125f + 1.01f * (float)ld * 5.9604645e-8f; }

int scan_blocks(const rag_index* h, int k);     // api.hip

struct VerifyArgs {
    const float* Dk;        // [npre] approximate scores, descending, -FLT_MAX padded
    const long long* Ik;    // [npre] row positions, -1 padded
    int npre;               // 64 or 1024 (a power of two)
    const float* X;
    const float* q;         // fp32 padded query
    const int* maxnorm_bits;
    int ld;
    int* flag;
    MergeArgs fin;
};

__global__ __launch_bounds__(1024) void verify_rescore_kernel(VerifyArgs v) {
    __shared__ u64 keys[PRE1_MAX];
    __shared__ float s_qn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int PRE1 = v.npre, nwaves = blockDim.x >> 6;
    const int k = v.fin.k;
    const f4* q4 = reinterpret_cast<const f4*>(v.q);
    const int ld4 = v.ld / 4;
    if (wave == 0) {
        float ss = 0.f;
        for (int c = lane; c < ld4; c += 64) {
            const f4 y = q4[c];
            ss = fmaf(y.x, y.x, ss);
            ss = fmaf(y.y, y.y, ss);
            ss = fmaf(y.z, y.z, ss);
            ss = fmaf(y.w, y.w, ss);
        }
        ss = wave_sum_to_lane63(ss);
        if (lane == 63) s_qn = sqrtf(ss);
    }
    for (int i = wave; i < PRE1; i += nwaves) {
        const long long row = v.Ik[i];
        u64 key = KEY_EMPTY;
        if (row >= 0) {
            const f4* x4 = reinterpret_cast<const f4*>(v.X + (size_t)row * v.ld);
            float acc = 0.f;
            for (int c = lane; c < ld4; c += 64) {
                const f4 x = x4[c], y = q4[c];
                acc = fmaf(x.x, y.x, acc);
                acc = fmaf(x.y, y.y, acc);
                acc = fmaf(x.z, y.z, acc);
                acc = fmaf(x.w, y.w, acc);
            }
            acc = wave_sum_to_lane63(acc);
            key = make_key(__shfl(acc, 63, 64), (uint32_t)row);
        }
        if (lane == 0) keys[i] = key;
    }
    __syncthreads();
    bitonic_sort_desc_lds(keys, PRE1, tid, blockDim.x, SyncBlock());
    // the kept set is everything when the list is not full; else rows outside score at most Dk[PRE1-1] + delta
    const bool full = v.Ik[PRE1 - 1] >= 0;
    const float delta = pref_err1(v.ld) * __int_as_float(*v.maxnorm_bits) * s_qn * 1.0001f + 1e-30f;
    const bool enough = keys[k - 1] != KEY_EMPTY;
    const bool ok = !full || (enough && key_score(keys[k - 1]) > v.Dk[PRE1 - 1] + delta);
    if (tid == 0) *v.flag = ok ? 0 : 1;
    if (ok)
        for (int j = tid; j < k; j += blockDim.x) finalise_scored(v.fin, 0, j, keys[j]);
}

static bool b16_scan_width(int ld) {
    if (ld % 8) return false;
    const int c = ld / 8;       // 16-byte chunks per bf16 row: the widths the scan kernel knows at compile time
    return c == 32 || c == 64 || c == 96 || c == 128 || c == 192 || c == 256 || c == 384 || c == 768;
}

bool prefilter_scan_supported(const rag_index* h, long long nq, int k) {
    return h->prefilter && h->kind == RAG_KIND_FLAT && nq == 1 && k <= 256 && b16_scan_width(h->ld) && h->n >= 65536;
}

// after the padded query: [fp32 scan partial lists | bf16 scan partial lists | PRE1 scores | PRE1 rows | flag]
size_t prefilter_scan_workspace(const rag_index* h, int k) {
    const size_t blocks = (size_t)scan_blocks(h, k);
    return 4 * blocks * k * sizeof(u64) + 256 + 4 * blocks * PRE1_MAX * sizeof(u64) + 256 + PRE1_MAX * 4 + PRE1_MAX * 8 + 256;
}

int prefilter_scan_search(rag_index* h, const float* qbuf, int k, float min_score, const uint8_t* mask,
                          float* D, long long* I, hipStream_t s) {
    if (int rc = prefilter_sync_rows(h, s)) return rc;
    const int blocks = scan_blocks(h, k);
    const size_t qbytes = ((size_t)h->ld * sizeof(float) + 255) & ~(size_t)255;
    if (h->ws.bytes < qbytes + prefilter_scan_workspace(h, k))
        return fail(RAG_ERR_INVALID, "internal: workspace not sized for the bf16 scan");
    char* w = static_cast<char*>(h->ws.ptr) + qbytes;
    const int PRE1 = pre1_for(k);
    u64* part32 = reinterpret_cast<u64*>(w);
    w += (((size_t)4 * blocks * k * sizeof(u64)) + 255) & ~(size_t)255;
    u64* part16 = reinterpret_cast<u64*>(w);
    w += (((size_t)4 * blocks * PRE1_MAX * sizeof(u64)) + 255) & ~(size_t)255;
    float* Dk = reinterpret_cast<float*>(w);
    long long* Ik = reinterpret_cast<long long*>(w + PRE1_MAX * 4);
    int* flag = reinterpret_cast<int*>(w + PRE1_MAX * 4 + PRE1_MAX * 8);

    // 1. approximate scan over the bf16 rows, PRE1 best (score, row position)
    ScanArgs a = {};
    a.X = reinterpret_cast<const float*>(h->Xb);
    a.Q = qbuf;
    a.mask = mask;
    a.part = part16;
    a.n = h->n;
    a.d4 = h->ld / 8;
    a.k = PRE1;
    a.ivf_slices = 1;
    a.grid_y = 1;
    a.nq_valid = 1;
    a.b16 = 1;
    a.nt = ((size_t)h->n * h->ld * 2 > (192ull << 20)) ? 1 : 0;
    if (int rc = profile_begin(h, s)) return rc;
    RAG_HIP_CHECK(launch_flat_scan(a, 1, blocks, s));
    if (int rc = profile_end(h, s)) return rc;
    MergeArgs m = {};
    m.keys = part16;
    m.per_query = (long long)blocks * scan_lists_per_block(PRE1) * PRE1;
    m.k = PRE1;
    m.cap = merge_cap(m.per_query, PRE1);
    m.min_score = -FLT_MAX;
    m.D = Dk;
    m.I = Ik;
    RAG_HIP_CHECK(launch_merge_keys(m, 1, s));
    // 2. exact scores of the kept rows, acceptance test, result
    MergeArgs fin = {};
    fin.k = k;
    fin.min_score = min_score;
    fin.row_ids = h->ids_dev;
    fin.id_base = h->id_base;
    fin.rows_out = h->merge_rows_out;
    fin.D = D;
    fin.I = I;
    VerifyArgs v = {};
    v.Dk = Dk;
    v.Ik = Ik;
    v.X = h->X;
    v.q = qbuf;
    v.maxnorm_bits = h->xb_maxnorm;
    v.ld = h->ld;
    v.flag = flag;
    v.fin = fin;
    v.npre = PRE1;
    hipLaunchKernelGGL(verify_rescore_kernel, dim3(1), dim3(PRE1 == 64 ? 256 : 1024), 0, s, v);
    RAG_HIP_CHECK(hipGetLastError());
    // 3. the fp32 scan, which runs only if the test failed
    ScanArgs f = {};
    f.X = h->X;
    f.Q = qbuf;
    f.mask = mask;
    f.part = part32;
    f.n = h->n;
    f.d4 = h->ld / 4;
    f.k = k;
    f.ivf_slices = 1;
    f.grid_y = 1;
    f.nq_valid = 1;
    f.nt = ((size_t)h->n * h->ld * sizeof(float) > (192ull << 20)) ? 1 : 0;
    f.run_if = flag;
    RAG_HIP_CHECK(launch_flat_scan(f, 1, blocks, s));
    MergeArgs m2 = fin;
    m2.keys = part32;
    m2.per_query = (long long)blocks * scan_lists_per_block(k) * k;
    m2.cap = merge_cap(m2.per_query, k);
    m2.run_if = flag;
    RAG_HIP_CHECK(launch_merge_keys(m2, 1, s));
    h->last_path = "scan-bf16";
    return RAG_OK;
}

}  // namespace rag
