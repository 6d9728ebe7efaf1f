// Batched exhaustive search: S = X * Q^T on the fp32 matrix cores, fused with
// top-k selection so that no score is ever written to memory (gfx950).
//
// Roofline: fp32 MFMA (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD).  Algorithmic
// work per launch = 2 * nq * n * ld flops.  Scores are bit-for-bit a k-ordered
// fp32 fma chain, the same arithmetic a CPU inner product performs.
//
// Tiling.  A workgroup (4 waves) owns a block of 128 queries and a contiguous
// range of corpus rows, which it walks in tiles of 128 rows.  Per tile it runs
// the K loop in steps of 32 through LDS (double buffered, one barrier per step;
// the next step's global loads are in flight while the current step computes).
// Each wave holds a 64 x 64 piece of the 128 x 128 score tile as 2 x 2 MFMA
// accumulators.  Corpus rows sit on the MFMA's row index (registers), queries on
// its column index (lanes): a lane owns two queries and sees 2 x 32 rows of each.
//
// LDS image.  [128 rows][32 + 4 floats]: the 4-float pad makes the 16-byte
// fragment reads of any 16 consecutive rows hit 16 different bank groups.  A
// lane reads k = 8*kb + 4*h + s (h = lane >> 5, s = 0..3) as one ds_read_b128 and
// feeds element s to MFMA step s; both operands use the same k order, so the sum
// over (h, s) covers the 8 k's of the block exactly once.
//
// Selection.  Per query the workgroup keeps a threshold (LDS) and a candidate
// buffer of 256 keys (global memory, private to the workgroup, L2 resident).  A
// score enters the buffer only if it beats the threshold, which after the first
// few tiles is rare.  When a query's buffer could overflow on the next tile
// (count > 256 - 128) one wave sorts it in LDS, keeps k and tightens the
// threshold.  At the end every query's k best keys go to `part`, and
// merge_keys_kernel merges the row-range partials.
//
// Placement.  blockIdx -> (query block, row range) is chosen so that the
// workgroups of one XCD (blockIdx % 8) share few query blocks: their query tiles
// stay in that XCD's L2 while the corpus streams through.
#include "index.h"
#include "topk_keys.h"

namespace rag {

int merge_cap(long long per_query, int k);

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int GT = 128;          // tile edge: rows and queries
constexpr int GK = 32;           // K step
constexpr int GLD = GK + 4;      // padded LDS row stride (floats)
constexpr int CAPQ = 256;        // candidate keys per query per workgroup
constexpr int GEMM_MAX_K = 64;

// Timing-only ablations for diagnostic builds (outputs are wrong when non-zero):
// bit 0 skip the tile epilogue, bit 1 load global tiles only once, bit 2 stage LDS only twice.
#ifndef RAG_GEMM_DIAG
#define RAG_GEMM_DIAG 0
#endif
constexpr int DIAG = RAG_GEMM_DIAG;

struct GemmArgs {
    const float* X;
    const float* Q;
    const uint8_t* mask;
    u64* cand;       // [grid][GT][CAPQ]
    u64* part;       // [nq][S][k]
    long long n;
    long long ntiles;
    int nq;
    int ld;
    int k;
    int nqb;         // query blocks
    int nqb_pad;     // query blocks rounded up for the XCD-friendly decode
    int S;           // row ranges
};

// rare path of the tile epilogue, kept out of line so that the 64 call sites stay small
__device__ __attribute__((noinline)) void push_candidate(const GemmArgs& a, u64* cand, int* s_cnt, int* s_flag, int ql,
                                                         long long row, float sc) {
    if (row < a.n && (!a.mask || a.mask[row])) {
        const int pos = atomicAdd(&s_cnt[ql], 1);
        cand[(size_t)ql * CAPQ + pos] = make_key(sc, (uint32_t)row);
        if (pos + 1 > CAPQ - GT) *s_flag = 1;       // compact before the next tile can overflow it
    }
}

__global__ __launch_bounds__(256, 2) void gemm_topk_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* As = reinterpret_cast<float*>(smem);                       // [2][GT*GLD] corpus rows
    float* Bs = As + 2 * GT * GLD;                                     // [2][GT*GLD] queries
    float* s_thr = Bs + 2 * GT * GLD;                                  // [GT]
    int* s_cnt = reinterpret_cast<int*>(s_thr + GT);                   // [GT]
    int* s_flag = s_cnt + GT;                                          // [1] (+3 pad)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1;        // which 64 corpus rows of the tile
    const int wq = wave & 1;         // which 64 queries of the block
    const int l31 = lane & 31;
    const int h = lane >> 5;

    const int b = blockIdx.x;
    const int qb = b % a.nqb_pad;
    const int sp = b / a.nqb_pad;
    if (qb >= a.nqb) return;
    const int q0 = qb * GT;
    const int k = a.k;
    const int KT = a.ld / GK;
    const long long t_begin = a.ntiles * sp / a.S;
    const long long t_end = a.ntiles * (sp + 1) / a.S;
    u64* cand = a.cand + (size_t)b * GT * CAPQ;

    for (int i = tid; i < GT; i += 256) { s_thr[i] = -INFINITY; s_cnt[i] = 0; }
    if (tid == 0) *s_flag = 0;
    __syncthreads();

    // ---- loader: thread -> 4 x (row, 16-byte column) of each operand tile ------------
    const f4* X4 = reinterpret_cast<const f4*>(a.X);
    const f4* Q4 = reinterpret_cast<const f4*>(a.Q);
    const int ld4 = a.ld / 4;
    const int lc4 = tid & 7;
    const int lrow0 = tid >> 3;                         // rows lrow0 + 32*i, i = 0..3
    const f4* qp[4];                                    // this thread's query chunks, k step 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int qr = q0 + lrow0 + 32 * i;
        qr = qr < a.nq ? qr : a.nq - 1;
        qp[i] = Q4 + (long long)qr * ld4 + lc4;
    }
    const f4* xp[4];                                    // this thread's corpus chunks of the tile being loaded
    auto set_tile = [&](long long t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long r = t * GT + lrow0 + 32 * i;
            r = r < a.n ? r : a.n - 1;
            xp[i] = X4 + r * ld4 + lc4;
        }
    };
    f4 xr[4], qr4[4];
    auto issue_load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xr[i] = xp[i][kt * (GK / 4)];
            qr4[i] = qp[i][kt * (GK / 4)];
        }
    };
    const int lds_w = lrow0 * GLD + lc4 * 4;            // float offset of this thread's first staged chunk
    auto store_lds = [&](int buf) {
        float* Ab = As + buf * GT * GLD + lds_w;
        float* Bb = Bs + buf * GT * GLD + lds_w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f4*>(Ab + i * 32 * GLD) = xr[i];
            *reinterpret_cast<f4*>(Bb + i * 32 * GLD) = qr4[i];
        }
    };
    // one wave sorts the candidate buffer of every query that could overflow on the next
    // tile; `scratch_buf` is the LDS stage buffer nobody reads or writes right now
    auto compact = [&](int scratch_buf) {
        u64* scratch = reinterpret_cast<u64*>(As + scratch_buf * GT * GLD) + wave * CAPQ;
        for (int ql = wave; ql < GT; ql += 4) {
            const int c = s_cnt[ql];
            if (c <= CAPQ - GT) continue;            // wave-uniform
            for (int i = lane; i < CAPQ; i += 64) scratch[i] = (i < c) ? cand[(size_t)ql * CAPQ + i] : KEY_EMPTY;
            bitonic_sort_desc_lds(scratch, CAPQ, lane, 64, SyncWave());
            for (int i = lane; i < k; i += 64) cand[(size_t)ql * CAPQ + i] = scratch[i];
            if (lane == 0) {
                s_cnt[ql] = c < k ? c : k;
                if (c >= k) s_thr[ql] = key_score(scratch[k - 1]);
            }
            SyncWave()();
        }
    };

    f16v acc[2][2];
    int buf = 0;
    if (t_begin < t_end) { set_tile(t_begin); issue_load(0); }
    for (long long t = t_begin; t < t_end; ++t) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;
        for (int kt = 0; kt < KT; ++kt, buf ^= 1) {
            if (!(DIAG & 4) || (t == t_begin && kt < 2)) {
                store_lds(buf);
                __syncthreads();
            }
            // a candidate buffer filled up during the previous tile's epilogue: every wave has
            // passed the barrier above, so the flag is stable and buffer buf^1 is idle
            if (kt == 0 && *s_flag != 0) {
                compact(buf ^ 1);
                __syncthreads();
                if (tid == 0) *s_flag = 0;
            }
            if (!(DIAG & 2) || (t == t_begin && kt == 0)) {
                if (kt + 1 < KT) issue_load(kt + 1);
                else if (t + 1 < t_end) { set_tile(t + 1); issue_load(0); }
            }
            const float* Ab = As + buf * GT * GLD + (wm * 64 + l31) * GLD + h * 4;
            const float* Bb = Bs + buf * GT * GLD + (wq * 64 + l31) * GLD + h * 4;
            // fragments of k block kb+1 are requested before the MFMAs of block kb issue
            f4 fa0 = *reinterpret_cast<const f4*>(Ab);
            f4 fa1 = *reinterpret_cast<const f4*>(Ab + 32 * GLD);
            f4 fb0 = *reinterpret_cast<const f4*>(Bb);
            f4 fb1 = *reinterpret_cast<const f4*>(Bb + 32 * GLD);
#pragma unroll
            for (int kb = 0; kb < GK / 8; ++kb) {
                const f4 a0 = fa0, a1 = fa1, b0 = fb0, b1 = fb1;
                if (kb + 1 < GK / 8) {
                    fa0 = *reinterpret_cast<const f4*>(Ab + (kb + 1) * 8);
                    fa1 = *reinterpret_cast<const f4*>(Ab + 32 * GLD + (kb + 1) * 8);
                    fb0 = *reinterpret_cast<const f4*>(Bb + (kb + 1) * 8);
                    fb1 = *reinterpret_cast<const f4*>(Bb + 32 * GLD + (kb + 1) * 8);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
                }
            }
        }
        if (DIAG & 1) {
            float keep = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
            if (keep == 123.456f) s_thr[0] = keep;      // keeps the accumulators live
            continue;
        }

        // ---- tile epilogue.  A lane owns two queries and 32 scores of each.  First the
        // cheap test "does any of my scores beat its query's threshold" (a max tree and one
        // ballot); only a wave with a hit walks its accumulators.
        const long long n0 = t * GT;
        float thr[2];
        bool hit = false;
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            const int ql = wq * 64 + nj * 32 + l31;
            thr[nj] = (q0 + ql) < a.nq ? s_thr[ql] : INFINITY;
            float m = -INFINITY;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[mi][nj][r]);
            hit |= m > thr[nj];
        }
        if (__ballot(hit) != 0ull) {
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                const int ql = wq * 64 + nj * 32 + l31;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float sc = acc[mi][nj][r];
                        if (sc > thr[nj])
                            push_candidate(a, cand, s_cnt, s_flag, ql, n0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, sc);
                    }
                }
            }
        }
    }
    // ---- final: every query's k best keys of this row range -> part ---------------------
    __syncthreads();
    {
        u64* scratch = reinterpret_cast<u64*>(As) + wave * CAPQ;
        for (int ql = wave; ql < GT; ql += 4) {
            if (q0 + ql >= a.nq) continue;
            const int c = s_cnt[ql];
            for (int i = lane; i < CAPQ; i += 64) scratch[i] = (i < c) ? cand[(size_t)ql * CAPQ + i] : KEY_EMPTY;
            bitonic_sort_desc_lds(scratch, CAPQ, lane, 64, SyncWave());
            u64* out = a.part + ((size_t)(q0 + ql) * a.S + sp) * k;
            for (int i = lane; i < k; i += 64) out[i] = scratch[i];
            SyncWave()();
        }
    }
}

bool gemm_topk_supported(const rag_index* h, long long nq, int k) {
    static const int min_nq = [] { const char* v = getenv("RAG_GEMM_MIN_NQ"); return (v && *v) ? atoi(v) : 16; }();
    return h->kind == RAG_KIND_FLAT && nq >= min_nq && k <= GEMM_MAX_K && (h->ld % GK) == 0 && h->n >= 1;
}

int gemm_topk_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask,
                     float* D, long long* I, hipStream_t s) {
    static const int target_wgs = [] { const char* v = getenv("RAG_GEMM_WGS"); return (v && *v) ? atoi(v) : 512; }();
    GemmArgs a;
    a.X = h->X;
    a.mask = mask;
    a.n = h->n;
    a.ntiles = (h->n + GT - 1) / GT;
    a.nq = (int)nq;
    a.ld = h->ld;
    a.k = k;
    a.nqb = (int)((nq + GT - 1) / GT);
    a.nqb_pad = a.nqb > 8 ? ((a.nqb + 7) & ~7) : a.nqb;
    long long S = target_wgs / a.nqb_pad;
    if (S < 1) S = 1;
    // a row range should hold at least 8 tiles, or splitting costs more than it spreads
    const long long max_s = (a.ntiles + 7) / 8;
    if (S > max_s) S = max_s < 1 ? 1 : max_s;
    a.S = (int)S;
    const int grid = a.nqb_pad * a.S;
    // workspace: [queries | candidate buffers | partial keys]; the queries are already at ws.ptr
    const size_t qbytes = ((size_t)nq * h->ld * sizeof(float) + 255) & ~(size_t)255;
    const size_t cbytes = (size_t)grid * GT * CAPQ * sizeof(u64);
    const size_t pbytes = (size_t)nq * a.S * k * sizeof(u64);
    if (h->ws.bytes < qbytes + cbytes + pbytes)
        return fail(RAG_ERR_INVALID, "internal: workspace not sized for the gemm path");
    a.Q = qbuf;
    a.cand = reinterpret_cast<u64*>(static_cast<char*>(h->ws.ptr) + qbytes);
    a.part = reinterpret_cast<u64*>(static_cast<char*>(h->ws.ptr) + qbytes + cbytes);
    const size_t lds = (size_t)(4 * GT * GLD) * sizeof(float) + GT * 8 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        RAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_topk_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    if (int rc = profile_begin(h, s)) return rc;
    hipLaunchKernelGGL(gemm_topk_kernel, dim3(grid), dim3(256), lds, s, a);
    RAG_HIP_CHECK(hipGetLastError());
    if (int rc = profile_end(h, s)) return rc;
    MergeArgs m;
    m.keys = a.part;
    m.per_query = (long long)a.S * k;
    m.k = k;
    m.cap = merge_cap(m.per_query, k);
    m.min_score = min_score;
    m.row_ids = h->ids_dev;
    m.id_base = h->id_base;
    m.slot_ids = nullptr;
    m.slot_stride = 0;
    m.D = D;
    m.I = I;
    RAG_HIP_CHECK(launch_merge_keys(m, nq, s));
    h->last_path = "gemm";
    return RAG_OK;
}

// bytes of workspace the gemm path needs after the padded queries
size_t gemm_topk_workspace(const rag_index* h, long long nq, int k) {
    static const int target_wgs = [] { const char* v = getenv("RAG_GEMM_WGS"); return (v && *v) ? atoi(v) : 512; }();
    const int nqb = (int)((nq + GT - 1) / GT);
    const int nqb_pad = nqb > 8 ? ((nqb + 7) & ~7) : nqb;
    long long S = target_wgs / nqb_pad;
    if (S < 1) S = 1;
    const long long ntiles = (h->n + GT - 1) / GT;
    const long long max_s = (ntiles + 7) / 8;
    if (S > max_s) S = max_s < 1 ? 1 : max_s;
    const size_t cbytes = (size_t)nqb_pad * S * GT * CAPQ * sizeof(u64);
    const size_t pbytes = (size_t)nq * S * k * sizeof(u64);
    return cbytes + pbytes;
}

}  // namespace rag
