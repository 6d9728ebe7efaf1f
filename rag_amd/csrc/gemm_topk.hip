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
// A lane reads k = 8*kb + 4*h + s (h = lane >> 5, s = 0..3) of its row as one
// ds_read_b128 and feeds element s to MFMA step s; both operands use the same k
// order, so the sum over (h, s) covers the 8 k's of the block exactly once.
//
// Staging, two variants (same results):
//   DMA = true   global_load_lds_dwordx4 writes the tiles straight into LDS (no
//                staging registers, no ds_write).  The LDS image is dense
//                [128 rows][8 x 16 B]; the 16-byte chunk c of row r sits at
//                position c ^ ((r >> 1) & 7), applied on the SOURCE address of
//                the DMA and again on the fragment read, which makes the reads
//                of any 16 rows that differ mod 16 hit 16 different bank groups.
//   DMA = false  registers + ds_write_b128 into rows padded to 36 floats.
//
// Selection.  Per query the workgroup keeps a threshold (LDS) and a candidate
// buffer of 256 keys (global memory, private to the workgroup, L2 resident).  A
// score enters the buffer only if it beats the threshold, which after the first
// few tiles is rare: the epilogue is a max tree + one ballot per wave, and only
// a wave with a hit walks its accumulators.  When a query's buffer could
// overflow on the next tile (count > 256 - 128) one wave sorts it in LDS, keeps k
// and tightens the threshold.  At the end every query's k best keys go to `part`,
// and merge_keys_kernel merges the row-range partials.
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
constexpr int GLD = GK + 4;      // padded LDS row stride (floats), register-staged variant
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

// one wave per query: sort the candidate buffer of every query that could overflow on the
// next tile (all == false) or of every query (all == true, final pass writing `out`)
__device__ __forceinline__ void compact_queries(u64* scratch, u64* cand, int* s_cnt, float* s_thr, int k, int wave, int lane) {
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
}

// Tile epilogue.  A lane owns two queries and 32 scores of each.  First the cheap test "does
// any of my scores beat its query's threshold" (a max tree and one ballot); only a wave
// with a hit walks its accumulators.
__device__ __forceinline__ void tile_epilogue(const GemmArgs& a, f16v (&acc)[2][2], u64* cand, float* s_thr, int* s_cnt,
                                              int* s_flag, int q0, long long n0, int wm, int wq, int l31, int h) {
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

template <bool DMA>
__global__ __launch_bounds__(256, 2) void gemm_topk_kernel(GemmArgs a) {
    constexpr int ROWF = DMA ? GK : GLD;                 // floats per LDS tile row
    constexpr int TILEF = GT * ROWF;                     // floats per operand tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* As = reinterpret_cast<float*>(smem);                       // [2][TILEF] corpus rows
    float* Bs = As + 2 * TILEF;                                        // [2][TILEF] queries
    float* s_thr = Bs + 2 * TILEF;                                     // [GT]
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

    const f4* X4 = reinterpret_cast<const f4*>(a.X);
    const f4* Q4 = reinterpret_cast<const f4*>(a.Q);
    const int ld4 = a.ld / 4;

    // ---- loader ------------------------------------------------------------------------
    // DMA: wave instruction i (0..3) fills tile rows (4*wave + i)*8 .. +7, lane -> (row = lane>>3,
    //      position = lane&7); the chunk fetched for a position is position ^ ((row>>1)&7).
    // registers: thread -> rows (tid>>3) + 32*i, chunk tid&7.
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    int trow[4], tchunk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (DMA) {
            trow[i] = (wv * 4 + i) * 8 + (lane >> 3);
            tchunk[i] = (lane & 7) ^ ((trow[i] >> 1) & 7);
        } else {
            trow[i] = (tid >> 3) + 32 * i;
            tchunk[i] = tid & 7;
        }
    }
    const f4* qp[4];
    const f4* xp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int qr = q0 + trow[i];
        qr = qr < a.nq ? qr : a.nq - 1;
        qp[i] = Q4 + (long long)qr * ld4 + tchunk[i];
    }
    auto set_tile = [&](long long t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long r = t * GT + trow[i];
            r = r < a.n ? r : a.n - 1;
            xp[i] = X4 + r * ld4 + tchunk[i];
        }
    };
    f4 xr[DMA ? 1 : 4], qr4[DMA ? 1 : 4];
    // request the tiles of K step kt (of the tile xp points at) for LDS buffer `buf`
    auto issue_load = [&](int kt, int buf) {
        if constexpr (DMA) {
            unsigned char* Ab = reinterpret_cast<unsigned char*>(As + buf * TILEF) + wv * 4096;
            unsigned char* Bb = reinterpret_cast<unsigned char*>(Bs + buf * TILEF) + wv * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xp[i] + kt * (GK / 4)),
                                                 (__attribute__((address_space(3))) void*)(Ab + i * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qp[i] + kt * (GK / 4)),
                                                 (__attribute__((address_space(3))) void*)(Bb + i * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xr[i] = xp[i][kt * (GK / 4)];
                qr4[i] = qp[i][kt * (GK / 4)];
            }
        }
    };
    auto store_lds = [&](int buf) {
        if constexpr (!DMA) {
            float* Ab = As + buf * TILEF + (tid >> 3) * GLD + (tid & 7) * 4;
            float* Bb = Bs + buf * TILEF + (tid >> 3) * GLD + (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f4*>(Ab + i * 32 * GLD) = xr[i];
                *reinterpret_cast<f4*>(Bb + i * 32 * GLD) = qr4[i];
            }
        }
    };
    // fragment read offsets (floats) of this lane inside an operand tile, per 8-wide k block
    int foff[GK / 8];
#pragma unroll
    for (int kb = 0; kb < GK / 8; ++kb) {
        if constexpr (DMA) foff[kb] = l31 * GK + (((kb * 2 + h) ^ ((l31 >> 1) & 7)) << 2);
        else foff[kb] = l31 * GLD + kb * 8 + h * 4;
    }

    f16v acc[2][2];
    int buf = 0;
    if (t_begin < t_end) {
        set_tile(t_begin);
        issue_load(0, 0);
    }
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
                __syncthreads();     // stage `buf` is complete (an LDS-DMA in flight is drained here too)
            }
            // a candidate buffer filled up during the previous tile's epilogue: every wave has
            // passed the barrier above, so the flag is stable and stage buf^1 is idle
            if (kt == 0 && *s_flag != 0) {
                compact_queries(reinterpret_cast<u64*>(As + (buf ^ 1) * TILEF) + wave * CAPQ, cand, s_cnt, s_thr, k, wave, lane);
                __syncthreads();
                if (tid == 0) *s_flag = 0;
            }
            if (!(DIAG & 2) || (t == t_begin && kt == 0)) {
                if (kt + 1 < KT) issue_load(kt + 1, buf ^ 1);
                else if (t + 1 < t_end) { set_tile(t + 1); issue_load(0, buf ^ 1); }
            }
            const float* Ab = As + buf * TILEF + (wm * 64) * ROWF;
            const float* Bb = Bs + buf * TILEF + (wq * 64) * ROWF;
            // fragments of k block kb+1 are requested while the MFMAs of block kb issue: one
            // ds_read_b128 in front of every four MFMAs (two register sets, kb parity)
            f4 fa0[2], fa1[2], fb0[2], fb1[2];
            fa0[0] = *reinterpret_cast<const f4*>(Ab + foff[0]);
            fa1[0] = *reinterpret_cast<const f4*>(Ab + 32 * ROWF + foff[0]);
            fb0[0] = *reinterpret_cast<const f4*>(Bb + foff[0]);
            fb1[0] = *reinterpret_cast<const f4*>(Bb + 32 * ROWF + foff[0]);
#pragma unroll
            for (int kb = 0; kb < GK / 8; ++kb) {
                const int cur = kb & 1, nxt = cur ^ 1;
                if (kb + 1 < GK / 8) {
                    fa0[nxt] = *reinterpret_cast<const f4*>(Ab + foff[kb + 1]);
                    fa1[nxt] = *reinterpret_cast<const f4*>(Ab + 32 * ROWF + foff[kb + 1]);
                    fb0[nxt] = *reinterpret_cast<const f4*>(Bb + foff[kb + 1]);
                    fb1[nxt] = *reinterpret_cast<const f4*>(Bb + 32 * ROWF + foff[kb + 1]);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cur][s], fb0[cur][s], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cur][s], fb1[cur][s], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cur][s], fb0[cur][s], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cur][s], fb1[cur][s], acc[1][1], 0, 0, 0);
                }
                if (kb + 1 < GK / 8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
                    }
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
        tile_epilogue(a, acc, cand, s_thr, s_cnt, s_flag, q0, t * GT, wm, wq, l31, h);
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

static int env_int_g(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

bool gemm_topk_supported(const rag_index* h, long long nq, int k) {
    static const int min_nq = env_int_g("RAG_GEMM_MIN_NQ", 16);
    return h->kind == RAG_KIND_FLAT && nq >= min_nq && k <= GEMM_MAX_K && (h->ld % GK) == 0 && h->n >= 1;
}

// (query blocks, padded query blocks, row ranges) for a batch of nq queries
static void gemm_plan(const rag_index* h, long long nq, int* nqb, int* nqb_pad, int* S) {
    static const int target_wgs = env_int_g("RAG_GEMM_WGS", 512);
    *nqb = (int)((nq + GT - 1) / GT);
    *nqb_pad = *nqb > 8 ? ((*nqb + 7) & ~7) : *nqb;
    long long s = target_wgs / *nqb_pad;
    if (s < 1) s = 1;
    // a row range should hold at least 8 tiles, or splitting costs more than it spreads
    const long long ntiles = (h->n + GT - 1) / GT;
    const long long max_s = (ntiles + 7) / 8;
    if (s > max_s) s = max_s < 1 ? 1 : max_s;
    *S = (int)s;
}

// bytes of workspace the gemm path needs after the padded queries
size_t gemm_topk_workspace(const rag_index* h, long long nq, int k) {
    int nqb, nqb_pad, S;
    gemm_plan(h, nq, &nqb, &nqb_pad, &S);
    return (size_t)nqb_pad * S * GT * CAPQ * sizeof(u64) + (size_t)nq * S * k * sizeof(u64);
}

int gemm_topk_search(rag_index* h, long long nq, const float* qbuf, int k, float min_score, const uint8_t* mask,
                     float* D, long long* I, hipStream_t s) {
    static const int use_dma = env_int_g("RAG_GEMM_DMA", 1);
    GemmArgs a;
    a.X = h->X;
    a.mask = mask;
    a.n = h->n;
    a.ntiles = (h->n + GT - 1) / GT;
    a.nq = (int)nq;
    a.ld = h->ld;
    a.k = k;
    gemm_plan(h, nq, &a.nqb, &a.nqb_pad, &a.S);
    const int grid = a.nqb_pad * a.S;
    // workspace: [queries | candidate buffers | partial keys]
    const size_t qbytes = ((size_t)nq * h->ld * sizeof(float) + 255) & ~(size_t)255;
    const size_t cbytes = (size_t)grid * GT * CAPQ * sizeof(u64);
    const size_t pbytes = (size_t)nq * a.S * k * sizeof(u64);
    if (h->ws.bytes < qbytes + cbytes + pbytes)
        return fail(RAG_ERR_INVALID, "internal: workspace not sized for the gemm path");
    a.Q = qbuf;
    a.cand = reinterpret_cast<u64*>(static_cast<char*>(h->ws.ptr) + qbytes);
    a.part = reinterpret_cast<u64*>(static_cast<char*>(h->ws.ptr) + qbytes + cbytes);
    const size_t lds = (size_t)(4 * GT * (use_dma ? GK : GLD)) * sizeof(float) + GT * 8 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        const size_t lds_max = (size_t)(4 * GT * GLD) * sizeof(float) + GT * 8 + 16;
        RAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_topk_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        RAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_topk_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        attr_set = true;
    }
    if (int rc = profile_begin(h, s)) return rc;
    if (use_dma) hipLaunchKernelGGL(gemm_topk_kernel<true>, dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(gemm_topk_kernel<false>, dim3(grid), dim3(256), lds, s, a);
    RAG_HIP_CHECK(hipGetLastError());
    if (int rc = profile_end(h, s)) return rc;
    MergeArgs m = {};
    m.keys = a.part;
    m.per_query = (long long)a.S * k;
    m.k = k;
    m.cap = merge_cap(m.per_query, k);
    m.min_score = min_score;
    m.row_ids = h->ids_dev;
    m.id_base = h->id_base;
    m.D = D;
    m.I = I;
    RAG_HIP_CHECK(launch_merge_keys(m, nq, s));
    h->last_path = "gemm";
    return RAG_OK;
}

}  // namespace rag
