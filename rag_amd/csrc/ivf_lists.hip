// List-major IVF scan, second form (gfx950): the (query, probe) pairs of a batch grouped by list, a work item =
// "up to 32 pairs of one list", persistent workgroups that take work items from a device counter.
//
// Roofline: HBM.  Every probed list is read once per 32 probing queries: algorithmic bytes = rows of the probed
// lists * ld * 4.  The arithmetic runs on v_mfma_f32_16x16x4_f32 in 16-query column blocks, and a work item with
// 16 pairs or fewer multiplies ONE block: at nq * nprobe / nlist = 16 pairs per list on average more than half of
// the work items do, which halves their matrix-core time (with 32 columns always multiplied that time is about
// what the bytes need at the HBM peak, so both roofs bound the 256 x 32 LDS-tiled shape this replaces).
//
// Structure (the register-streaming scheme of narrow_topk.hip, with the queries gathered per work item).  The four
// waves of a workgroup split a list's rows into four contiguous quarters.  A wave walks its rows in tiles of 32; per
// K step (32 floats) it fetches 32 rows x 128 contiguous bytes (four wave instructions of 8 rows x 128 B,
// non-temporal) into a ring of DEPTH register sets, so DEPTH - 1 steps stay in flight per wave while one is
// multiplied; a landed step goes through the wave-private 4 KiB LDS transposer (ds_write_b128 rows, ds_read_b128
// fragments, 16-byte chunks XOR-swizzled by row pair: conflict free both ways).  The queries of the work item are
// shared: per rotation of four K steps each wave fetches ONE step's query image (up to 32 gathered queries x 128 B,
// coalesced, from L2), one rotation ahead, writes it to LDS in the transposer's layout, and after one LDS-only
// barrier per rotation (the row loads stay in flight across it) every wave reads its B fragments from the four
// images -- a quarter of the query traffic of four independent waves, and no query registers in the ring.
//
// Selection (k <= 32), per wave and pair: a 64-key candidate buffer in global memory (L2) and a threshold in LDS; a
// query's lists share one threshold word (atomicMax on ordered-float bits), so a pair starts from the k-th best
// score any other list of its query has reached.  A buffer past 32 keys is sorted in registers (one key per lane),
// cut to k and the threshold tightened.  When a work item's rows are done the four waves' lists of a pair are
// merged (two barriers per work item) and the pair's k keys go to its slot; merge_keys_*_kernel reduces a query's
// nprobe slots as before.  Candidate keys carry the storage position (inside a list the same order as the insertion
// position); the kept keys of a pair are renamed to insertion positions (row_alias) when they are written; masks are
// read where a row is tested.
//
// Scores are a k-ordered fp32 fma chain per (row, query), independent of the launch geometry and of which
// workgroup takes which work item.
#include <float.h>

#include <algorithm>

#include "index.h"
#include "ivf_lists.h"
#include "topk_keys.h"

namespace rag {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int LS_WAVES = 4;
constexpr int LS_TILE = 32;            // rows per wave step (two 16-row MFMA row groups)
constexpr int LS_TRIG = 32;            // compact a candidate buffer once it holds more keys than this
constexpr int LS_CAP = LS_TRIG + LS_TILE;   // 64: one key per lane when sorted

// timing-only ablations for diagnostic builds (outputs are wrong when non-zero): bit 0 skips the tile epilogue,
// bit 1 re-reads the first K step's query fragments (L1-resident), bit 2 re-reads the wave's first tile (cache-resident
// rows), bit 3 multiplies two column blocks for every work item, bit 4 skips the merge of a work item's lists,
// bit 5 one fma in place of each MFMA, bit 6 no transposer, bit 7 no query loads, bit 8 no row loads
#ifndef RAG_LISTS_DIAG
#define RAG_LISTS_DIAG 0
#endif
// K steps in the register ring for work items of one / two column blocks (measurement knobs)
#ifndef RAG_LISTS_DEPTH1
#define RAG_LISTS_DEPTH1 8
#endif
#ifndef RAG_LISTS_DEPTH2
#define RAG_LISTS_DEPTH2 4
#endif

// issue order of a K step (measurement knob): 1 deals the next step's LDS operations between this step's MFMAs
#ifndef RAG_LISTS_SGB
#define RAG_LISTS_SGB 1
#endif
constexpr int LS_ROT = 4;              // K steps per rotation of the query images (one step's image per wave)

// one K step of a wave's rows as loaded: [row group][8-row half], 16 bytes per lane each
struct LsStep {
    f4 a[2][2];
};

__device__ __forceinline__ long long row_lo_key(u64 key) { return (long long)key_row(key); }

// workgroup barrier that orders LDS accesses only: the register ring's loads stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The rows [r_lo, r_hi) of one list against the work item's queries, by one wave, in step with the other waves
// of its workgroup: every wave walks T tiles (a wave with fewer rows re-reads its last ones and selects nothing).
// tp: the wave's transposer; bq: the workgroup's query images [2][LS_ROT][256 f4]; my_thr / my_cnt / my_cand: the
// wave's thresholds, counts and candidate buffers for the item's (up to 32) pairs.
template <int NCB, int DEPTH, bool MASKED>
__device__ __forceinline__ void ls_scan_rows(const ListScanArgs& a, f4* tp, f4* bq, float* my_thr, int* my_cnt, u64* my_cand,
                                             const int* pq, int m_valid, long long r_lo, long long r_hi, long long T, bool live,
                                             int wave, int lane) {
    static_assert(DEPTH % LS_ROT == 0, "the row ring turns in whole rotations of the query images");
    const int r16 = lane & 15, kq = lane >> 4;
    const int lrow = lane >> 3, lchunk = lane & 7;      // as loaded: row within an 8-row half, 16-byte chunk
    const int k = a.k;
    const int ld4 = a.ld / 4;
    const int steps = a.ld / 32;                        // K steps per tile, a multiple of DEPTH
    const f4* X4 = reinterpret_cast<const f4*>(a.X);
    const f4* Q4 = reinterpret_cast<const f4*>(a.Q);
    int wr_off[2], rd_off[2];                           // f4 units inside one 16-row image
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int ru = hh * 8 + lrow;
        wr_off[hh] = ru * 8 + (lchunk ^ ((ru >> 1) & 7));
        rd_off[hh] = r16 * 8 + ((4 * hh + kq) ^ ((r16 >> 1) & 7));      // hh = which 16 floats of the step (m)
    }
    // ---- queries: one K step of all columns is an image like the transposer's (16 columns per group, same swizzle);
    // wave w fetches the image of step 4 r + w of every rotation r -- coalesced, 8 queries x 128 contiguous bytes
    // per wave instruction -- one rotation ahead, and every wave reads its B fragments from the images.  Surplus
    // columns repeat the last pair's query and are never selected.
    const f4* bp[2 * NCB];
#pragma unroll
    for (int i = 0; i < 2 * NCB; ++i) {
        const int ql = 8 * i + lrow;
        bp[i] = Q4 + (size_t)pq[ql < m_valid ? ql : m_valid - 1] * ld4 + lchunk;
    }
    int qg[NCB];                                        // the query of this lane's column (for the shared threshold)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int ql = cb * 16 + r16;
        qg[cb] = pq[ql < m_valid ? ql : m_valid - 1];
    }
    int bs = wave;                                      // K step (within a tile) of the image this wave fetches next
    f4 share[2 * NCB];
    auto fetch_b = [&]() {
#pragma unroll
        for (int i = 0; i < 2 * NCB; ++i) share[i] = (RAG_LISTS_DIAG & 128) ? (f4){1.f, 2.f, 3.f, 4.f} : bp[i][bs * 8];
        bs += LS_ROT;
        if (bs >= steps) bs -= steps;
    };
    auto write_b = [&](int buf) {
        f4* dst = bq + (buf * LS_ROT + wave) * 256;
#pragma unroll
        for (int i = 0; i < 2 * NCB; ++i) dst[(i >> 1) * 128 + wr_off[i & 1]] = share[i];
    };
    f4 bf[2][2][NCB];                                   // [parity][m][column block]
    auto read_b = [&](int buf, int step, int par) {
        const f4* src = bq + (buf * LS_ROT + step) * 256;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) bf[par][m][cb] = src[cb * 128 + rd_off[m]];
    };
    // ---- rows: rows past the end repeat the last row; tiles past the range repeat the last tile: loaded, never used
    auto row_ptr = [&](long long t, int rg, int hh) {
        t = t < T ? t : T - 1;
        long long row = r_lo + t * LS_TILE + rg * 16 + hh * 8 + lrow;
        row = row < r_hi ? row : r_hi - 1;
        return X4 + row * ld4 + lchunk;
    };
    LsStep ring[DEPTH];
    long long lt = 0;
    int ls = 0;
    const f4* lp[2][2];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) lp[rg][hh] = row_ptr(0, rg, hh);
    auto issue = [&](LsStep& s) {
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if (RAG_LISTS_DIAG & 256) s.a[rg][hh] = (f4){(float)ls, 1.f, 2.f, 3.f};
                else s.a[rg][hh] = __builtin_nontemporal_load(lp[rg][hh] + ls * 8);
            }
        if (++ls == steps) {
            ls = 0;
            if (!(RAG_LISTS_DIAG & 4)) ++lt;
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) lp[rg][hh] = row_ptr(lt, rg, hh);
        }
    };
    fetch_b();                                          // the first rotation's image of this wave: the oldest load
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) issue(ring[s]);
    // The other lists' progress on this lane's queries, read ONCE per work item, behind the ring's first loads and
    // in front of the first step: a value loaded inside the loop would be waited for with every younger load of the
    // ring (loads return in order), i.e. it would drain the ring.  The wave's own thresholds tighten inside the
    // work item; the shared word is raised again when the work item's lists are merged.
    float tg[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const unsigned int bits = __hip_atomic_load(&a.gthr[qg[cb]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tg[cb] = bits != 0u ? ordered_f32(bits) : -INFINITY;
    }
    f4 af[2][2][2];                                     // [parity][row group][m]
    auto transpose = [&](const LsStep& c, int par) {
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) tp[rg * 128 + wr_off[hh]] = c.a[rg][hh];
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int m = 0; m < 2; ++m) af[par][rg][m] = tp[rg * 128 + rd_off[m]];
    };
    write_b(0);
    lds_barrier();
    transpose(ring[0], 0);
    read_b(0, 0, 0);

    // sort every buffer holding more than min_count keys, keep its k best, tighten the threshold
    auto compact = [&](int min_count) {
        // the pushed keys are re-read through memory by other lanes of this wave: drain the stores first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int q = 0; q < m_valid; ++q) {
            const int c = my_cnt[q];                     // wave-uniform
            if (c <= min_count) continue;
            u64 key = lane < c ? my_cand[(size_t)q * LS_CAP + lane] : KEY_EMPTY;
            key = wave_sort_desc(key, lane);             // empty slots (masked rows, rows past the end) sort last
            const int keep = c < k ? c : k;
            if (lane < keep) my_cand[(size_t)q * LS_CAP + lane] = key;
            const u64 kth = readlane_u64(key, k - 1);
            if (c >= k && kth != KEY_EMPTY) {
                const float t_new = key_score(kth);
                if (lane == 0) {
                    my_thr[q] = t_new;
                    atomicMax(&a.gthr[pq[q]], f32_ordered(t_new));
                }
            }
            if (lane == 0) my_cnt[q] = keep;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    constexpr bool has_mask = MASKED;          // a kernel of its own: the unmasked epilogue must not contain a load at all
    // tile epilogue: lane (c = r16, kq) holds rows 4 kq .. 4 kq + 3 of each row group for column cb * 16 + c
    auto epilogue = [&](long long t, const f4 (&acc)[2][NCB]) {
        if (!live) return;
        if (RAG_LISTS_DIAG & 1) {
            float keep = tg[0];
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) keep += acc[rg][cb][0] + acc[rg][cb][1] + acc[rg][cb][2] + acc[rg][cb][3];
            if (keep == 123.456f) my_thr[0] = keep;      // keeps the accumulators live
            return;
        }
        const long long row_base = r_lo + t * LS_TILE + kq * 4;
        float thr[NCB];
        bool any = false;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int q = cb * 16 + r16;
            thr[cb] = q < m_valid ? my_thr[q] : INFINITY;
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) {
                const float m = fmaxf(fmaxf(acc[rg][cb][0], acc[rg][cb][1]), fmaxf(acc[rg][cb][2], acc[rg][cb][3]));
                any |= m > thr[cb] && m >= tg[cb];       // a tie with another list's k-th best may still win on row number
            }
        }
        if (__ballot(any) == 0ull) return;
        bool over = false;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int q = cb * 16 + r16;
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float sc = acc[rg][cb][reg];
                    const long long row = row_base + rg * 16 + reg;
                    if (sc > thr[cb] && sc >= tg[cb] && row < r_hi) {
                        // Candidate keys carry the STORAGE position: inside a list storage order is insertion order, so ties
                        // fall the same way, and the position a caller sees is looked up once per kept key when the work
                        // item's lists are merged.  Without masks a push therefore loads nothing (a load here would be
                        // waited for together with every load of the ring).
                        bool ok = true;
                        if constexpr (has_mask) {
                            const long long shown = a.row_alias ? a.row_alias[row] : row;     // masks count rows in insertion order
                            ok = (!a.mask || a.mask[shown]) && (!a.mask2 || a.mask2[shown]);
                        }
                        if (ok) {
                            const int pos = atomicAdd(&my_cnt[q], 1);
                            my_cand[(size_t)q * LS_CAP + pos] = make_key(sc, (uint32_t)row);
                            over |= pos + 1 > LS_TRIG;
                        }
                    }
                }
            }
        }
        if (__ballot(over) != 0ull) {
            SyncWave()();                                // the lanes' LDS updates, in order, before the counts are read
            compact(LS_TRIG);
            // this (rare) path ends with nothing outstanding: where it joins the common path the compiler otherwise
            // waits for every load of the ring (it merges the two paths' outstanding-load state conservatively)
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        } else if constexpr (has_mask) {
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
    };

    // ONE loop over every ring rotation of every tile (the epilogue sits inside it behind a wave-uniform branch): a
    // tile loop around a K loop makes the compiler drain every outstanding load in front of each K loop
    f4 acc[2][NCB];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rg][cb] = (f4){0.f, 0.f, 0.f, 0.f};
    const long long turns = T * (steps / DEPTH);
    long long t = 0;
    int g = 0, buf = 0;
    for (long long it = 0; it < turns; ++it) {
#pragma unroll
        for (int rr = 0; rr < DEPTH / LS_ROT; ++rr) {
            fetch_b();                                   // this wave's image of the next rotation
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < LS_ROT; ++u) {
                const int slot = rr * LS_ROT + u, par = u & 1;
                if (RAG_LISTS_SGB) {
                    // The next step (of this tile, or step 0 of the next one: the ring holds it already) goes through the
                    // transposer WHILE this step multiplies: its LDS operations are dealt between the MFMAs, one per PER of
                    // them, so neither their issue nor their latency stands between two steps' MFMAs.
                    constexpr int OPS_B = 8 + 2 * NCB, PER = 16 * NCB / OPS_B;
                    transpose(ring[(slot + 1) % DEPTH], par ^ 1);
                    if (u + 1 < LS_ROT) read_b(buf, u + 1, par ^ 1);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                                for (int cb = 0; cb < NCB; ++cb)
                                    acc[rg][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[par][rg][m][v], bf[par][m][cb][v], acc[rg][cb], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                    }
                    if (RAG_LISTS_SGB == 2) {
                        // the row loads of the step DEPTH ahead as well, behind the LDS operations
                        issue(ring[slot]);
                        if (u + 1 < LS_ROT) {
#pragma unroll
                            for (int i = 0; i < 2 * NCB; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        }
                        if (u + 1 < LS_ROT) __builtin_amdgcn_sched_group_barrier(0x008, 16 * NCB - PER * OPS_B - 4, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x008, 16 * NCB - PER * 8 - 4, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        continue;
                    }
                    if (u + 1 < LS_ROT) {
#pragma unroll
                        for (int i = 0; i < 2 * NCB; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 16 * NCB - PER * OPS_B, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 16 * NCB - PER * 8, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    issue(ring[slot]);
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                            for (int cb = 0; cb < NCB; ++cb)
                                acc[rg][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[par][rg][m][v], bf[par][m][cb][v], acc[rg][cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // BEHIND the step's MFMAs: the next step (of this tile, or step 0 of the next one: the ring holds it
                // already) goes through the transposer -- its rows have had DEPTH - 1 MFMA phases to arrive -- and the set
                // that was just multiplied takes the step DEPTH ahead, so DEPTH - 1 steps are in flight while the MFMAs run
                transpose(ring[(slot + 1) % DEPTH], par ^ 1);
                if (u + 1 < LS_ROT) read_b(buf, u + 1, par ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                issue(ring[slot]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the rotation ends: this wave's image of the next one goes to the other buffer (nobody reads that one:
            // its last readers passed the previous barrier), and behind the barrier every wave's image is there
            write_b(buf ^ 1);
            lds_barrier();
            buf ^= 1;
            read_b(buf, 0, 0);
        }
        g += DEPTH;
        if (g == steps) {
            epilogue(t, acc);
            g = 0;
            ++t;
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[rg][cb] = (f4){0.f, 0.f, 0.f, 0.f};
        }
    }
}

template <int DEPTH1, int DEPTH2, bool MASKED>
__global__ __launch_bounds__(LS_WAVES * 64, RAG_LISTS_WGS_PER_CU) void ivf_lists_kernel(ListScanArgs a) {
    __shared__ __attribute__((aligned(16))) f4 s_tp[LS_WAVES * 256];          // transposers, 4 KiB per wave
    __shared__ __attribute__((aligned(16))) f4 s_bq[2 * LS_ROT * 256];        // query images: two rotations of four K steps, 4 KiB each
    __shared__ float s_thr[LS_WAVES * LS_QT];
    __shared__ int s_cnt[LS_WAVES * LS_QT];
    __shared__ int s_task;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = a.k;
    const int ntasks = a.counters[0];
    float* my_thr = s_thr + wave * LS_QT;
    int* my_cnt = s_cnt + wave * LS_QT;
    u64* wg_cand = a.cand + (size_t)blockIdx.x * LS_WAVES * LS_QT * LS_CAP;
    u64* my_cand = wg_cand + (size_t)wave * LS_QT * LS_CAP;
    for (;;) {
        if (tid == 0) s_task = atomicAdd(&a.counters[1], 1);
        __syncthreads();
        const int task = s_task;
        if (task >= ntasks) break;                      // every wave of every workgroup gets here: the counter only grows
        const int lst = a.task_list[task];
        const int g = task - a.task_begin[lst];
        const int p0 = a.pair_begin[lst] + g * LS_QT;
        int m_valid = a.pair_cnt[lst] - g * LS_QT;
        m_valid = m_valid < LS_QT ? m_valid : LS_QT;
        const long long row_lo = a.list_begin[lst], row_hi = a.list_begin[lst + 1];
        // this wave's quarter of the list, in whole 16-row groups
        const long long per = (((row_hi - row_lo + LS_WAVES - 1) / LS_WAVES) + 15) & ~15LL;
        const long long r_lo = std::min(row_hi, row_lo + wave * per), r_hi = std::min(row_hi, r_lo + per);
        if (lane < LS_QT) {
            my_thr[lane] = -INFINITY;
            my_cnt[lane] = 0;
        }
        SyncWave()();
        {
            // every wave walks the same number of tiles (they meet at a barrier per rotation of the query images); a
            // wave without rows walks the list's last row and selects nothing
            const long long t_all = (per + LS_TILE - 1) / LS_TILE;
            const bool live = r_lo < r_hi;
            const long long lo = live ? r_lo : row_hi - 1, hi = live ? r_hi : row_hi;
            f4* tp = s_tp + wave * 256;
            const int* pq = a.pair_q + p0;
            const bool deep = (a.ld % (32 * DEPTH1)) == 0;       // K steps per row fill whole turns of the deeper ring
            if (m_valid <= 16 && !(RAG_LISTS_DIAG & 8)) {
                if (deep) ls_scan_rows<1, DEPTH1, MASKED>(a, tp, s_bq, my_thr, my_cnt, my_cand, pq, m_valid, lo, hi, t_all, live, wave, lane);
                else ls_scan_rows<1, LS_ROT, MASKED>(a, tp, s_bq, my_thr, my_cnt, my_cand, pq, m_valid, lo, hi, t_all, live, wave, lane);
            } else {
                ls_scan_rows<2, DEPTH2, MASKED>(a, tp, s_bq, my_thr, my_cnt, my_cand, pq, m_valid, lo, hi, t_all, live, wave, lane);
            }
        }
        // ---- the four waves' lists of every pair -> the pair's k keys ----------------------------------
        __syncthreads();                                 // (drains every wave's stores to its candidate buffers)
        for (int q = (RAG_LISTS_DIAG & 16) ? m_valid : wave; q < m_valid; q += LS_WAVES) {
            int c[LS_WAVES], total = 0;
#pragma unroll
            for (int w = 0; w < LS_WAVES; ++w) {
                c[w] = s_cnt[w * LS_QT + q];             // wave-uniform
                total += c[w];
            }
            // once a query's lists share a threshold most pairs keep nothing: their slot stays empty (it was cleared)
            if (total == 0) continue;
            u64 best;                                    // lane j < k: the j-th best of the pair
            if (total <= 64) {
                // the common case: every kept key of the four waves fits one key per lane -> one sort
                int w_of = 0, off = lane;
#pragma unroll
                for (int w = 0; w < LS_WAVES - 1; ++w)
                    if (w_of == w && off >= c[w]) { off -= c[w]; w_of = w + 1; }
                best = lane < total ? wg_cand[((size_t)w_of * LS_QT + q) * LS_CAP + off] : KEY_EMPTY;
                best = wave_sort_desc(best, lane);
            } else {
                // lists are unsorted runs of at most 64 keys; fold them in one at a time, keeping the k <= 32 best
                best = KEY_EMPTY;
                for (int w = 0; w < LS_WAVES; ++w) {
                    const u64* src = wg_cand + ((size_t)w * LS_QT + q) * LS_CAP;
                    u64 key = lane < c[w] ? src[lane] : KEY_EMPTY;
                    key = wave_sort_desc(key, lane);
                    // the k best of (best, key): both descending, k <= 32 -> the other list's head goes to lanes 32..
                    const u64 moved = shfl_u64(key, (lane - 32) & 63);
                    const u64 both = lane < 32 ? (lane < k ? best : KEY_EMPTY) : (lane - 32 < k ? moved : KEY_EMPTY);
                    best = wave_sort_desc(both, lane);
                }
            }
            // storage position -> the position callers see (equal scores keep their order: within a list both orders agree)
            if (a.row_alias && best != KEY_EMPTY) best = (best & 0xFFFFFFFF00000000ull) | (u64)(0xFFFFFFFFu - (uint32_t)a.row_alias[row_lo_key(best)]);
            if (lane < k) a.part[(size_t)a.pair_slot[p0 + q] * k + lane] = best;
            const u64 kth = readlane_u64(best, k - 1);
            if (kth != KEY_EMPTY && lane == 0) atomicMax(&a.gthr[a.pair_q[p0 + q]], f32_ordered(key_score(kth)));
        }
        __syncthreads();                                 // the next work item re-initialises counts and thresholds
    }
}

static int env_int_l(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

bool ivf_lists_kernel_ok(const rag_index* h, int k) {
    static const int on = env_int_l("RAG_IVF_LISTS_KERNEL", 1);          // 0: the LDS-tiled 256 x 32 shape (measurement knob)
    // K steps per row must fill whole ring rotations of either depth
    return on && k <= LS_MAX_K && (h->ld % 128) == 0;
}

size_t ivf_lists_cand_bytes() { return (size_t)LS_GRID * LS_WAVES * LS_QT * LS_CAP * sizeof(u64); }

int ivf_lists_launch(const ListScanArgs& a, hipStream_t s) {
    static const int grid = std::max(1, env_int_l("RAG_IVF_LISTS_GRID", LS_GRID));      // measurement knob (<= LS_GRID: the candidate buffers)
    const dim3 g(std::min(grid, LS_GRID)), b(LS_WAVES * 64);
    if (a.mask || a.mask2) hipLaunchKernelGGL((ivf_lists_kernel<RAG_LISTS_DEPTH1, RAG_LISTS_DEPTH2, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((ivf_lists_kernel<RAG_LISTS_DEPTH1, RAG_LISTS_DEPTH2, false>), g, b, 0, s, a);
    RAG_HIP_CHECK(hipGetLastError());
    return RAG_OK;
}

}  // namespace rag
