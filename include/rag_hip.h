/*
 * rag_hip.h -- C ABI of librag_hip.so, the MI355X (gfx950) implementation of the
 * dense-retrieval hot path of llano1025/rAG.
 *
 * Every entry point takes plain pointers and sizes, returns an int status
 * (RAG_OK = 0) and leaves a message for rag_last_error() on failure.  Each
 * declaration names the reference interface it stands in for (paths relative to
 * the reference tree).  Host pointers unless a parameter is named *_dev.
 *
 * Threading: every entry point that takes an index holds that index's mutex for
 * the host side of the call, so host threads may share an index (their calls are
 * serialised); calls on different indexes are independent; the fusion and MMR
 * entry points keep per-thread workspaces.  rag_index_free must not overlap
 * another call on the same index.
 */
#ifndef RAG_HIP_H
#define RAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAG_ABI_VERSION 1

enum rag_status {
    RAG_OK = 0,
    RAG_ERR_INVALID = 1,      /* bad argument (message says which) */
    RAG_ERR_HIP = 2,          /* a HIP runtime call failed */
    RAG_ERR_OOM = 3,          /* device allocation failed */
    RAG_ERR_UNSUPPORTED = 4   /* valid request this build does not cover */
};

/* vector_db/search_optimizer.py:33-36 MetricType (mapped to a distance at :100-109).  COSINE rows
 * are normalised when added and queries when searched (what a Qdrant COSINE collection does),
 * after which the scan is an inner product.
 * L2 (MetricType.EUCLIDEAN; FAISS IndexFlatL2 semantics; FLAT indexes only): a search returns
 * SQUARED Euclidean distances, nearest first (equal distances: lower row first), empty slots as
 * I = -1 / D = FLT_MAX, and its min_score argument is the LARGEST squared distance returned
 * (negative: no limit) -- Qdrant reads score_threshold as an upper bound for its Euclid distance
 * (call site :213-222).  The rows are ranked on the same scan / matrix-core kernels by
 * q.x - |x|^2 / 2 (a row stores -|x|^2 / 2 in an extra column, its stride is a multiple of 32
 * floats); the k kept rows are then re-scored as sums of squared fp32 differences, so a row equal
 * to the query is at distance exactly 0.  rag_multi_* merges the devices' answers nearest first.
 * Segment search and the fusion + MMR call rank by similarity and return RAG_ERR_UNSUPPORTED, as do
 * adopt and the bf16 first pass (an L2 row is wider than the caller's). */
enum rag_metric { RAG_METRIC_IP = 0, RAG_METRIC_COSINE = 1, RAG_METRIC_L2 = 2 };

/* FLAT = exhaustive scan (FAISS IndexFlatIP semantics); IVF = coarse lists
 * (FAISS IndexIVFFlat semantics with an inner-product flat quantiser). */
enum rag_kind { RAG_KIND_FLAT = 0, RAG_KIND_IVF = 1 };

/* similarity used by MMR, vector_db/mmr_diversifier.py:257-268 */
enum rag_mmr_metric { RAG_MMR_COSINE = 0, RAG_MMR_DOT = 1, RAG_MMR_EUCLIDEAN = 2 };

typedef struct rag_index rag_index_t;

/* ---- library ------------------------------------------------------------ */
int rag_version(void);
/* thread-local message of the last failing call on this thread */
const char* rag_last_error(void);
int rag_device_count(int* count);
/* Host-pointer searches, rag_merge_topk_device and the fusion / MMR entry points stage their data in
 * device memory that belongs to the calling host thread (grow-only, one block per purpose and device).
 * A thread that is about to end -- a worker of a thread pool that is being resized -- hands it back
 * with this call; it must not race with another call of the same thread (it cannot) and is optional. */
int rag_thread_release(void);

/* ---- index lifetime ------------------------------------------------------
 * One rag_index_t replaces one Qdrant collection + its QdrantSearchOptimizer
 * (vector_db/search_optimizer.py:68-86, created at
 * vector_db/storage_manager.py:190-221).  Rows live in HBM on `device`. */
int rag_index_create(int dim, int metric, int kind, int device, rag_index_t** out);
/* vector_db/storage_manager.py:718-750 delete_index / :985-993 cleanup */
int rag_index_free(rag_index_t* h);
/* capacity hint so that a known-size corpus is allocated once */
int rag_index_reserve(rag_index_t* h, int64_t nrows);
/* vector_db/storage_manager.py:254-366 add_vectors -> qdrant upsert
 * (vector_db/qdrant_client.py:207-246).  x is row-major n x dim fp32.  ids == NULL
 * gives each row the id "id_base + row position" (FAISS insertion order). */
int rag_index_add(rag_index_t* h, int64_t n, const float* x, const int64_t* ids);
/* same, rows already in device memory of the index's device (copied) */
int rag_index_add_device(rag_index_t* h, int64_t n, const float* x_dev, const int64_t* ids);
/* zero-copy: search directly in the caller's device buffer (row-major n x dim,
 * dim % 4 == 0, 16-byte aligned, kept alive by the caller; COSINE rows are
 * normalised in place).  Only on an empty index. */
int rag_index_adopt_device(rag_index_t* h, int64_t n, float* x_dev);
/* ids reported for rows without explicit ids start here (row shard of a
 * multi-GPU corpus: base = first global row of the shard) */
int rag_index_set_id_base(rag_index_t* h, int64_t base);
/* Optional acceleration of large batches (vector_db/search_optimizer.py:276-323 batch_search,
 * k <= 32, from 2 queries on a corpus of 262144+ rows and from 1024 queries below, dim % 64 == 0, FLAT): mode 1 keeps a bf16 copy of the rows, runs the
 * contraction on bf16 matrix cores and keeps every row whose approximate score is within a
 * proven error margin of the k-th best; those rows are re-scored in fp32 and the k best are
 * returned, so ids and scores are those of the fp32 path.  A single query with k <= 256 scans the
 * bf16 rows instead of the fp32 ones under the same guarantee (half the bytes).  Costs dim*2 bytes per row of device
 * memory.  mode 0 (default) switches it off and frees the copy.  The copy follows rows that are
 * added; a caller who rewrites rows of an adopted buffer in place switches the mode off and on. */
int rag_index_set_prefilter(rag_index_t* h, int mode);
int64_t rag_index_ntotal(const rag_index_t* h);
int rag_index_dim(const rag_index_t* h);
/* RAG_METRIC_* / RAG_KIND_* of the index (what a snapshot was written with); -1 for NULL */
int rag_index_metric(const rag_index_t* h);
int rag_index_kind(const rag_index_t* h);
/* vector_db/qdrant_client.py:826-883 soft delete (payload is_active) and
 * restore: rows[i] are row positions; active = 0 hides them from every search. */
int rag_index_set_active(rag_index_t* h, int64_t n, const int64_t* rows, int active);
/* vector_db/storage_manager.py:558-602 get_vector_by_chunk_id ->
 * qdrant retrieve(with_vectors): copy rows (by row position) to out[n x dim]. */
int rag_index_reconstruct(rag_index_t* h, int64_t n, const int64_t* rows, float* out);

/* vector_db/qdrant_client.py:475-621 backup_collection / :623-764 restore_collection:
 * a binary snapshot (header, fp32 rows in insertion order, ids, soft-delete bytes; an IVF index
 * adds its centroids and the list of every row, and comes back with those lists -- nothing is
 * re-assigned on load) in place of the reference's JSON list-of-floats file.  Payloads stay
 * with the caller. */
int rag_index_save(rag_index_t* h, const char* path);
int rag_index_load(const char* path, int device, rag_index_t** out);

/* ---- search ---------------------------------------------------------------
 * vector_db/search_optimizer.py:146-274 search (nq = 1) and :276-323
 * batch_search (nq > 1), i.e. the client.search call at :213-222.
 *   q         nq x dim fp32
 *   k         hits per query (limit), 1..1024.  Every k takes one pass over the corpus per
 *             batch: up to 64 (200 from 2048 queries) the matrix-core kernel selects as it
 *             scores; above that, from 2 queries, the scores of a group of 32768-row segments
 *             are written out, selected per segment and the lists merged on the device
 *   min_score keep hits with score >= min_score (score_threshold); pass
 *             -FLT_MAX for none
 *   nprobe    IVF lists to scan (ignored by FLAT)
 *   row_mask  NULL or ntotal bytes, 1 = row may be returned (payload filters
 *             resolved by the caller, search_optimizer.py:173-201)
 *   D, I      nq x k, best first; ties broken by lower row position; unfilled
 *             slots I = -1, D = -FLT_MAX. */
int rag_index_search(rag_index_t* h, int64_t nq, const float* q, int k, float min_score,
                     int nprobe, const uint8_t* row_mask, float* D, int64_t* I);
/* same with every buffer in device memory; enqueued on `stream` (a hipStream_t,
 * NULL = default stream) and not synchronised. */
int rag_index_search_device(rag_index_t* h, int64_t nq, const float* q_dev, int k,
                            float min_score, int nprobe, const uint8_t* row_mask_dev,
                            float* D_dev, int64_t* I_dev, void* stream);
/* Search restricted to row ranges of a FLAT index ("segments", e.g. the chunks of one
 * document inside a pooled collection): the k best of every (query, segment) pair in
 * ONE pass.  Replaces the per-document / per-strategy loops of
 * vector_db/search_manager.py:879-913 and :998-1015 (one RPC per pair there).
 *   seg_begin/seg_end  nseg row ranges [begin, end) within [0, ntotal)
 *   D, I               [nq][nseg][k], same conventions as rag_index_search */
int rag_index_search_segments(rag_index_t* h, int64_t nq, const float* q, int64_t nseg,
                              const int64_t* seg_begin, const int64_t* seg_end, int k,
                              float min_score, const uint8_t* row_mask, float* D, int64_t* I);
/* The segment search above followed on the device by the fusion of
 * vector_db/search_manager.py:898-932: every hit's score (widened to fp64) times the
 * weight of its query vector (:902); per segment and row the best weighted score, the
 * earlier query vector winning equal scores (:904-909); all segments' entries sorted by
 * that score, best first, equal scores in the order the reference's dictionaries would
 * hold them (segment order, then the first list and rank at which the row shows up);
 * the first `limit` returned (:928-932).  Segments stand for the reference's documents
 * and must not share rows.
 *   weights              nq fp64 (0.4 / 0.6 / 0.3 at :860-874)
 *   k                    hits per (query vector, segment) pair (limit * 2 at :891); nq * k <= 1024
 *   limit                1..1024
 *   out_rows/out_scores/out_strategy   `limit` slots: row position, weighted score, index of
 *                        the query vector that gave it; *n_out entries are filled */
int rag_index_search_segments_fused(rag_index_t* h, int64_t nq, const float* q, const double* weights,
                                    int64_t nseg, const int64_t* seg_begin, const int64_t* seg_end,
                                    int k, float min_score, const uint8_t* row_mask, int limit,
                                    int64_t* out_rows, double* out_scores, int32_t* out_strategy,
                                    int32_t* n_out);
/* ---- caller filters kept on the device ---------------------------------------
 * The reference sends the same payload filter with every search of a user
 * (vector_db/search_manager.py:893 `{"user_id": ...}`, turned into Filter(must=...) at
 * vector_db/search_optimizer.py:173-201 and evaluated by the server against its payload
 * index).  A row_mask argument is copied to the device on every call -- one byte per row,
 * 10 MB per search of a 10M-row index; a filter object holds those bytes in device memory
 * instead, so a repeated filter costs nothing per search.
 *   rag_filter_create   copies n bytes (1 = row may be returned) to `device`; blocking
 *   rag_filter_free     releases them (waits for searches that still read them)
 * A filter fits an index on the same device with exactly n rows: after rows were added the
 * search calls below answer RAG_ERR_INVALID and the caller makes a new filter.  Filters are
 * read-only and may be shared by threads and indexes.  rag_filter_device_bytes gives the
 * device address for the *_device entry points' row_mask_dev arguments. */
typedef struct rag_filter rag_filter_t;
int rag_filter_create(int device, int64_t n, const uint8_t* row_mask, rag_filter_t** out);
int rag_filter_free(rag_filter_t* f);
int64_t rag_filter_rows(const rag_filter_t* f);
const uint8_t* rag_filter_device_bytes(const rag_filter_t* f);
/* rag_index_search / rag_index_search_segments / rag_index_search_segments_fused with the
 * filter taken from a filter object (NULL = no filter); everything else as documented there */
int rag_index_search_filtered(rag_index_t* h, int64_t nq, const float* q, int k, float min_score,
                              int nprobe, const rag_filter_t* filter, float* D, int64_t* I);
int rag_index_search_segments_filtered(rag_index_t* h, int64_t nq, const float* q, int64_t nseg,
                                       const int64_t* seg_begin, const int64_t* seg_end, int k,
                                       float min_score, const rag_filter_t* filter, float* D, int64_t* I);
int rag_index_search_segments_fused_filtered(rag_index_t* h, int64_t nq, const float* q, const double* weights,
                                             int64_t nseg, const int64_t* seg_begin, const int64_t* seg_end,
                                             int k, float min_score, const rag_filter_t* filter, int limit,
                                             int64_t* out_rows, double* out_scores, int32_t* out_strategy,
                                             int32_t* n_out);
/* Searches of one index share its workspaces: a search on another stream than the one before it waits for that
 * one's end.  By default every *_device search marks its end with an event (a marker packet: 2.4-3 us between two
 * searches on the device).  streams_outlive_searches = 1 is the caller's promise that a stream handed to a search
 * of this index is not destroyed before the NEXT search of the index has been issued (or the index freed); the end
 * of a search is then marked only when a search on another stream needs it, and never between searches that follow
 * each other on one stream.  The host-pointer entry points run on the index's own stream and always work that way.
 * No counterpart in the reference (its searches are blocking RPCs). */
int rag_index_set_stream_lifetime(rag_index_t* h, int streams_outlive_searches);
/* which kernel path the last search on this index used: "none", "empty", "scan" (1-4 queries, HBM scan), "direct" (1-4
 * queries over <= 32768 rows), "narrow" (5-32 queries, one HBM-rate pass on the matrix cores), "gemm" (the selecting
 * matrix-core kernel), "gemm-seeded" (the same with thresholds that start from a bound taken from the leading rows: k in 65..200 on 2^20+ rows,
 * large batches over short corpora), "gemm-scores" (large batch over a short corpus: score matrix + selection), "gemm-segments" (other batches
 * with k above 64: the same per 32768-row segment), "scan-bf16" / "gemm-bf16" (opt-in bf16 first pass + fp32 re-score),
 * "ivf" (per-query list scan), "ivf-listmajor" (batched list scan on the matrix cores), "segments", "segments-fused" */
const char* rag_index_last_path(const rag_index_t* h);

/* multi-GPU step after the all-gather of per-shard results: merge `nlists`
 * result sets, list g at D_dev + g*d_list_stride / I_dev + g*i_list_stride
 * (strides in elements; 0 = dense nq*k), each [nq][k], into the k best per
 * query (same order rule; ids are final ids here; ties between lists go to the
 * lower list).  Device buffers, enqueued on `stream`.  The candidate keys go through a
 * grow-only device block that belongs to the calling host thread: merges that one host thread
 * issues on DIFFERENT streams share it and must not overlap (keep a thread's merges on one
 * stream, or synchronise between them). */
int rag_merge_topk_device(int nlists, int64_t nq, int k, const float* D_dev, int64_t d_list_stride,
                          const int64_t* I_dev, int64_t i_list_stride, float* D_out_dev,
                          int64_t* I_out_dev, int device, void* stream);

/* The same merge for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  It calls no
 * hipMalloc / hipFree: its candidate keys live in a static device array of the library, one copy
 * per GPU held until the process ends (RAG_MERGE_STATIC_MAX_KEYS * 8 bytes = 8 MiB).  Limit:
 * nq * nlists * k <= RAG_MERGE_STATIC_MAX_KEYS, beyond which it returns RAG_ERR_INVALID.  Every
 * call on a GPU shares that GPU's array: calls that overlap on one GPU (different streams or
 * threads) race; calls on one stream are ordered.  The first call on a device looks up the
 * array's address (and k > 64 may set a kernel attribute once): make that call outside a graph
 * capture.  Same arguments and results as rag_merge_topk_device. */
#define RAG_MERGE_STATIC_MAX_KEYS (1 << 20)
int rag_merge_topk_device_static(int nlists, int64_t nq, int k, const float* D_dev, int64_t d_list_stride,
                                 const int64_t* I_dev, int64_t i_list_stride, float* D_out_dev,
                                 int64_t* I_out_dev, int device, void* stream);

/* measurement: while enabled, every search records a HIP event pair around its
 * dominant kernel (the corpus scan / GEMM) on the search's stream.  read
 * synchronises those events and returns the summed kernel time and the number
 * of launches since the last reset. */
int rag_index_profile_enable(rag_index_t* h, int enable);
int rag_index_profile_read(rag_index_t* h, double* kernel_ms, int64_t* launches, int reset);

/* ---- IVF ------------------------------------------------------------------
 * FAISS IndexIVFFlat: centroids given (nlist x dim) or, with centroids == NULL,
 * taken as the first nlist rows of `sample`; later adds go to the list of the
 * centroid with the largest inner product. */
int rag_index_train_ivf(rag_index_t* h, int nlist, const float* centroids,
                        int64_t nsample, const float* sample);

/* ---- score fusion (device kernels over the candidate set) -------------------
 * vector_db/storage_manager.py:653-709 contextual_search fusion.
 * Candidates 0..n-1 in dict insertion order; a[i] / b[i] the content / context
 * score (0.0 where a side is missing).  out_score[i] = a*wa + b*wb; kept when
 * !use_threshold or score >= threshold; order[] = kept candidates by score
 * descending, stable, first k.  Returns the count through n_out. */
int rag_fuse_weighted(int64_t n, const double* a, const double* b, double wa, double wb,
                      int use_threshold, double threshold, int k,
                      int32_t* order, double* out_score, int32_t* n_out);
/* vector_db/search_manager.py:1041-1075 hybrid fusion: a / b = semantic / BM25
 * score per candidate, has_a / has_b presence flags, max_a / max_b the divisors
 * (the caller computes them over the full lists, 1.0 for an empty list).
 * A zero divisor is reported as RAG_ERR_INVALID (the reference's
 * ZeroDivisionError path, :1098-1101, is the caller's to take). */
int rag_fuse_maxnorm(int64_t n, const double* a, const double* b, const uint8_t* has_a,
                     const uint8_t* has_b, double max_a, double max_b, double wa, double wb,
                     int k, int32_t* order, double* out_score, double* out_a, double* out_b,
                     int32_t* n_out);

/* vector_db/mmr_diversifier.py:216-289 _mmr_greedy_selection.
 *   emb       n x d, fp32 (emb_is_f64 = 0) or fp64 (1); arithmetic is fp64
 *   rel       n normalised relevance scores (mmr_diversifier.py:188-200)
 *   out_idx   max_results selected candidate positions in selection order */
int rag_mmr_select(int64_t n, int d, const void* emb, int emb_is_f64, const double* rel,
                   double lambda, double similarity_threshold, int metric, int max_results,
                   int device, int32_t* out_idx, int32_t* n_out);
/* vector_db/mmr_diversifier.py:291-319 compute_similarity_matrix: S_out[i*n + j] (n x n fp64, row-major) =
 * similarity of rows i and j under RAG_MMR_COSINE (rows divided by their norm, a zero row stays zero),
 * RAG_MMR_DOT, or RAG_MMR_EUCLIDEAN (1 / (1 + distance)); the values the MMR selection itself uses, the
 * matrix symmetric bit for bit.  n <= 4096. */
int rag_similarity_matrix(int64_t n, int d, const void* emb, int emb_is_f64, int metric, int device,
                          double* S_out);
/* same, embeddings taken from rows of an index (no host copy of vectors; replaces the vector
 * fetch of vector_db/search_manager.py:1699-1727).  rows[i] are ROW POSITIONS (insertion order,
 * 0 .. ntotal-1), not ids: with an id base or explicit ids the caller maps ids back to positions.
 * FLAT and IVF indexes alike (an IVF index translates positions to its list-ordered storage). */
int rag_mmr_select_rows(rag_index_t* h, int64_t n, const int64_t* rows, const double* rel,
                        double lambda, double similarity_threshold, int metric,
                        int max_results, int32_t* out_idx, int32_t* n_out);

/* ---- one index over several GPUs of this process ---------------------------------------
 * The reference's server is ONE process per worker and its search call blocks in it
 * (vector_db/search_optimizer.py:213-222, vector_db/storage_manager.py:427-511): a caller like
 * that, or any non-Python program, cannot join a torch.distributed world.  rag_multi_* gives it
 * the row-sharded index inside the process: rows of every add are dealt over the devices in
 * contiguous parts, a search runs on all devices at once (one host thread per device) and the
 * per-device answers are merged under the library's order rule (score descending, ties to the
 * lower id).  The merge runs on the first device: every device copies its k-per-query lists
 * there peer to peer and one kernel merges them, so only the merged nq x k answer crosses to the
 * host; where a device cannot reach the first one peer to peer the lists are merged on the host
 * (rag_multi_last_merge says which: "device" / "host").  Same arguments and results as the
 * rag_index_* call of the same name; ids == NULL numbers the rows 0, 1, 2, ... in the order they
 * were added, as a single index would.  An add is all or nothing: when a device fails, the parts
 * the other devices took are given back.  device_ids may name a device more than once (two
 * shards on one GPU). */
typedef struct rag_multi rag_multi_t;
int rag_multi_create(int dim, int metric, int kind, int n_devices, const int* device_ids, rag_multi_t** out);
int rag_multi_free(rag_multi_t* h);
int64_t rag_multi_ntotal(const rag_multi_t* h);
int rag_multi_ndevices(const rag_multi_t* h);
/* rows held by shard s (0 <= s < ndevices) */
int64_t rag_multi_shard_rows(const rag_multi_t* h, int s);
int rag_multi_train_ivf(rag_multi_t* h, int nlist, const float* centroids, int64_t nsample, const float* sample);
int rag_multi_add(rag_multi_t* h, int64_t n, const float* x, const int64_t* ids);
int rag_multi_set_prefilter(rag_multi_t* h, int mode);
/* rows[] / row_mask count rows in the order they were added, over all devices */
int rag_multi_set_active(rag_multi_t* h, int64_t n, const int64_t* rows, int active);
int rag_multi_reconstruct(rag_multi_t* h, int64_t n, const int64_t* rows, float* out);
int rag_multi_search(rag_multi_t* h, int64_t nq, const float* q, int k, float min_score, int nprobe,
                     const uint8_t* row_mask, float* D, int64_t* I);
/* A caller filter for a several-GPU index (see rag_filter_create): ntotal bytes in insertion order, cut into the
 * shards' own row orders and kept on each shard's device.  Fits the index until rows are added. */
typedef struct rag_multi_filter rag_multi_filter_t;
int rag_multi_filter_create(rag_multi_t* h, const uint8_t* row_mask, rag_multi_filter_t** out);
int rag_multi_filter_free(rag_multi_filter_t* f);
int rag_multi_search_filtered(rag_multi_t* h, int64_t nq, const float* q, int k, float min_score, int nprobe,
                              const rag_multi_filter_t* filter, float* D, int64_t* I);
/* where the last rag_multi_search merged the devices' lists: "device", "host" or "none" */
const char* rag_multi_last_merge(const rag_multi_t* h);

/* ---- keyword side: BM25 over a token-id corpus in device memory -----------------------
 * vector_db/search_manager.py:142-180 (CorpusStatsManager.calculate_bm25_score), :182-263
 * (normalize_scores, "minmax") and :1159-1213 (_keyword_search: score every candidate,
 * normalise, drop below min_score, stable sort descending, cut to `limit`).
 * The reference tokenises the candidate text for every query; here a document's kept tokens
 * (preprocess_query, :57-66: lower-cased \b\w+\b words, stop words and words of <= 2
 * characters dropped, or the first three words when nothing is left) are stored ONCE as
 * vocabulary ids, document after document, and a query brings its term ids (a term outside
 * the vocabulary: any negative id) with their idf values (:113-121; 0.5 for a term without
 * one, :170).  fp64, the reference's operation order, no fused multiply-add: bit-identical
 * scores.  avgdl == 0 (the reference's ZeroDivisionError path with its substring fallback,
 * :179-180, which needs the text) is refused with RAG_ERR_INVALID. */
typedef struct rag_bm25 rag_bm25_t;
int rag_bm25_create(int device, rag_bm25_t** out);
void rag_bm25_destroy(rag_bm25_t* h);
int64_t rag_bm25_ndocs(const rag_bm25_t* h);
/* append n_new documents: doc_len[i] kept tokens each, ids concatenated in `tokens` */
int rag_bm25_add(rag_bm25_t* h, int64_t n_new, const int64_t* doc_len, const int32_t* tokens);
int rag_bm25_clear(rag_bm25_t* h);
/* raw BM25 of m candidate documents (cand[i] = document number; NULL: documents 0..m-1;
 * a number outside the corpus scores 0.0 like an empty text), 1..64 query terms */
int rag_bm25_score(rag_bm25_t* h, int64_t m, const int64_t* cand, int n_terms, const int32_t* term_ids,
                   const double* term_idf, double avgdl, double k1, double b, double* raw_out);
int rag_bm25_score_device(rag_bm25_t* h, int64_t m, const int64_t* cand_dev, int n_terms,
                          const int32_t* term_ids_dev, const double* term_idf_dev, double avgdl,
                          double k1, double b, double* raw_out_dev, void* stream);
/* score, normalise (min-max over the m candidates; all equal -> 0.0 if zero else 1.0; a single
 * candidate -> sigmoid 1/(1+exp(-2s))), keep normalised >= min_score when use_min_score, stable
 * sort descending, first `limit` (1..4096; m <= 4096).  out_rows / out_norm / out_raw [limit]:
 * document numbers, normalised and raw scores, padded with -1 / 0.0 beyond *n_out -- the layout
 * rag_index_hybrid_mmr[_device] takes as its keyword hits. */
int rag_bm25_rank(rag_bm25_t* h, int64_t m, const int64_t* cand, int n_terms, const int32_t* term_ids,
                  const double* term_idf, double avgdl, double k1, double b, int use_min_score,
                  double min_score, int limit, int64_t* out_rows, double* out_norm, double* out_raw,
                  int32_t* n_out);
/* device pointers, enqueued on `stream`, not synchronised; raw_scratch_dev [m] doubles.  The
 * single-candidate sigmoid uses the device exp (within 1 ulp of libm; the host form above
 * recomputes it with libm). */
int rag_bm25_rank_device(rag_bm25_t* h, int64_t m, const int64_t* cand_dev, int n_terms,
                         const int32_t* term_ids_dev, const double* term_idf_dev, double avgdl,
                         double k1, double b, int use_min_score, double min_score, int limit,
                         double* raw_scratch_dev, int64_t* out_rows_dev, double* out_norm_dev,
                         double* out_raw_dev, int32_t* n_out_dev, void* stream);

/* ---- search -> hybrid fusion -> MMR as one enqueue ---------------------------------
 * vector_db/search_manager.py:938-970 (_contextual_search: semantic hits + keyword hits ->
 * _combine_semantic_and_bm25_contextual, :1023-1101 -> stable sort descending -> cut) followed
 * by :1620-1681 (_apply_mmr_diversification; vectors come from the index rows instead of the
 * per-result fetch of :1683-1735).  One query.  The steps run back to back on one stream;
 * candidate lists, fused scores and the fp64 rows MMR reads stay in device memory throughout.
 *   topn            semantic hits to fetch (1..1024), with min_score / nprobe / row_mask as in
 *                   rag_index_search
 *   bm25_rows/scores nb (0..1024) keyword hits as (row POSITION, BM25 score); positions outside
 *                   [0, ntotal) are ignored; a repeated row keeps its later score and both
 *                   maxima run over every entry, as the dicts of :1032-1046 do
 *   w_sem, w_bm25   0.65 / 0.35 in the reference (:1072-1075); each side is divided by its own
 *                   maximum (an empty side by 1.0); a zero maximum is the reference's
 *                   ZeroDivisionError path (:1098-1101): status = 1 and the semantic list with
 *                   its raw scores takes the place of the fused list
 *   fuse_limit      the fused list (semantic hits first, then keyword-only rows in order of first
 *                   appearance; stable sort by fused score descending) is cut to this many (1..1024)
 *   MMR             relevance = min-max of the kept fused scores (all equal -> 1.0),
 *                   vector_db/mmr_diversifier.py:188-289 with lambda / threshold / metric,
 *                   max_results picks
 * Results (pointers in rag_hybrid_out_t; ids, scores, n_out required, the rest may be NULL):
 *   ids/scores/picked [max_results] final hits in selection order: id, fused score, position in
 *   the fused list; n_out their count; fused_rows/scores/sem/bm [fuse_limit] the fused list (row
 *   position, fused score, normalised semantic / keyword score); n_fused; status. */
typedef struct rag_hybrid_out {
    int64_t* ids;
    double* scores;
    int32_t* picked;
    int32_t* n_out;
    int64_t* fused_rows;
    double* fused_scores;
    double* fused_sem;
    double* fused_bm;
    int32_t* n_fused;
    int32_t* status;
} rag_hybrid_out_t;
/* host pointers; returns when the results are in the caller's arrays (one synchronisation at the end) */
int rag_index_hybrid_mmr(rag_index_t* h, const float* q, int topn, float min_score, int nprobe,
                         const uint8_t* row_mask, int64_t nb, const int64_t* bm25_rows,
                         const double* bm25_scores, double w_sem, double w_bm25, int fuse_limit,
                         double mmr_lambda, double mmr_threshold, int mmr_metric, int max_results,
                         const rag_hybrid_out_t* out);
/* every buffer (and every pointer inside *out_dev; the struct itself is host memory) in device
 * memory; enqueued on `stream` and not synchronised */
int rag_index_hybrid_mmr_device(rag_index_t* h, const float* q_dev, int topn, float min_score, int nprobe,
                                const uint8_t* row_mask_dev, int64_t nb, const int64_t* bm25_rows_dev,
                                const double* bm25_scores_dev, double w_sem, double w_bm25,
                                int fuse_limit, double mmr_lambda, double mmr_threshold, int mmr_metric,
                                int max_results, const rag_hybrid_out_t* out_dev, void* stream);
/* the same for a batch of queries (no counterpart in the reference, whose search() answers one query per call,
 * vector_db/search_manager.py:938-970 + :1620-1681; for callers that batch requests): ONE search of all nq queries for their
 * topn semantic hits, then fusion and MMR query by query on the same stream.  q_dev [nq][dim]; every query brings nb keyword
 * hits: bm25_rows_dev / bm25_scores_dev [nq][nb] (row positions outside the index are ignored, so a shorter list is padded
 * with -1).  Results: ids_dev / scores_dev [nq][M] with M = min(max_results, fuse_limit, topn + nb), n_out_dev [nq]; ids past
 * a query's count are -1.  Device memory throughout, enqueued on `stream`, not synchronised. */
int rag_index_hybrid_mmr_batch_device(rag_index_t* h, int64_t nq, const float* q_dev, int topn, float min_score, int nprobe,
                                      const uint8_t* row_mask_dev, int64_t nb, const int64_t* bm25_rows_dev,
                                      const double* bm25_scores_dev, double w_sem, double w_bm25, int fuse_limit,
                                      double mmr_lambda, double mmr_threshold, int mmr_metric, int max_results,
                                      int64_t* ids_dev, double* scores_dev, int32_t* n_out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAG_HIP_H */
