#!/usr/bin/env python3
"""Latency of the reference's per-document search fan-out
(vector_db/search_manager.py:879-913: N_docs x strategies sequential searches)
done (a) call by call through the storage-manager facade and (b) as ONE
segment-restricted device pass over pooled collections.  Also times fusion and
MMR of a 200-candidate set (configuration "top-200 -> fusion + MMR -> top-10")."""
import asyncio
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rag_amd import DeviceIndex, fusion
from rag_amd.mmr_diversifier import mmr_select_device, mmr_select_rows, normalise_relevance
from rag_amd.pooled import PooledCollections
from rag_amd.storage_manager import VectorStorageManager

run = asyncio.get_event_loop().run_until_complete
rng = np.random.default_rng(0)
ndocs, d, limit = int(os.environ.get("NDOCS", 100)), 384, 10


def unit(n, dd=d):
    x = rng.standard_normal((n, dd)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


mgr = VectorStorageManager()
pmgr = VectorStorageManager(pooled=True)
pool = PooledCollections(d, metric="cosine")
for doc in range(ndocs):
    n = int(rng.integers(50, 400))
    x = unit(n)
    ids = [f"doc_{doc}_chunk_{i}" for i in range(n)]
    meta = [{"document_id": doc} for _ in range(n)]
    for m in (mgr, pmgr):
        run(m.create_index(f"doc_{doc}", d))
        run(m.add_vectors(f"doc_{doc}", x.tolist(), x.tolist(), meta, ids))
    pool.upsert(f"doc_{doc}", x, [dict(m, chunk_id=c) for m, c in zip(meta, ids)], ids)
Q = unit(3)
names = [f"doc_{doc}" for doc in range(ndocs)]


def loop_fanout():
    out = []
    for name in names:
        for qi in range(3):
            out.append(run(mgr.search_vectors(name, Q[qi].tolist(), "content", limit=2 * limit, score_threshold=0.1)))
    return out


def pooled_fanout():
    return pool.search_many(names, Q, 2 * limit, min_score=0.1)


res = {}
for fn in (loop_fanout, pooled_fanout):
    fn()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        fn()
    res[fn.__name__ + "_ms"] = (time.perf_counter() - t0) / reps * 1e3

# the same fan-out through the engine method (search + formatting + max-fusion), per-collection and pooled manager
from rag_amd.schemas import SearchFilters
from rag_amd.search_engine import HipSearchEngine
strat = [("full_query", Q[0].tolist(), 0.4), ("concepts", Q[1].tolist(), 0.6), ("expanded", Q[2].tolist(), 0.3)]
filt = SearchFilters(document_ids=list(range(ndocs)), min_score=None)
for label, m in (("engine_semantic_search_per_collection_ms", mgr), ("engine_semantic_search_pooled_ms", pmgr)):
    eng = HipSearchEngine(m)
    first = run(eng.semantic_search(strat, filt, limit))
    t0 = time.perf_counter()
    for _ in range(5):
        run(eng.semantic_search(strat, filt, limit))
    res[label] = (time.perf_counter() - t0) / 5 * 1e3
    res[label[:-3] + "_top"] = [r.chunk_id for r in first[:3]]

# the device call alone (H2D of queries and ranges, one scan pass, D2H of the lists)
segs = [tuple(r) for nme in names for r in pool.ranges[nme]]
pool.index.search_segments(Q, segs, 2 * limit, min_score=0.1)
t0 = time.perf_counter()
for _ in range(20):
    pool.index.search_segments(Q, segs, 2 * limit, min_score=0.1)
res["pooled_device_call_ms"] = (time.perf_counter() - t0) / 20 * 1e3

# the same with weighting, best-per-chunk and ranking on the device (D2H of `limit` entries)
w = [0.4, 0.6, 0.3]
pool.index.search_segments_fused(Q, w, segs, 2 * limit, limit, min_score=0.1)
t0 = time.perf_counter()
for _ in range(20):
    pool.index.search_segments_fused(Q, w, segs, 2 * limit, limit, min_score=0.1)
res["pooled_fused_device_call_ms"] = (time.perf_counter() - t0) / 20 * 1e3
# the engine method on the pooled manager without the fused call
pmgr.search_vectors_fused = None
eng = HipSearchEngine(pmgr)
run(eng.semantic_search(strat, filt, limit))
t0 = time.perf_counter()
for _ in range(5):
    run(eng.semantic_search(strat, filt, limit))
res["engine_semantic_search_pooled_list_fusion_ms"] = (time.perf_counter() - t0) / 5 * 1e3
del pmgr.search_vectors_fused

# fusion + MMR on 200 candidates of a 1M x 768 corpus
n, dd = 1_000_000, 768
X = unit(n, dd)
idx = DeviceIndex(dd, metric="ip")
idx.add(X)
q = unit(1, dd)
t0 = time.perf_counter()
D, I = idx.search(q, 200)
res["search_top200_1Mx768_host_api_ms"] = (time.perf_counter() - t0) * 1e3
sem = [(f"k{int(i)}", float(s)) for s, i in zip(D[0], I[0])]
bm = [(f"k{int(i)}", float(rng.uniform())) for i in rng.choice(I[0], 50, replace=False)] + \
     [(f"b{j}", float(rng.uniform())) for j in range(50)]
for _ in range(3):
    t0 = time.perf_counter()
    fused = fusion.hybrid_fuse(sem, bm, 200)
    res["hybrid_fuse_250_candidates_ms"] = (time.perf_counter() - t0) * 1e3
rel = normalise_relevance([float(s) for s in D[0]])
for _ in range(3):
    t0 = time.perf_counter()
    pick = mmr_select_rows(idx, I[0], rel, 0.5, 0.8, "cosine", 10)
    res["mmr_200_to_10_rows_on_device_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    pick2 = mmr_select_device(X[I[0]], rel, 0.5, 0.8, "cosine", 10)
    res["mmr_200_to_10_host_embeddings_ms"] = (time.perf_counter() - t0) * 1e3
assert pick == pick2
res["ndocs"] = ndocs
res["searches_in_fanout"] = 3 * ndocs
print(json.dumps(res))
