#!/usr/bin/env python3
"""Per-stage latency of configuration "hybrid semantic+BM25 fusion + MMR (lambda 0.5)
re-rank of top-200 to top-10, 10M x 768": exact top-200 scan, max-normalised
weighted fusion with a keyword result list, MMR selection on device rows."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex, fusion
from rag_amd.mmr_diversifier import mmr_select_rows, normalise_relevance

rows = int(os.environ.get("ROWS", 10_000_000))
dim, topn, final = 768, 200, 10
dev = torch.device("cuda", 0)
X = make_corpus(torch, 0, rows, dim, dev)
Q = make_queries(torch, 32, dim, dev).cpu().numpy()
idx = DeviceIndex(dim, metric="ip")
idx.adopt(X)
if os.environ.get("PREFILTER", "0") == "1":
    idx.set_prefilter(True)
rng = np.random.default_rng(5)
stages = {"search_top200_ms": [], "hybrid_fuse_ms": [], "mmr_200_to_10_ms": [], "total_ms": []}
idx.profile(True)
for it in range(len(Q) + 3):
    q = Q[it % len(Q)][None, :]
    t0 = time.perf_counter()
    D, I = idx.search(q, topn)                                   # host-pointer API, blocking
    t1 = time.perf_counter()
    sem = [(int(i), float(s)) for s, i in zip(D[0], I[0])]
    # keyword list: 50 of the semantic hits + 50 other rows, BM25-like positive scores
    kw_ids = list(rng.choice(I[0], 50, replace=False)) + list(rng.integers(0, rows, 50))
    bm = [(int(i), float(s)) for i, s in zip(kw_ids, rng.gamma(2.0, 2.0, size=100))]
    t2 = time.perf_counter()
    fused = fusion.hybrid_fuse(sem, bm, topn)
    t3 = time.perf_counter()
    cand = np.array([f[0] for f in fused], dtype=np.int64)
    rel = normalise_relevance([f[1] for f in fused])
    pick = mmr_select_rows(idx, cand, rel, 0.5, 0.8, "cosine", final)
    t4 = time.perf_counter()
    if it >= 3:
        stages["search_top200_ms"].append((t1 - t0) * 1e3)
        stages["hybrid_fuse_ms"].append((t3 - t2) * 1e3)
        stages["mmr_200_to_10_ms"].append((t4 - t3) * 1e3)
        stages["total_ms"].append((t1 - t0 + t4 - t2) * 1e3)
ms, launches = idx.profile_read(True)
out = {"rows": rows, "dim": dim, "top_n": topn, "final": final, "queries": len(Q),
       "scan_kernel_ms_k200": ms / launches, "scan_GBps_k200": rows * dim * 4 / (ms / launches * 1e-3) / 1e9}
for k, v in stages.items():
    out[k] = {"mean": float(np.mean(v)), "p50": float(np.median(v)), "max": float(np.max(v))}
out["pipeline_qps_sequential"] = 1e3 / out["total_ms"]["mean"]
print(json.dumps(out))
