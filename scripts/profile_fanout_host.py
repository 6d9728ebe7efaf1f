#!/usr/bin/env python3
"""cProfile of the pooled engine fan-out (host side) -- where the time outside the device call goes."""
import asyncio
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rag_amd.schemas import SearchFilters
from rag_amd.search_engine import HipSearchEngine
from rag_amd.storage_manager import VectorStorageManager

run = asyncio.get_event_loop().run_until_complete
rng = np.random.default_rng(0)
ndocs, d, limit = 100, 384, 10
pmgr = VectorStorageManager(pooled=True)
for doc in range(ndocs):
    n = int(rng.integers(50, 400))
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    ids = [f"doc_{doc}_chunk_{i}" for i in range(n)]
    run(pmgr.create_index(f"doc_{doc}", d))
    run(pmgr.add_vectors(f"doc_{doc}", x.tolist(), x.tolist(), [{"document_id": doc, "user_id": 7} for _ in range(n)], ids))
Q = rng.standard_normal((3, d)).astype(np.float32)
Q /= np.linalg.norm(Q, axis=1, keepdims=True)
strat = [("full_query", Q[0].tolist(), 0.4), ("concepts", Q[1].tolist(), 0.6), ("expanded", Q[2].tolist(), 0.3)]
filt = SearchFilters(document_ids=list(range(ndocs)), min_score=None, user_id=7)
eng = HipSearchEngine(pmgr)
run(eng.semantic_search(strat, filt, limit))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    run(eng.semantic_search(strat, filt, limit))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
