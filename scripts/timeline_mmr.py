#!/usr/bin/env python3
"""MMR 200 -> 10 over rows of a device index, a few calls, for a rocprofv3 kernel trace; prints host time per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rag_amd import DeviceIndex
from rag_amd.mmr_diversifier import mmr_select_rows, normalise_relevance

rng = np.random.default_rng(0)
X = rng.standard_normal((100_000, 768)).astype(np.float32)
X /= np.linalg.norm(X, axis=1, keepdims=True)
idx = DeviceIndex(768, metric="ip")
idx.add(X)
D, I = idx.search(X[:1] + 0.1, 200)
rel = normalise_relevance([float(s) for s in D[0]])
for _ in range(5):
    mmr_select_rows(idx, I[0], rel, 0.5, 0.8, "cosine", 10)
t0 = time.perf_counter()
for _ in range(20):
    mmr_select_rows(idx, I[0], rel, 0.5, 0.8, "cosine", 10)
print(f"host time per call {(time.perf_counter() - t0) / 20 * 1e6:.1f} us")
