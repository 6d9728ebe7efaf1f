#!/usr/bin/env python3
"""Kernel durations of the fused fan-out (run under rocprofv3 --kernel-trace --stats):
NDOCS documents x 3 query vectors, 2*limit hits per pair, ranked to `limit`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rag_amd import DeviceIndex

rng = np.random.default_rng(0)
ndocs, d, limit = int(os.environ.get("NDOCS", 100)), 384, int(os.environ.get("LIMIT", 10))
sizes = rng.integers(50, 400, ndocs)
x = rng.standard_normal((int(sizes.sum()), d)).astype(np.float32)
idx = DeviceIndex(d, metric="cosine")
idx.add(x)
ends = np.cumsum(sizes)
segs = [(int(e - n), int(e)) for e, n in zip(ends, sizes)]
q = rng.standard_normal((3, d)).astype(np.float32)
for _ in range(30):
    rows, scores, strat = idx.search_segments_fused(q, [0.4, 0.6, 0.3], segs, 2 * limit, limit, min_score=0.02)
print(len(rows), scores[:3])
