#!/usr/bin/env python3
"""The search -> fusion -> MMR pipeline (device variant) a few times on a 10M x 768 corpus, for a
rocprofv3 kernel trace; prints wall time per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex
from rag_amd.index import hybrid_mmr

rows = int(os.environ.get("ROWS", 10_000_000))
dev = torch.device("cuda", 0)
X = make_corpus(torch, 0, rows, 768, dev)
Q = make_queries(torch, 8, 768, dev)
idx = DeviceIndex(768, metric="ip")
idx.adopt(X)
g = torch.Generator(device=dev)
g.manual_seed(1)
bm_rows = torch.randint(0, rows, (100,), generator=g, device=dev)
bm_scores = torch.rand(100, generator=g, device=dev, dtype=torch.float64)
keep = []
for i in range(4):
    keep.append(hybrid_mmr(idx, Q[i % 8], 200, bm_rows, bm_scores, max_results=10, mmr_lambda=0.5))
torch.cuda.synchronize()
n = 12
t0 = time.perf_counter()
for i in range(n):
    keep.append(hybrid_mmr(idx, Q[i % 8], 200, bm_rows, bm_scores, max_results=10, mmr_lambda=0.5))
torch.cuda.synchronize()
print(f"wall per call {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
