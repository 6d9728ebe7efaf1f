#!/usr/bin/env python3
"""Batch-1 latency on small and medium corpora (device-resident queries and results)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex

dev = torch.device("cuda", 0)
for rows, dim, k in ((100_000, 384, 10), (16_384, 1024, 64), (1_000_000, 768, 10), (1_000_000, 768, 200)):
    X = make_corpus(torch, 0, rows, dim, dev)
    Q = make_queries(torch, 64, dim, dev)
    idx = DeviceIndex(dim, metric="ip")
    idx.adopt(X)
    for i in range(20):
        idx.search(Q[i % 64:i % 64 + 1], k)
    torch.cuda.synchronize()
    idx.profile(True)
    idx.profile_read(True)
    steps = 300
    t0 = time.perf_counter()
    for i in range(steps):
        idx.search(Q[i % 64:i % 64 + 1], k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, n = idx.profile_read(True)
    print(f"{rows}x{dim} k={k}: {dt / steps * 1e6:8.1f} us/query wall, scan kernel {ms / n * 1e3:7.1f} us "
          f"({rows * dim * 4 / (ms / n * 1e-3) / 1e12:.2f} TB/s)", flush=True)
    idx.close()
    del X
