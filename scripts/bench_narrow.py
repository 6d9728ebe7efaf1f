#!/usr/bin/env python3
"""Kernel time of small batches (5..32 queries) over the 10M x 768 corpus: ms per batch and the
corpus bytes per second they stand for."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex

rows = int(os.environ.get("ROWS", 10_000_000))
dim = int(os.environ.get("DIM", 768))
dev = torch.device("cuda", 0)
X = make_corpus(torch, 0, rows, dim, dev)
Q = make_queries(torch, 64, dim, dev)
idx = DeviceIndex(dim, metric="ip")
idx.adopt(X)
idx.profile(True)
for nq in [int(v) for v in os.environ.get("BATCHES", "5,8,16,17,32").split(",")]:
    for k in (10, 20):
        for i in range(3):
            idx.search(Q[:nq], k)
        torch.cuda.synchronize()
        idx.profile_read(True)
        t0 = time.perf_counter()
        for i in range(10):
            idx.search(Q[:nq], k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        ms, n = idx.profile_read(True)
        print(f"nq={nq:3d} k={k:2d} path={idx.last_path:7s} wall {dt * 1e3:6.3f} ms  kernel {ms / n:6.3f} ms  "
              f"{rows * dim * 4 / (ms / n * 1e-3) / 1e12:5.2f} TB/s", flush=True)
