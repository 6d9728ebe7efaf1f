#!/usr/bin/env python3
"""A large batch over a short corpus: the IVF coarse step in isolation (16384 centroids x 1024, 4096 queries,
top-64) and a few neighbouring shapes.  Prints ms per batch and the matrix-core rate; knobs come from the
environment (RAG_GEMM_MIN_TILES_SHORT, RAG_GEMM_QT, RAG_GEMM_WGS)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex

dev = torch.device("cuda", 0)
shapes = ((16384, 1024, 4096, 64), (16384, 1024, 1024, 64), (65536, 768, 4096, 10), (100_000, 384, 4096, 10), (4096, 1024, 4096, 10))
if os.environ.get("SHAPES"):        # SHAPES="rows:dim:nq:k,..."
    shapes = tuple(tuple(int(v) for v in t.split(":")) for t in os.environ["SHAPES"].split(","))
for rows, dim, nq, k in shapes:
    X = make_corpus(torch, 0, rows, dim, dev)
    Q = make_queries(torch, nq, dim, dev)
    idx = DeviceIndex(dim, metric="ip")
    idx.adopt(X)
    for _ in range(3):
        idx.search(Q, k)
    torch.cuda.synchronize()
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.search(Q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{rows}x{dim} nq={nq} k={k}: {dt * 1e3:7.3f} ms  {2.0 * nq * rows * dim / dt / 1e12:6.1f} TFLOP/s  path {idx.last_path}", flush=True)
    idx.close()
    del X
