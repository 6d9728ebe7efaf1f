#!/usr/bin/env python3
"""Batches with k above 64 (the selecting matrix-core kernel stops at 64): ms per batch at 10M x 768 unless ROWS / DIM
say otherwise.  RAG_GEMM_SEGMENTS_MIN_NQ=0 switches the segment walk off (one corpus scan per query, as before)."""
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
idx = DeviceIndex(dim, metric="ip")
idx.adopt(X)
cases = [(8, 100), (32, 100), (64, 100), (64, 64), (128, 200), (256, 200), (512, 200), (1024, 200), (4096, 200), (256, 1024)]
if os.environ.get("CASES"):                          # "nq:k,nq:k"
    cases = [tuple(int(v) for v in c.split(":")) for c in os.environ["CASES"].split(",")]
elif os.environ.get("RAG_GEMM_SEGMENTS_MIN_NQ") == "0":
    cases = [(8, 100), (64, 100), (64, 64)]          # the per-query scans of larger batches take minutes
Q = make_queries(torch, max(4096, max(nq for nq, _ in cases)), dim, dev)
for nq, k in cases:
    idx.search(Q[:nq], k)
    torch.cuda.synchronize()
    steps = 2 if nq >= 512 else 3
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.search(Q[:nq], k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{rows}x{dim} nq={nq} k={k}: {dt * 1e3:9.2f} ms per batch, {nq / dt:9.1f} queries/s, path {idx.last_path}", flush=True)
