#!/usr/bin/env python3
"""Kernel-time probe for the scan path: prints dominant-kernel ms for a batch size.
Knobs come from the environment (RAG_SCAN_BLOCKS, RAG_SCAN_NT)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex

rows = int(os.environ.get("ROWS", 10_000_000))
dim = int(os.environ.get("DIM", 768))
nq = int(os.environ.get("NQ", 1))
k = int(os.environ.get("K", 10))
steps = int(os.environ.get("STEPS", 30))
dev = torch.device("cuda", 0)
X = make_corpus(torch, 0, rows, dim, dev)
Q = make_queries(torch, max(nq, 64), dim, dev)
idx = DeviceIndex(dim, metric="ip")
idx.adopt(X)
if os.environ.get("PREFILTER", "0") == "1":
    idx.set_prefilter(True)
idx.profile(True)
for i in range(int(os.environ.get("WARMUP", 5))):
    idx.search(Q[:nq], k)
torch.cuda.synchronize()
idx.profile_read(True)
t0 = time.perf_counter()
for i in range(steps):
    idx.search(Q[i % 8:i % 8 + nq] if nq < 64 else Q[:nq], k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms, n = idx.profile_read(True)
gb = rows * dim * 4 / 1e9
print(f"rows={rows} dim={dim} nq={nq} k={k} blocks={os.environ.get('RAG_SCAN_BLOCKS','dflt')} nt={os.environ.get('RAG_SCAN_NT','auto')} "
      f"path={idx.last_path} kernel_ms={ms/n:.4f} GB/s={gb/(ms/n)*1e3:.0f} wall_ms/step={dt/steps*1e3:.3f} launches/step={n/steps:.1f}", flush=True)
