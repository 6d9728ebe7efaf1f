#!/usr/bin/env python3
"""The L2 metric at the headline shape (10M x 768 unless ROWS / DIM say otherwise): batch 1, 32 and 4096
on an 'l2' index beside the same corpus as an inner-product index (what the extra column and the
finishing kernel cost).  Device-resident queries and results."""
import json
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
out = {"rows": rows, "dim": dim}
for metric in ("ip", "l2"):
    X = make_corpus(torch, 0, rows, dim, dev)
    Q = make_queries(torch, 4096, dim, dev)
    idx = DeviceIndex(dim, metric=metric)
    if metric == "ip":
        idx.adopt(X)
    else:
        idx.reserve(rows)
        step = 1_000_000
        for r0 in range(0, rows, step):
            idx.add(X[r0:r0 + step])
        del X
        torch.cuda.empty_cache()
    for B, steps in ((1, 50), (32, 20), (4096, 3)):
        idx.search(Q[:B], 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            idx.search(Q[i % 8:i % 8 + B] if B == 1 else Q[:B], 10)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[f"{metric}_batch_{B}"] = {"ms_per_step": dt * 1e3, "queries_per_s": B / dt, "path": idx.last_path}
        print(metric, B, f"{dt * 1e3:.3f} ms", idx.last_path, flush=True)
    idx.close()
    X = None
    torch.cuda.empty_cache()
print(json.dumps(out))
