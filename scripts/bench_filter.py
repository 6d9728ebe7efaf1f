#!/usr/bin/env python3
"""A repeated caller filter at 10M x 768 (top-10, one query per call, host arrays in and out): its bytes passed as row_mask with
every call against the same bytes kept on the device as a filter object."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex

rows, dim = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("DIM", 768))
dev = torch.device("cuda", 0)
X = make_corpus(torch, 0, rows, dim, dev)
Q = make_queries(torch, 64, dim, dev).cpu().numpy()
idx = DeviceIndex(dim, metric="ip")
idx.adopt(X)
mask = (np.random.default_rng(3).uniform(size=rows) < 0.5).astype(np.uint8)
flt = idx.make_filter(mask)


def timed(f, n=40):
    for i in range(3):
        f(i)
    t0 = time.perf_counter()
    for i in range(n):
        f(i)
    return (time.perf_counter() - t0) / n * 1e3


out = {"rows": rows, "dim": dim}
out["no_filter_ms"] = timed(lambda i: idx.search(Q[i % 64:i % 64 + 1], 10))
out["row_mask_every_call_ms"] = timed(lambda i: idx.search(Q[i % 64:i % 64 + 1], 10, row_mask=mask))
out["filter_object_ms"] = timed(lambda i: idx.search(Q[i % 64:i % 64 + 1], 10, row_mask=flt))
a = idx.search(Q[:4], 10, row_mask=mask)
b = idx.search(Q[:4], 10, row_mask=flt)
out["same_answers"] = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))
print(json.dumps(out))
