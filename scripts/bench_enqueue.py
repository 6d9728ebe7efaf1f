#!/usr/bin/env python3
"""Host cost of enqueueing one search (device-resident queries and results, nothing synchronised inside the loop): a corpus small
enough that the GPU is never the bound.  DeviceIndex.search allocates its result tensors per call; search_into writes into caller
memory (the C ABI call + its launches alone)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rag_amd import DeviceIndex

dev = torch.device("cuda", 0)
out = {}
for name, n, d in (("scan_40k_x_96", 40_000, 96), ("direct_2k_x_96", 2_000, 96)):
    x = torch.randn((n, d), device=dev)
    x /= x.norm(dim=1, keepdim=True)
    q = x[:64].clone()
    idx = DeviceIndex(d, metric="ip")
    idx.add(x)
    D = torch.empty((1, 10), dtype=torch.float32, device=dev)
    I = torch.empty((1, 10), dtype=torch.int64, device=dev)
    for label, f in (("search", lambda i: idx.search(q[i % 64:i % 64 + 1], 10)),
                     ("search_into", lambda i: idx.search_into(q[i % 64:i % 64 + 1], 10, D.data_ptr(), I.data_ptr()))):
        for i in range(200):
            f(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5000):
            f(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out[f"{name}_{label}_enqueue_us"] = (t1 - t0) / 5000 * 1e6
        out[f"{name}_{label}_with_drain_us"] = (t2 - t0) / 5000 * 1e6
    out[f"{name}_path"] = idx.last_path
    idx.close()
print(json.dumps(out, indent=1))
