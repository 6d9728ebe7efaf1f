#!/usr/bin/env python3
"""What a caller of the drop-in Python surface pays per query on top of the device call: config 1 (100k x 384, top-10)
through VectorStorageManager.search_vectors (Python list in, list of dicts out), through HipSearchOptimizer.search_sync,
through DeviceIndex.search on host arrays and on device tensors."""
import asyncio
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rag_amd.storage_manager import VectorStorageManager

run = asyncio.get_event_loop().run_until_complete
rows, d, k = int(os.environ.get("ROWS", 100_000)), int(os.environ.get("DIM", 384)), 10
rng = np.random.default_rng(1234)
x = rng.standard_normal((rows, d)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
mgr = VectorStorageManager()
run(mgr.create_index("docs", d))
ids = [f"doc_1_chunk_{i}" for i in range(rows)]
meta = [{"document_id": 1, "user_id": 7, "chunk_id": ids[i], "start_char": 0, "end_char": 10, "text": "t"} for i in range(rows)]
step = 20000
for a in range(0, rows, step):
    run(mgr.add_vectors("docs", x[a:a + step].tolist(), x[a:a + step].tolist(), meta[a:a + step], ids[a:a + step]))
Q = rng.standard_normal((64, d)).astype(np.float32)
Q /= np.linalg.norm(Q, axis=1, keepdims=True)
Ql = [q.tolist() for q in Q]
opt = mgr.search_optimizers["docs_content"]
index = opt.index
Qd = torch.from_numpy(Q).cuda()


def timed(f, n=400):
    for i in range(20):
        f(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        f(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


out = {"rows": rows, "dim": d, "k": k}
out["search_vectors_us"] = timed(lambda i: run(mgr.search_vectors("docs", Ql[i % 64], "content", k)))
out["search_vectors_filtered_us"] = timed(lambda i: run(mgr.search_vectors("docs", Ql[i % 64], "content", k, None, {"user_id": 7})))
out["optimizer_search_sync_us"] = timed(lambda i: opt.search_sync(Ql[i % 64], k, 0.0))
out["device_index_host_arrays_us"] = timed(lambda i: index.search(Q[i % 64:i % 64 + 1], k))
out["device_index_device_tensors_us"] = timed(lambda i: index.search(Qd[i % 64:i % 64 + 1], k))
print(json.dumps(out, indent=1))
