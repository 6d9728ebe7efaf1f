#!/usr/bin/env python3
"""Ingest and snapshot rates: rows added from host memory (numpy), from device memory (torch), saved to and loaded from a snapshot."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rag_amd import DeviceIndex

rows, dim = int(os.environ.get("ROWS", 2_000_000)), int(os.environ.get("DIM", 768))
rng = np.random.default_rng(0)
x = rng.standard_normal((rows, dim), dtype=np.float32)
gb = x.nbytes / 1e9
out = {"rows": rows, "dim": dim, "GB": gb}
for metric in ("ip", "cosine"):
    idx = DeviceIndex(dim, metric=metric)
    t0 = time.perf_counter()
    for a in range(0, rows, 250_000):
        idx.add(x[a:a + 250_000])
    torch.cuda.synchronize()
    out[f"add_host_{metric}_GBps"] = gb / (time.perf_counter() - t0)
    idx.close()
xd = torch.from_numpy(x).cuda()
idx = DeviceIndex(dim, metric="cosine")
t0 = time.perf_counter()
idx.add(xd)
torch.cuda.synchronize()
out["add_device_cosine_GBps"] = gb / (time.perf_counter() - t0)
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    p = os.path.join(d, "snap.bin")
    t0 = time.perf_counter()
    idx.save(p)
    out["save_GBps"] = gb / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    back = DeviceIndex.load(p)
    torch.cuda.synchronize()
    out["load_GBps"] = gb / (time.perf_counter() - t0)
    out["loaded_rows"] = len(back)
    back.close()
idx.close()
print(json.dumps(out))
