#!/usr/bin/env python3
"""IVF-flat at the per-GPU size of configuration 5 (100M x 1024 over 8 GPUs =
12.5M rows per GPU, nlist 16384, nprobe 64): build time, queries/sec for batch 1
and 256, achieved HBM rate of the list scan, recall of the exhaustive answer."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex

rows = int(os.environ.get("ROWS", 12_500_000))
dim = int(os.environ.get("DIM", 1024))
nlist = int(os.environ.get("NLIST", 16384))
nprobe = int(os.environ.get("NPROBE", 64))
k = 10
dev = torch.device("cuda", 0)
clustered = os.environ.get("CLUSTERED", "1") == "1"
if clustered:
    # nlist unit centres, each row = its centre + noise of norm ~0.6, re-normalised;
    # a query = a corpus row + noise of norm ~0.3: its neighbours sit in few lists
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    C = torch.nn.functional.normalize(torch.randn(nlist, dim, device=dev, generator=g), dim=1)
    X = torch.empty(rows, dim, device=dev)
    for r0 in range(0, rows, 500_000):
        c = min(500_000, rows - r0)
        a = torch.randint(0, nlist, (c,), device=dev, generator=g)
        X[r0:r0 + c] = torch.nn.functional.normalize(
            C[a] + (0.6 / dim ** 0.5) * torch.randn(c, dim, device=dev, generator=g), dim=1)
    pick = torch.randint(0, rows, (4096,), device=dev, generator=g)
    Q = torch.nn.functional.normalize(X[pick] + (0.3 / dim ** 0.5) * torch.randn(4096, dim, device=dev, generator=g), dim=1)
    cent = C.cpu().numpy()
else:
    X = make_corpus(torch, 0, rows, dim, dev)
    Q = make_queries(torch, 4096, dim, dev)
    cent = X[torch.arange(0, rows, rows // nlist, device=dev)[:nlist]].cpu().numpy()
ivf = DeviceIndex(dim, metric="ip", kind="ivf")
ivf.train_ivf(nlist, centroids=cent)
ivf.reserve(rows)
t0 = time.perf_counter()
chunk = 1_000_000
for r0 in range(0, rows, chunk):
    ivf.add(X[r0:r0 + chunk])
t_add = time.perf_counter() - t0
t0 = time.perf_counter()
ivf.search(Q[:1], k, nprobe=nprobe)          # first search packs the lists
torch.cuda.synchronize()
t_pack = time.perf_counter() - t0
flat = DeviceIndex(dim, metric="ip")
flat.adopt(X)
out = {"clustered_corpus": clustered, "rows": rows, "dim": dim, "nlist": nlist, "nprobe": nprobe, "add_assign_s": t_add, "first_search_pack_s": t_pack}
batches = [int(b) for b in os.environ.get("BATCHES", "1,256,4096").split(",")]
for nq, steps in [(b, 50 if b == 1 else (5 if b <= 512 else 3)) for b in batches]:
    ivf.profile(True)
    ivf.search(Q[:nq], k, nprobe=nprobe)
    torch.cuda.synchronize()
    ivf.profile_read(True)
    t0 = time.perf_counter()
    for i in range(steps):
        D, I = ivf.search(Q[:nq], k, nprobe=nprobe)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, launches = ivf.profile_read(True)
    ivf.profile(False)
    Df, If = flat.search(Q[:nq], k)
    hit = np.mean([len(set(a) & set(b)) / k for a, b in zip(I.cpu().numpy(), If.cpu().numpy())])
    scanned = nq * nprobe * (rows / nlist) * dim * 4          # expected bytes per batch (uniform lists)
    out[f"batch_{nq}"] = {"qps": nq * steps / dt, "ms_per_batch": dt / steps * 1e3,
                          "scan_kernel_ms": ms / max(launches, 1), "scan_GBps_expected_bytes": scanned / (ms / max(launches, 1) * 1e-3) / 1e9,
                          "recall_at_10_vs_exhaustive": float(hit), "path": ivf.last_path}
print(json.dumps(out))
