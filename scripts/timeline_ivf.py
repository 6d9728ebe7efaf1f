#!/usr/bin/env python3
"""IVF-flat batch-1 searches at the per-GPU share of the 100M x 1024 configuration (clustered rows),
for a rocprofv3 kernel trace; prints wall time per query."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rag_amd import DeviceIndex

rows = int(os.environ.get("ROWS", 12_500_000))
dim, nlist, nprobe, k = 1024, 16384, 64, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(99)
C = torch.nn.functional.normalize(torch.randn(nlist, dim, device=dev, generator=g), dim=1)
ivf = DeviceIndex(dim, metric="ip", kind="ivf")
ivf.train_ivf(nlist, centroids=C.cpu().numpy())
ivf.reserve(rows)
keep = None
for r0 in range(0, rows, 500_000):
    c = min(500_000, rows - r0)
    a = torch.randint(0, nlist, (c,), device=dev, generator=g)
    blk = torch.nn.functional.normalize(C[a] + (0.6 / dim ** 0.5) * torch.randn(c, dim, device=dev, generator=g), dim=1)
    if keep is None:
        keep = blk[:64].clone()
    ivf.add(blk)
    del blk
Q = torch.nn.functional.normalize(keep + (0.3 / dim ** 0.5) * torch.randn(64, dim, device=dev, generator=g), dim=1)
for i in range(10):
    ivf.search(Q[i:i + 1], k, nprobe=nprobe)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for i in range(n):
    ivf.search(Q[i % 64:i % 64 + 1], k, nprobe=nprobe)
torch.cuda.synchronize()
print(f"wall per query {(time.perf_counter() - t0) / n * 1e6:.1f} us, path {ivf.last_path}")
