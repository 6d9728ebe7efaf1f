#!/usr/bin/env python3
"""Experiment: the list-scan kernel of IVF batches over a whole corpus (every query probes every list), as a
stand-in for a flat search of 17..32 queries: ROWS x DIM rows in NLIST even lists, NQ queries, nprobe = NLIST.
Prints the list-scan kernel time beside the narrow kernel's time on a flat index of the same rows."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rag_amd import DeviceIndex

rows = int(os.environ.get("ROWS", 10_000_000))
dim = int(os.environ.get("DIM", 768))
nlist = int(os.environ.get("NLIST", 512))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
C = torch.nn.functional.normalize(torch.randn(nlist, dim, device=dev, generator=g), dim=1)
ivf = DeviceIndex(dim, metric="ip", kind="ivf", device=0)
ivf.train_ivf(nlist, centroids=C.cpu().numpy())
ivf.reserve(rows)
flat = DeviceIndex(dim, metric="ip", device=0)
flat.reserve(rows)
chunk = 500_000
for r0 in range(0, rows, chunk):
    c = min(chunk, rows - r0)
    a = torch.randint(0, nlist, (c,), device=dev, generator=g)
    blk = torch.nn.functional.normalize(C[a] + (0.6 / dim ** 0.5) * torch.randn(c, dim, device=dev, generator=g), dim=1)
    ivf.add(blk)
    flat.add(blk)
    del blk, a
Q = torch.nn.functional.normalize(torch.randn(64, dim, device=dev, generator=g), dim=1)
ivf.search(Q[:1], 10, nprobe=nlist)
torch.cuda.synchronize()
for nq in [int(v) for v in os.environ.get("BATCHES", "32,24,16").split(",")]:
    for name, idx, kw in (("lists", ivf, {"nprobe": nlist}), ("flat", flat, {})):
        idx.profile(False)
        idx.search(Q[:nq], 10, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            idx.search(Q[:nq], 10, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        idx.profile(True)
        idx.profile_read(True)
        for _ in range(10):
            idx.search(Q[:nq], 10, **kw)
        torch.cuda.synchronize()
        ms, n = idx.profile_read(True)
        print(f"{name}: nq={nq} path={idx.last_path} wall {dt * 1e3:.3f} ms, kernel {ms / max(n, 1):.3f} ms x {n / 10:.1f} per search, "
              f"{rows * dim * 4 / (ms / max(n, 1) * 1e-3) / 1e12:.2f} TB/s", flush=True)
Df, If = flat.search(Q[:32], 10)
Di, Ii = ivf.search(Q[:32], 10, nprobe=nlist)
Df, If, Di, Ii = (torch.as_tensor(v).cpu() for v in (Df, If, Di, Ii))
print("same ids:", bool((If == Ii).all().item()), "max score diff", float((Df - Di).abs().max().item()))
