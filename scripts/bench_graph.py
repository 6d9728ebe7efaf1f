#!/usr/bin/env python3
"""A batch-1 search replayed from a HIP graph: the launches of one rag_index_search_device call (query preparation,
scan or the IVF chain, merge) captured once on a side stream and replayed per query, against the same call enqueued
directly.  Results must be identical; prints microseconds per query both ways."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex

dev = torch.device("cuda", 0)


def run(name, idx, Q, k, steps=400, **kw):
    q = Q[0:1].clone()
    D = torch.empty((1, k), dtype=torch.float32, device=dev)
    I = torch.empty((1, k), dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        for j in range(3):                                   # sizes every workspace on the stream that is captured
            idx.search_into(q, k, D.data_ptr(), I.data_ptr(), **kw)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        idx.search_into(q, k, D.data_ptr(), I.data_ptr(), **kw)
    # same answers as the direct call
    for j in (1, 5, 9):
        q.copy_(Q[j:j + 1])
        g.replay()
        torch.cuda.synchronize()
        Dg, Ig = D.clone(), I.clone()
        Dd, Id = idx.search(Q[j:j + 1], k, **kw)
        assert torch.equal(Ig, Id) and torch.equal(Dg, Dd), (name, j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(steps):
        q.copy_(Q[j % 64:j % 64 + 1])
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / steps
    with torch.cuda.stream(side):
        t0 = time.perf_counter()
        for j in range(steps):
            idx.search_into(Q[j % 64:j % 64 + 1], k, D.data_ptr(), I.data_ptr(), **kw)
        side.synchronize()
        t_direct = (time.perf_counter() - t0) / steps
    print(f"{name}: graph replay {t_graph * 1e6:7.1f} us/query, direct enqueue {t_direct * 1e6:7.1f} us/query", flush=True)


X = make_corpus(torch, 0, 100_000, 384, dev)
Q = make_queries(torch, 64, 384, dev)
idx = DeviceIndex(384, metric="ip")
idx.adopt(X)
run("flat 100k x 384, top-10", idx, Q, 10)
idx.close()

X = make_corpus(torch, 0, 1_000_000, 768, dev)
Q = make_queries(torch, 64, 768, dev)
idx = DeviceIndex(768, metric="ip")
idx.adopt(X)
run("flat 1M x 768, top-10", idx, Q, 10)
idx.close()

rows, dim, nlist = int(os.environ.get("IVF_ROWS", 2_000_000)), 1024, 16384
X = make_corpus(torch, 0, rows, dim, dev)
Q = make_queries(torch, 64, dim, dev)
iv = DeviceIndex(dim, metric="ip", kind="ivf")
iv.train_ivf(nlist, centroids=X[:nlist].cpu().numpy())
iv.add(X)
iv.search(Q[:1], 10, nprobe=64)                               # packs the lists
run(f"IVF {rows} x 1024, nlist 16384, nprobe 64, top-10", iv, Q, 10, nprobe=64)
iv.close()
