#!/usr/bin/env python3
"""Diagnostic (library built with -DRAG_STAMPS): where the single-workgroup kernels behind a top-200 search
-> fusion -> MMR call spend their time.  Prints the stamps (100 MHz wall clock) of the last call."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex, _lib
from rag_amd.index import hybrid_mmr

rows = int(os.environ.get("ROWS", 10_000_000))
dev = torch.device("cuda", 0)
X = make_corpus(torch, 0, rows, 768, dev)
Q = make_queries(torch, 8, 768, dev)
idx = DeviceIndex(768, metric="ip")
idx.adopt(X)
g = torch.Generator(device=dev)
g.manual_seed(1)
bm_rows = torch.randint(0, rows, (100,), generator=g, device=dev)
bm_scores = torch.rand(100, generator=g, device=dev, dtype=torch.float64)
for i in range(4):
    keep = hybrid_mmr(idx, Q[i % 8], 200, bm_rows, bm_scores, max_results=10, mmr_lambda=0.5)
    torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
for tag in ("merge", "fuse"):
    buf = (C.c_longlong * 64)()
    rc = getattr(lib, "rag_debug_stamps_" + tag)(buf)
    v = list(buf)
    base = v[0]
    print(tag, "rc", rc, "us since stamp 0:", [round((x - base) / 100.0, 2) for x in v[:8]], "extra:", v[8:12])
