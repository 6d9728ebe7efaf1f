#!/usr/bin/env python3
"""Step time of one rank of the 8-GPU strong-scaling run (1.25M x 768 rows per rank) with the
exchange step switched on in a one-rank RCCL group: local scan, all-gather, device merge."""
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex
from rag_amd.sharded import ShardedIndex

rows = int(os.environ.get("ROWS", 1_250_000))
dev = torch.device("cuda", 0)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
X = make_corpus(torch, 0, rows, 768, dev)
Q = make_queries(torch, 4096, 768, dev)
idx = DeviceIndex(768, metric="ip")
idx.adopt(X)
sh = ShardedIndex(idx)
for label, fn in (("local search only", idx.search), ("with all-gather + merge", None)):
    if fn is None:
        sh.exchange_when_alone = True
        fn = sh.search
    for nq, steps in ((1, 300), (4096, 5)):
        for i in range(5):
            fn(Q[i:i + nq] if nq == 1 else Q[:nq], 10)
        torch.cuda.synchronize()
        idx.profile(True)
        idx.profile_read(True)
        t0 = time.perf_counter()
        for i in range(steps):
            fn(Q[i % 64:i % 64 + nq] if nq == 1 else Q[:nq], 10)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, n = idx.profile_read(True)
        idx.profile(False)
        print(f"{label}: batch {nq}: {dt / steps * 1e3:.4f} ms/step, dominant kernel {ms / max(n, 1):.4f} ms", flush=True)
dist.destroy_process_group()
