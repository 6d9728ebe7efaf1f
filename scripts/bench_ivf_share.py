#!/usr/bin/env python3
"""The IVF leg of bench.py alone (one GPU's share of the 100M x 1024 configuration: 12.5M rows, nlist 16384,
nprobe 64; batch 1 and batch 4096), for A/B runs of the list-scan kernels:
    RAG_IVF_LISTS_KERNEL=0 python scripts/bench_ivf_share.py     # the LDS-tiled 256 x 32 shape
    python scripts/bench_ivf_share.py                            # the register-streaming kernel
ROWS / NLIST shrink the problem for quick checks."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

args = bench.parse([])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = bench.leg_ivf_share(torch, args, dev, rows=int(os.environ.get("ROWS", 12_500_000)), nlist=int(os.environ.get("NLIST", 16384)))
print(json.dumps({"lists_kernel": os.environ.get("RAG_IVF_LISTS_KERNEL", "1"), **out}))
