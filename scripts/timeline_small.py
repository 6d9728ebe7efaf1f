#!/usr/bin/env python3
"""Batch-1 searches on a 100k x 384 corpus (device-resident), for a rocprofv3 kernel trace."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_corpus, make_queries
from rag_amd import DeviceIndex

dev = torch.device("cuda", 0)
X = make_corpus(torch, 0, 100_000, 384, dev)
Q = make_queries(torch, 64, 384, dev)
idx = DeviceIndex(384, metric=os.environ.get("METRIC", "ip"))
idx.adopt(X)
for i in range(60):
    idx.search(Q[i % 64:i % 64 + 1], 10)
torch.cuda.synchronize()
