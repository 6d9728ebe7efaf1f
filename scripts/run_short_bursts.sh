#!/bin/bash
# Does the rate of a short HBM-bound launch depend on how long the device has been busy?  The 100k x 384 scan, 200 .. 20000 launches back to back.
set -e
set -o pipefail
export ROWS=100000 DIM=384 NQ=1 WARMUP=5
for steps in 200 2000 20000; do
  STEPS=$steps timeout -k 10 120 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/.*path=/steps='$steps' path=/'
done
# the same launches behind two seconds of the 10M x 768 scan in the same process
timeout -k 10 200 python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from bench import make_corpus, make_queries
from rag_amd import DeviceIndex
dev = torch.device("cuda", 0)
big = DeviceIndex(768, metric="ip"); Xb = make_corpus(torch, 0, 10_000_000, 768, dev); big.adopt(Xb)
small = DeviceIndex(384, metric="ip"); Xs = make_corpus(torch, 0, 100_000, 384, dev); small.adopt(Xs)
Qb = make_queries(torch, 64, 768, dev); Qs = make_queries(torch, 64, 384, dev)
small.profile(True)
for rep in range(3):
    for i in range(400): big.search(Qb[i % 64:i % 64 + 1], 10)
    for i in range(200): small.search(Qs[i % 64:i % 64 + 1], 10)
    torch.cuda.synchronize()
    ms, n = small.profile_read(True)
    print(f"behind 1.8 s of the large scan: small scan kernel {ms / n * 1e3:.2f} us over {n} launches", flush=True)
PY
