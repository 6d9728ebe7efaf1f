#!/usr/bin/env python3
"""Randomised parity run of the bf16 first pass in the narrow shapes (2..64 queries, k <= 32, corpus of
262144+ rows) against the CPU oracle: varying row norms, exact ties, near-duplicate clusters (many keys
within the margin, which forces the fp32 fall-back), masks and min_score.  CASES=n, SEED=s."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import flat
from rag_amd import DeviceIndex

rng = np.random.default_rng(int(os.environ.get("SEED", 7)))
bad = 0
t0 = time.time()
for case in range(int(os.environ.get("CASES", 60))):
    d = int(rng.choice([64, 128, 256]))
    n = int(rng.integers(262_144, 420_000))
    nq = int(rng.integers(2, 65))
    k = int(rng.choice([1, 2, 5, 10, 11, 16, 17, 20, 32]))
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    kind = rng.uniform()
    if kind < 0.3:
        X *= rng.uniform(0.5, 1.5, (n, 1)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    if kind > 0.6:                                  # a cluster of near duplicates around the first queries
        m = int(rng.choice([40, 300, 3000]))
        at = rng.integers(0, n, m)
        X[at] = Q[0] + float(rng.choice([0.0, 1e-3, 1e-2])) * rng.standard_normal((m, d)).astype(np.float32)
    mask = (rng.uniform(size=n) < 0.6).astype(np.uint8) if rng.uniform() < 0.3 else None
    min_score = float(rng.choice([-1e30, 0.0, 0.1]))
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    idx.set_prefilter(True)
    D, I = idx.search(Q, k, row_mask=mask, min_score=min_score)
    D2, I2 = idx.search(Q, k, row_mask=mask, min_score=min_score)
    path = idx.last_path
    idx.close()
    Dr, Ir = flat.flat_ip_search(X, Q, k, row_mask=mask, min_score=min_score)
    diff = flat.compare_topk(D, I, Dr, Ir, rtol=1e-5, atol=1e-6)
    same = np.array_equal(D, D2) and np.array_equal(I, I2)
    ok = diff == [] and same and path == "gemm-bf16"
    bad += not ok
    print(f"case {case} n={n} d={d} nq={nq} k={k} kind={kind:.2f} mask={mask is not None} path={path} "
          f"{'ok' if ok else 'FAILED ' + str(diff[:2]) + ' repeat=' + str(same)} ({time.time() - t0:.0f}s)", flush=True)
print("failures:", bad, flush=True)
sys.exit(1 if bad else 0)
