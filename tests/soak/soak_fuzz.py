#!/usr/bin/env python3
"""Longer run of the randomised parity checks of tests/test_fuzz_gpu.py over more seeds
(SEEDS=first:last, IVF_SEEDS=first:last), each search issued twice (answers must repeat bit
for bit), plus random MMR cases against the oracle.  Prints one line per seed."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import test_fuzz_gpu as fz
from oracle import flat, mmr as ommr
from rag_amd import DeviceIndex
from rag_amd.mmr_diversifier import mmr_select_device, normalise_relevance


def span(name, dflt):
    a, b = os.environ.get(name, dflt).split(":")
    return range(int(a), int(b))


bad = 0
t0 = time.time()
for seed in span("SEEDS", "100:160"):
    try:
        fz.test_random_configurations(seed)
        print(f"flat seed {seed} ok ({time.time() - t0:.0f}s)", flush=True)
    except AssertionError as e:
        bad += 1
        print(f"flat seed {seed} FAILED: {str(e)[:400]}", flush=True)
for seed in span("IVF_SEEDS", "100:120"):
    try:
        fz.test_random_ivf_configurations(seed)
        print(f"ivf seed {seed} ok ({time.time() - t0:.0f}s)", flush=True)
    except AssertionError as e:
        bad += 1
        print(f"ivf seed {seed} FAILED: {str(e)[:400]}", flush=True)
# determinism: the same search twice, every path
rng = np.random.default_rng(7)
for rep in range(30):
    d = int(rng.choice([64, 128, 384, 768]))
    n = int(rng.integers(1000, 80_000))
    nq = int(rng.choice([1, 4, 9, 40, 130, 1100]))
    k = int(rng.choice([1, 10, 64, 100]))
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    idx = DeviceIndex(d, metric="cosine")
    idx.add(X)
    if rep % 3 == 0:
        idx.set_prefilter(True)
    D1, I1 = idx.search(Q, k)
    D2, I2 = idx.search(Q, k)
    if not (np.array_equal(I1, I2) and np.array_equal(D1, D2)):
        bad += 1
        print(f"determinism FAILED n={n} d={d} nq={nq} k={k} path={idx.last_path}", flush=True)
    idx.close()
print(f"determinism done ({time.time() - t0:.0f}s)", flush=True)
# MMR against the oracle: clustered candidates so that the penalty branch fires
for rep in range(40):
    n = int(rng.integers(2, 260))
    d = int(rng.choice([8, 64, 384, 768]))
    centres = rng.standard_normal((max(1, n // 6), d))
    E = centres[rng.integers(0, centres.shape[0], n)] + 0.15 * rng.standard_normal((n, d))
    if rep % 5 == 0:
        E[n // 2] = E[0]
    if rep % 7 == 0:
        E[n - 1] = 0.0
    rel = rng.uniform(0, 1, n)
    if rep % 4 == 0:
        rel[: n // 2] = rel[0]
    lam = float(rng.choice([0.0, 0.3, 0.5, 0.7, 1.0]))
    thr = float(rng.choice([0.0, 0.5, 0.8]))
    metric = str(rng.choice(["cosine", "dot_product", "euclidean"]))
    m = int(rng.integers(1, n + 1))
    want = ommr.mmr_select([e for e in E], list(rel), lam, m, thr, metric)
    got = mmr_select_device(E, normalise_relevance(list(rel)), lam, thr, metric, m)
    if list(got) != list(want):
        bad += 1
        print(f"mmr FAILED n={n} d={d} lam={lam} thr={thr} metric={metric} m={m}", flush=True)
print(f"mmr done ({time.time() - t0:.0f}s)", flush=True)
print("SOAK", "FAILED" if bad else "ok", bad, flush=True)
sys.exit(1 if bad else 0)
