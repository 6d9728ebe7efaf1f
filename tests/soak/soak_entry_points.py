#!/usr/bin/env python3
"""Randomised parity run of the entry points behind the callers of the search: segment search, fused
fan-out, search -> fusion -> MMR pipeline (flat and IVF), keyword ranking, the several-GPU index
replayed on one GPU.  Each case draws its shape from its seed (SEEDS=first:last) and goes through the
checker of the collected test of the same entry point; prints one line per seed and section."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import test_bm25_gpu as tb
import test_fanout_gpu as tf
import test_hybrid_pipeline_gpu as tp
from oracle import flat
from rag_amd import DeviceIndex
from rag_amd.bm25 import CorpusStatsManager, DeviceKeywordIndex


def span(name, dflt):
    a, b = os.environ.get(name, dflt).split(":")
    return range(int(a), int(b))


def cut_segments(rng, n, nseg):
    cuts = np.sort(rng.choice(np.arange(1, n), min(nseg, n - 1) - 1, replace=False)) if nseg > 1 else np.zeros(0, int)
    segs = [(int(a), int(b)) for a, b in zip(np.r_[0, cuts], np.r_[cuts, n])]
    if rng.uniform() < 0.3:
        segs.insert(int(rng.integers(0, len(segs))), (5, 5))            # an empty document
    return segs


def segments_case(rng):
    d = int(rng.choice([36, 64, 100, 128, 384, 768, 1024]))
    n = int(rng.integers(50, 30_000))
    nq = int(rng.choice([1, 2, 3, 4]))
    k = int(rng.choice([1, 5, 10, 20, 64, 100, 200]))
    X = tf._unit(rng, n, d)
    if n > 200 and rng.uniform() < 0.3:
        X[n // 2: n // 2 + 30] = X[3]                                   # equal scores inside and across segments
    Q = tf._unit(rng, nq, d)
    segs = cut_segments(rng, n, int(rng.choice([1, 2, 7, 40, 150, 600])))
    mask = (rng.uniform(size=n) < 0.6).astype(np.uint8) if rng.uniform() < 0.4 else None
    min_score = float(rng.choice([-1e30, 0.0, 0.05]))
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    D, I = idx.search_segments(Q, segs, k, min_score=min_score, row_mask=mask)
    if mask is not None:                                                # the same bytes kept on the device as a filter object
        flt = idx.make_filter(mask)
        Df, If = idx.search_segments(Q, segs, k, min_score=min_score, row_mask=flt)
        assert np.array_equal(D, Df) and np.array_equal(I, If), dict(n=n, d=d, nq=nq, k=k, what="filter object, segments")
        Dm, Im = idx.search(Q, k, min_score=min_score, row_mask=mask)
        Dg, Ig = idx.search(Q, k, min_score=min_score, row_mask=flt)
        assert np.array_equal(Dm, Dg) and np.array_equal(Im, Ig), dict(n=n, d=d, nq=nq, k=k, what="filter object, search")
        flt.close()
    for si, (b, e) in enumerate(segs):
        Dr, Ir = flat.flat_ip_search(X[b:e], Q, k, min_score=min_score, row_mask=None if mask is None else mask[b:e])
        Ir = np.where(Ir >= 0, Ir + b, -1)
        bad = flat.compare_topk(D[:, si], I[:, si], Dr, Ir)
        assert bad == [], (dict(n=n, d=d, nq=nq, k=k, seg=(b, e)), bad[:3])
    # the fused fan-out over the same corpus (nq * k at most 1024)
    kf = min(k, 1024 // nq)
    limit = int(rng.choice([1, 10, 100, 1000]))
    weights = [float(w) for w in rng.choice([0.4, 0.6, 0.3, 0.5, 1.0], nq)]
    tf._check(idx, Q, weights, segs, kf, limit, min_score=float(rng.choice([0.0, 0.02, 0.1])), mask=mask)
    idx.close()


def pipeline_case(rng):
    d = int(rng.choice([64, 128, 256, 384, 768]))
    n = int(rng.integers(300, 50_000))
    X, cent = tp._clustered(rng, n, d, ncl=int(rng.choice([5, 20, 60])), noise=float(rng.choice([0.05, 0.2, 0.5])))
    q = cent[int(rng.integers(0, cent.shape[0]))] + 0.1 * rng.standard_normal(d)
    q = (q / np.linalg.norm(q)).astype(np.float32)
    topn = int(rng.choice([5, 17, 64, 100, 200, 300]))
    topn = min(topn, n)
    nb = int(rng.choice([0, 1, 10, 100, 400]))
    idx = DeviceIndex(d, metric="ip")
    ids = None
    id_base = 0
    if rng.uniform() < 0.3:
        ids = rng.permutation(10 * n)[:n].astype(np.int64)
        idx.add(X, ids=ids)
    else:
        if rng.uniform() < 0.5:
            id_base = int(rng.integers(1, 1 << 40))
            idx.set_id_base(id_base)
        idx.add(X)
    bm_rows = rng.integers(0, n, nb)
    if nb and rng.uniform() < 0.7:                                     # half the keyword hits among the semantic hits
        D, I = idx.search(q, topn)
        rows = I[0][I[0] >= 0]
        if ids is not None:
            back = {int(v): r for r, v in enumerate(ids)}
            rows = np.array([back[int(i)] for i in rows], dtype=np.int64)
        else:
            rows = rows - id_base
        if rows.size:
            bm_rows[: nb // 2] = rng.choice(rows, nb // 2)
    bm_scores = rng.gamma(2.0, 2.0, size=nb) if rng.uniform() < 0.9 else np.zeros(nb)
    mask = (rng.uniform(size=n) < 0.6).astype(np.uint8) if rng.uniform() < 0.3 else None
    lam = float(rng.choice([0.0, 0.3, 0.5, 0.7, 1.0]))
    thr = float(rng.choice([0.0, 0.5, 0.8]))
    metric = str(rng.choice(["cosine", "dot_product", "euclidean"]))
    fl = None if rng.uniform() < 0.5 else int(rng.integers(1, min(1024, topn + nb) + 1))
    tp._check(idx, X, q, topn, bm_rows, bm_scores, fuse_limit=fl, max_results=int(rng.choice([1, 10, 25, 60])), lam=lam, thr=thr,
              metric=metric, ids=ids, id_base=id_base, row_mask=mask, min_score=None if rng.uniform() < 0.6 else 0.3)
    idx.close()


def pipeline_batch_case(rng):
    d = int(rng.choice([64, 128, 256, 384]))
    n = int(rng.integers(2_000, 90_000))
    X, cent = tp._clustered(rng, n, d, ncl=int(rng.choice([5, 20, 60])), noise=float(rng.choice([0.05, 0.2, 0.5])))
    nq = int(rng.choice([1, 3, 8, 12, 40, 130]))
    topn = min(n, int(rng.choice([17, 64, 65, 100, 200, 300])))
    nb = int(rng.choice([0, 10, 100]))
    Q = cent[rng.integers(0, cent.shape[0], nq)] + 0.1 * rng.standard_normal((nq, d))
    Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    idx = DeviceIndex(d, metric="ip")
    if rng.uniform() < 0.4:
        ids = rng.permutation(10 * n)[:n].astype(np.int64)
        back = {int(v): r for r, v in enumerate(ids)}
        idx.add(X, ids=ids)
        to_rows, to_id = np.vectorize(lambda v: back[int(v)] if v >= 0 else -1), (lambda r: int(ids[r]))
    else:
        base = int(rng.choice([0, 7, 1 << 33]))
        idx.set_id_base(base)
        idx.add(X)
        to_rows, to_id = (lambda I: np.where(I >= 0, I - base, -1)), (lambda r: r + base)
    bm_rows = rng.integers(0, n, (nq, nb)).astype(np.int64)
    if nb:
        bm_rows[:, -3:] = -1
    bm_scores = rng.gamma(2.0, 2.0, size=(nq, nb)) if rng.uniform() < 0.9 else np.zeros((nq, nb))
    kw = {}
    if rng.uniform() < 0.3:
        kw["row_mask"] = (rng.uniform(size=n) < 0.6).astype(np.uint8)
    if rng.uniform() < 0.3:
        kw["min_score"] = 0.3
    tp._check_batch(idx, X, Q, topn, bm_rows, bm_scores, to_rows, to_id, max_results=int(rng.choice([1, 10, 25])),
                    lam=float(rng.choice([0.0, 0.5, 0.7, 1.0])), thr=float(rng.choice([0.0, 0.5, 0.8])),
                    metric=str(rng.choice(["cosine", "dot_product", "euclidean"])), **kw)
    idx.close()


def ivf_pipeline_case(rng):
    d = int(rng.choice([64, 128, 256]))
    n = int(rng.integers(3_000, 40_000))
    nlist = int(rng.choice([8, 32, 100]))
    X, cent = tp._clustered(rng, n, d, ncl=int(rng.choice([10, 30])))
    q = cent[int(rng.integers(0, cent.shape[0]))]
    q = (q / np.linalg.norm(q)).astype(np.float32)
    C = X[rng.choice(n, nlist, replace=False)].copy()
    iv = DeviceIndex(d, metric="ip", kind="ivf")
    iv.train_ivf(nlist, centroids=C)
    id_base = int(rng.choice([0, 500]))
    iv.set_id_base(id_base)
    cut = int(rng.integers(1, n))
    iv.add(X[:cut])
    if rng.uniform() < 0.5:
        iv.search(q, 1, nprobe=1)                                      # packs the first part
    iv.add(X[cut:])
    nb = int(rng.choice([0, 20, 80]))
    tp._check(iv, X, q, int(rng.choice([10, 64, 150])), rng.integers(0, n, nb), rng.gamma(2.0, 2.0, nb), id_base=id_base,
              nprobe=int(rng.integers(1, nlist + 1)), lam=float(rng.choice([0.3, 0.5, 0.7])), thr=float(rng.choice([0.0, 0.6, 0.8])))
    iv.close()


def multi_device_case(rng):
    from rag_amd import MultiDeviceIndex
    shards = int(rng.choice([2, 3, 5, 8]))
    d = int(rng.choice([36, 100, 128, 384, 768]))
    n = int(rng.integers(shards * 4, 40_000))
    nq = int(rng.choice([1, 3, 7, 40, 300]))
    k = int(rng.choice([1, 10, 64, 100]))
    X = tf._unit(rng, n, d)
    if n > 100:
        X[n // 2] = X[7]                                                # a tie across shards
    Q = tf._unit(rng, nq, d)
    multi = MultiDeviceIndex(d, [0] * shards, metric="ip")
    single = DeviceIndex(d, metric="ip")
    cut = int(rng.integers(1, n))
    for lo, hi in ((0, cut), (cut, n)):
        multi.add(X[lo:hi])
        single.add(X[lo:hi])
    mask = (rng.uniform(size=n) < 0.5).astype(np.uint8) if rng.uniform() < 0.5 else None
    min_score = float(rng.choice([-1e30, 0.0, 0.05]))
    D, I = multi.search(Q, k, min_score=min_score, row_mask=mask)
    Ds, Is = single.search(Q, k, min_score=min_score, row_mask=mask)
    # a shard's row count picks its own scoring variant, so a score may differ from the one-index
    # score in the last bit (another summation order): both sides are held to the oracle's tolerance
    bad = flat.compare_topk(D, I, Ds, Is)
    assert bad == [], (dict(shards=shards, n=n, d=d, nq=nq, k=k), "against the one-index answer", bad[:3])
    bad = flat.compare_topk(D, I, *flat.flat_ip_search(X, Q, k, min_score=min_score, row_mask=mask))
    assert bad == [], (dict(shards=shards, n=n, d=d, nq=nq, k=k), bad[:3])
    multi.close()
    single.close()


def keyword_case(seed):
    rng = random.Random(seed)
    host = CorpusStatsManager()
    n = rng.choice([1, 2, 5, 40, 300, 2000])
    texts = tb._texts(rng, n, 1, rng.choice([3, 30, 120]))      # a corpus of empty texts only is an error by design
    if n > 3:
        texts[rng.randrange(n)] = ""
    kw = DeviceKeywordIndex()
    kw.add_texts(texts)
    stats = host.compute_corpus_stats(texts)
    assert kw.corpus_stats() == stats
    for _ in range(8):
        query = " ".join(rng.choice(tb.WORDS + ["unseenword", "the", "zz"]) for _ in range(rng.randint(1, 12)))
        terms = host.preprocess_query(query)
        cand = None if rng.random() < 0.3 else [rng.randrange(n) for _ in range(rng.randint(1, min(100, 2 * n)))]
        ctexts = texts if cand is None else [texts[i] for i in cand]
        assert kw.score(terms, cand, stats) == [host.calculate_bm25_score(terms, t, stats) for t in ctexts]
        limit = rng.choice([1, 3, 10, 50, 100])
        min_score = rng.choice([None, 0.0, 0.25, 0.9])
        want = host.keyword_rank(query, ctexts, texts, limit, min_score)
        got = kw.rank(query, cand, limit, min_score, stats)
        assert got == [((i if cand is None else cand[i]), nv, rv) for i, nv, rv in want], (query, cand, limit, min_score)
    kw.close()


bad = 0
t0 = time.time()
for seed in span("SEEDS", "0:40"):
    for name, fn in (("segments+fanout", lambda: segments_case(np.random.default_rng(5000 + seed))),
                     ("pipeline", lambda: [pipeline_case(np.random.default_rng(6000 + 10 * seed + j)) for j in range(3)]),
                     ("pipeline batch", lambda: [pipeline_batch_case(np.random.default_rng(6500 + 10 * seed + j)) for j in range(2)]),
                     ("ivf pipeline", lambda: ivf_pipeline_case(np.random.default_rng(8000 + seed))),
                     ("several devices", lambda: multi_device_case(np.random.default_rng(9000 + seed))),
                     ("keyword", lambda: keyword_case(7000 + seed))):
        try:
            fn()
        except Exception as e:                                       # a case that raises is a failed case, the run goes on
            bad += 1
            print(f"{name} seed {seed} FAILED: {type(e).__name__}: {str(e)[:600]}", flush=True)
    print(f"seed {seed} done ({time.time() - t0:.0f}s)", flush=True)
print("SOAK", "FAILED" if bad else "ok", bad, flush=True)
sys.exit(1 if bad else 0)
