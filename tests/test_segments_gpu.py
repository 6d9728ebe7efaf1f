"""Segment-restricted search (one pass for a whole per-document fan-out) must give,
per (query, collection), exactly what a search of that collection alone gives."""
import asyncio

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def test_search_segments_matches_oracle_per_segment():
    from oracle import flat
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(41)
    for d, k in ((384, 20), (768, 10), (100, 5), (384, 100)):
        n = 6000
        X = _unit(rng, n, d)
        Q = _unit(rng, 3, d)
        cuts = np.sort(rng.choice(np.arange(1, n), 40, replace=False))
        segs = [(int(a), int(b)) for a, b in zip(np.r_[0, cuts], np.r_[cuts, n])]
        segs += [(100, 100), (0, n), (5999, 6000)]              # empty, whole corpus, one row
        idx = DeviceIndex(d, metric="ip")
        idx.add(X)
        mask = (rng.uniform(size=n) < 0.7).astype(np.uint8)
        for m in (None, mask):
            D, I = idx.search_segments(Q, segs, k, min_score=0.0, row_mask=m)
            assert idx.last_path == "segments" and D.shape == (3, len(segs), k)
            for si, (b, e) in enumerate(segs):
                sub_mask = None if m is None else m[b:e]
                Dr, Ir = flat.flat_ip_search(X[b:e], Q, k, min_score=0.0, row_mask=sub_mask)
                Ir = np.where(Ir >= 0, Ir + b, -1)
                bad = flat.compare_topk(D[:, si], I[:, si], Dr, Ir)
                assert bad == [], (d, k, si, bad[:3])
        idx.close()


def test_pooled_collections_equal_per_index_loop():
    from rag_amd.pooled import PooledCollections
    from rag_amd.storage_manager import VectorStorageManager
    rng = np.random.default_rng(43)
    d = 384
    mgr = VectorStorageManager()
    pool = PooledCollections(d, metric="cosine")
    docs = {}
    run = asyncio.get_event_loop().run_until_complete
    for doc in range(1, 26):
        n = int(rng.integers(5, 300))
        x = _unit(rng, n, d)
        ids = [f"doc_{doc}_chunk_{i}" for i in range(n)]
        meta = [{"document_id": doc, "text": f"{doc}/{i}"} for i in range(n)]
        run(mgr.create_index(f"doc_{doc}", d))
        run(mgr.add_vectors(f"doc_{doc}", x.tolist(), x.tolist(), meta, ids))
        pool.upsert(f"doc_{doc}", x, [dict(m, chunk_id=c) for m, c in zip(meta, ids)], ids)
        docs[doc] = ids
    extra = _unit(rng, 7, d)                                     # a later add to document 3: second range
    eids = [f"doc_3_chunk_x{i}" for i in range(7)]
    run(mgr.add_vectors("doc_3", extra.tolist(), extra.tolist(), [{"document_id": 3}] * 7, eids))
    pool.upsert("doc_3", extra, [{"document_id": 3, "chunk_id": c} for c in eids], eids)
    Q = _unit(rng, 3, d)
    names = [f"doc_{doc}" for doc in docs] + ["doc_unknown"]
    limit = 20
    hits = pool.search_many(names, Q, limit, min_score=0.1)
    for qi in range(3):
        for ci, name in enumerate(names):
            want = run(mgr.search_vectors(name, Q[qi].tolist(), "content", limit=limit, score_threshold=0.1))
            got = hits[qi][ci]
            assert [pool.payloads[r]["chunk_id"] for r, _ in got] == [w["metadata"]["chunk_id"] for w in want], (qi, name)
            assert np.allclose([s for _, s in got], [w["score"] for w in want], rtol=1e-5, atol=1e-6)
    assert len(pool.ranges["doc_3"]) == 2
    run(mgr.cleanup())
    pool.close()
