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


def _strip(hits):
    return [(h["metadata"]["chunk_id"], h["metadata"].get("document_id"), h["content"], h["source"]) for h in hits]


def test_pooled_manager_answers_like_the_per_collection_manager():
    """Every public search of the storage manager and the engine's two fan-outs
    (vector_db/search_manager.py:879-913, :995-1017) on a pooled manager against the
    manager that keeps one device index per collection."""
    from rag_amd.pooled import PooledView
    from rag_amd.schemas import SearchFilters
    from rag_amd.search_engine import HipSearchEngine
    from rag_amd.storage_manager import VectorStorageManager
    run = asyncio.get_event_loop().run_until_complete
    rng = np.random.default_rng(47)
    d = 384
    plain, pooled = VectorStorageManager(), VectorStorageManager(pooled=True)
    docs = list(range(1, 31))
    for doc in docs:
        n = int(rng.integers(3, 250))
        content, context = _unit(rng, n, d), _unit(rng, n, d)
        ids = [f"doc_{doc}_chunk_{i}" for i in range(n)]
        meta = [{"document_id": doc, "text": f"{doc}/{i}", "user_id": 7 if (doc + i) % 3 else 8, "page": i}
                for i in range(n)]
        for m in (plain, pooled):
            assert run(m.create_index(f"doc_{doc}", d))
            assert run(m.add_vectors(f"doc_{doc}", content.tolist(), context.tolist(), meta, ids)) == ids
    assert all(isinstance(o, PooledView) for o in pooled.search_optimizers.values())
    assert len(pooled._pools) == 2                                  # one pool per vector type
    # replace three points of document 4 and append to document 9 (second row range in the pool)
    rep = _unit(rng, 3, d)
    rids = [f"doc_4_chunk_{i}" for i in range(3)]
    extra = _unit(rng, 5, d)
    eids = [f"doc_9_chunk_x{i}" for i in range(5)]
    for m in (plain, pooled):
        run(m.add_vectors("doc_4", rep.tolist(), rep.tolist(), [{"document_id": 4, "text": "v2", "user_id": 7}] * 3, rids))
        run(m.add_vectors("doc_9", extra.tolist(), extra.tolist(), [{"document_id": 9, "user_id": 7}] * 5, eids))
    Q = _unit(rng, 3, d)

    def same(a, b):
        assert _strip(a) == _strip(b)
        assert np.allclose([h["score"] for h in a], [h["score"] for h in b], rtol=1e-6, atol=1e-7)

    for filt in (None, {"user_id": 7}, {"user_id": 7, "page": {"range": {"gte": 2, "lt": 40}}}):
        for name in ("doc_4", "doc_9", "doc_17", "doc_missing"):
            for vtype in ("content", "context"):
                same(run(pooled.search_vectors(name, Q[0].tolist(), vtype, limit=12, score_threshold=0.02, metadata_filters=filt)),
                     run(plain.search_vectors(name, Q[0].tolist(), vtype, limit=12, score_threshold=0.02, metadata_filters=filt)))
            got = run(pooled.contextual_search(name, Q[1].tolist(), limit=8, metadata_filters=filt))
            want = run(plain.contextual_search(name, Q[1].tolist(), limit=8, metadata_filters=filt))
            same(got, want)
            assert [h["score"] for h in got] == pytest.approx([h["score"] for h in want], rel=1e-6)
    # the many-variants equal the loops, on both managers
    names = [f"doc_{doc}" for doc in docs] + ["doc_missing"]
    for m in (plain, pooled):
        many = run(m.search_vectors_many(names, Q.tolist(), "content", 10, 0.05, {"user_id": 7}))
        ctx = run(m.contextual_search_many(names, Q[2].tolist(), limit=6, score_threshold=0.01))
        for ni, name in enumerate(names):
            for qi in range(3):
                same(many[qi][ni], run(plain.search_vectors(name, Q[qi].tolist(), "content", limit=10, score_threshold=0.05,
                                                            metadata_filters={"user_id": 7})))
            same(ctx[ni], run(plain.contextual_search(name, Q[2].tolist(), limit=6, score_threshold=0.01)))
    # engine fan-outs
    strat = [("full_query", Q[0].tolist(), 0.4), ("concepts", Q[1].tolist(), 0.6), ("expanded", Q[2].tolist(), 0.3)]
    filt = SearchFilters(user_id=7, document_ids=docs + [999], min_score=None)
    a, b = (run(HipSearchEngine(m).semantic_search(strat, filt, 10)) for m in (pooled, plain))
    assert len(a) == 10
    assert [(r.document_id, r.chunk_id, r.metadata["search_strategy"]) for r in a] == \
           [(r.document_id, r.chunk_id, r.metadata["search_strategy"]) for r in b]
    assert np.allclose([r.score for r in a], [r.score for r in b], rtol=1e-6)
    # documents with one row range each: searches, weighting and ranking are one device call, and the
    # answer is that of the list-by-list fusion exactly (documents 4 and 9 above took the latter)
    pool_index = pooled.search_optimizers["doc_1_content"].pool.index
    for ids, lim, ms in (([d_ for d_ in docs if d_ not in (4, 9)] + [999], 10, None), ([3, 1, 2], 25, 0.05), ([5], 7, None)):
        f1 = SearchFilters(user_id=7, document_ids=ids, min_score=ms)
        fast = run(HipSearchEngine(pooled).semantic_search(strat, f1, lim))
        assert pool_index.last_path == "segments-fused" and len(fast) > 0
        pooled.search_vectors_fused = None
        slow = run(HipSearchEngine(pooled).semantic_search(strat, f1, lim))
        del pooled.search_vectors_fused
        assert pool_index.last_path == "segments"
        assert [(r.document_id, r.chunk_id, r.score, r.metadata) for r in fast] == \
               [(r.document_id, r.chunk_id, r.score, r.metadata) for r in slow]
        assert [r.text for r in fast] == [r.text for r in slow]
    a, b = (run(HipSearchEngine(m).contextual_semantic(Q[0].tolist(), filt, 5)) for m in (pooled, plain))
    assert [(r.document_id, r.chunk_id) for r in a] == [(r.document_id, r.chunk_id) for r in b] and len(a) > 50
    # vector fetch, soft delete / restore, delete
    va, vb = (run(m.get_vector_by_chunk_id("doc_4_chunk_1")) for m in (pooled, plain))
    assert va["source"] == vb["source"] and np.array_equal(va["vector"], vb["vector"])
    for m in (plain, pooled):
        assert run(m.soft_delete_index("doc_17"))
    assert run(pooled.search_vectors("doc_17", Q[0].tolist(), limit=5)) == []
    assert run(pooled.search_vectors_many(["doc_17", "doc_4"], Q[:1].tolist(), limit=5))[0][0] == []
    for m in (plain, pooled):
        assert run(m.restore_index("doc_17"))
    same(run(pooled.search_vectors("doc_17", Q[0].tolist(), limit=5)), run(plain.search_vectors("doc_17", Q[0].tolist(), limit=5)))
    assert run(pooled.delete_index("doc_9")) and "doc_9_content" not in pooled.search_optimizers
    assert run(pooled.search_vectors("doc_9", Q[0].tolist(), limit=5)) == []
    assert run(pooled.create_index("doc_9", d)) and run(pooled.search_vectors("doc_9", Q[0].tolist(), limit=5)) == []
    stats = run(pooled.get_index_stats("doc_4"))
    assert stats["collections"]["content"]["points_count"] == run(plain.get_index_stats("doc_4"))["collections"]["content"]["points_count"]
    run(plain.cleanup())
    run(pooled.cleanup())


def test_pool_compaction_releases_dropped_rows_and_keeps_answers():
    from rag_amd.pooled import PooledCollections
    rng = np.random.default_rng(53)
    d = 128
    pool = PooledCollections(d, metric="cosine")
    data = {}
    for c in range(12):
        n = int(rng.integers(20, 200))
        x = _unit(rng, n, d) * rng.uniform(0.5, 2.0)
        ids = [f"c{c}_{i}" for i in range(n)]
        pool.upsert(f"c{c}", x, [{"chunk_id": i, "document_id": c} for i in ids], ids)
        data[f"c{c}"] = ids
    pool.upsert("c3", _unit(rng, 4, d), [{"chunk_id": i, "document_id": 3, "v": 2} for i in data["c3"][:4]], data["c3"][:4])
    pool.upsert("c5", _unit(rng, 6, d), [{"chunk_id": f"c5_x{i}", "document_id": 5} for i in range(6)], [f"c5_x{i}" for i in range(6)])
    pool.set_collection_active("c7", False)
    for name in ("c0", "c9"):
        pool.drop(name)
    names = [f"c{c}" for c in range(12)]
    Q = _unit(rng, 2, d)
    flt = {"document_id": {"range": {"gte": 2}}}

    def answers():
        hits = pool.search_many(names, Q, 8, min_score=-1.0, row_mask=pool.filter_mask(names, flt))
        return [[[(pool.payloads[r]["chunk_id"], pool.payloads[r].get("v"), s) for r, s in per] for per in qh] for qh in hits]

    before, total = answers(), len(pool.payloads)
    dead = pool.dead_rows()
    assert dead == len(data["c0"]) + len(data["c9"]) + 4
    assert pool.compact() == dead and pool.dead_rows() == 0 and len(pool.payloads) == total - dead
    assert len(pool.index) == total - dead and all(len(r) <= 1 for r in pool.ranges.values())
    after = answers()
    for qb, qa in zip(before, after):
        for pb, pa in zip(qb, qa):
            assert [(c, v) for c, v, _ in pb] == [(c, v) for c, v, _ in pa]
            assert np.allclose([s for *_x, s in pb], [s for *_x, s in pa], rtol=1e-6, atol=1e-7)
    assert after[0][7] == [] and after[0][0] == []                      # hidden stays hidden, dropped stays gone
    pool.set_collection_active("c7", True)
    assert len(pool.search_many(["c7"], Q, 8, min_score=-1.0)[0][0]) == 8
    assert pool.compact() == 0
    pool.close()


def test_pooled_manager_compacts_after_deletes():
    from rag_amd.storage_manager import VectorStorageManager
    run = asyncio.get_event_loop().run_until_complete
    rng = np.random.default_rng(59)
    d = 64
    mgr = VectorStorageManager(pooled=True)
    for doc in (1, 2, 3):
        x = _unit(rng, 3000, d)
        ids = [f"doc_{doc}_chunk_{i}" for i in range(3000)]
        assert run(mgr.create_index(f"doc_{doc}", d))
        assert run(mgr.add_vectors(f"doc_{doc}", x.tolist(), x.tolist(), [{"document_id": doc}] * 3000, ids)) == ids
    q = _unit(rng, 1, d)[0].tolist()
    before = run(mgr.search_vectors("doc_3", q, limit=10))
    pool = mgr._pools[("content", d)]
    assert len(pool.index) == 9000
    assert run(mgr.delete_index("doc_1")) and len(pool.index) == 9000       # a third dead: left alone
    assert run(mgr.delete_index("doc_2")) and len(pool.index) == 3000       # two thirds dead: rebuilt
    after = run(mgr.search_vectors("doc_3", q, limit=10))
    assert [h["metadata"]["chunk_id"] for h in after] == [h["metadata"]["chunk_id"] for h in before]
    assert np.allclose([h["score"] for h in after], [h["score"] for h in before], rtol=1e-6, atol=1e-7)
    assert run(mgr.get_vector_by_chunk_id("doc_3_chunk_5"))["source"] == "doc_3_content"
    run(mgr.cleanup())


def test_pool_backup_restore_and_last_duplicate_wins(tmp_path):
    from rag_amd.pooled import PooledCollections
    rng = np.random.default_rng(91)
    d = 96
    pool = PooledCollections(d, metric="cosine")
    names = []
    for c in range(6):
        n = int(rng.integers(30, 120))
        ids = [f"c{c}_{i}" for i in range(n)]
        pool.upsert(f"c{c}", _unit(rng, n, d), [{"chunk_id": i, "document_id": c} for i in ids], ids)
        names.append(f"c{c}")
    # one batch that names the same point twice: the later vector and payload win, the earlier row is hidden
    twice = _unit(rng, 3, d)
    pool.upsert("c2", twice, [{"chunk_id": "dup", "v": 1}, {"chunk_id": "other", "v": 1}, {"chunk_id": "dup", "v": 2}],
                ["dup", "other", "dup"])
    hits = pool.search_many(["c2"], twice[[0, 2]], 5, min_score=-1.0)
    for qi in range(2):
        got = [(pool.payloads[r]["chunk_id"], pool.payloads[r].get("v")) for r, _s in hits[qi][0]]
        assert ("dup", 1) not in got                        # not even for its own vector
    assert got[0] == ("dup", 2)
    assert pool.dead_rows() == 1
    pool.set_collection_active("c4", False)
    pool.drop("c1")
    Q = _unit(rng, 3, d)

    def answers(p):
        out = p.search_many(names, Q, 6, min_score=-1.0)
        return [[[(p.payloads[r]["chunk_id"], s) for r, s in per] for per in qh] for qh in out]

    before = answers(pool)
    path = str(tmp_path / "pool.bin")
    info = pool.backup(path)
    assert info["collections"] == 5
    back = PooledCollections.restore(path)
    assert answers(back) == before
    assert back.count("c2") == pool.count("c2") and back.row_of("c2", "dup") == pool.row_of("c2", "dup")
    back.set_collection_active("c4", True)
    assert len(back.search_many(["c4"], Q, 6, min_score=-1.0)[0][0]) == 6
    assert back.compact() == pool.dead_rows()                    # a restored pool vacuums like the original
    back.close()
    pool.close()


def test_a_payload_with_an_empty_chunk_id_keeps_the_host_fusion():
    """A payload that carries an explicit empty chunk_id: the reference skips such a hit before it cuts to
    `limit` (vector_db/search_manager.py:899-901, :928-932), so a full list still comes back.  The device-side
    cut cannot do that; collections with such a payload take the list fusion, and the answers equal the
    per-collection manager's."""
    from rag_amd.schemas import SearchFilters
    from rag_amd.search_engine import HipSearchEngine
    from rag_amd.storage_manager import VectorStorageManager
    run = asyncio.get_event_loop().run_until_complete
    rng = np.random.default_rng(5)
    d = 64
    plain, pooled = VectorStorageManager(), VectorStorageManager(pooled=True)
    Q = _unit(rng, 3, d)
    for doc in (1, 2):
        n = 40
        content = _unit(rng, n, d)
        content[:6] = Q[0] + 0.01 * rng.standard_normal((6, d)).astype(np.float32)      # the best hits of query 0 ...
        ids = [f"doc_{doc}_chunk_{i}" for i in range(n)]
        meta = [{"document_id": doc, "text": f"{doc}/{i}", "user_id": 7} for i in range(n)]
        for m in (plain, pooled):
            assert run(m.create_index(f"doc_{doc}", d))
            if doc == 1:
                assert run(m.add_vectors(f"doc_{doc}", content.tolist(), content.tolist(), meta, ids)) == ids
                continue
            # document 2's points come in below the manager (a collection filled by another writer, registered with
            # add_collection in the reference): add_vectors itself always stamps a chunk id into the payload
            payloads = [dict(p, chunk_id=cid, index_name="doc_2") for p, cid in zip(meta, ids)]
            payloads[0]["chunk_id"] = ""                                                 # ... one of which has no chunk id
            payloads[3]["chunk_id"] = None
            for vtype in ("content", "context"):
                m.search_optimizers[f"doc_2_{vtype}"].upsert(content, payloads, ids)
    strat = [("full_query", Q[0].tolist(), 0.4), ("concepts", Q[1].tolist(), 0.6), ("expanded", Q[2].tolist(), 0.3)]
    pool = pooled.search_optimizers["doc_1_content"].pool
    assert pool.chunk_ids_distinct("doc_1_content") and not pool.chunk_ids_distinct("doc_2_content")
    for docs, path in (([1], "segments-fused"), ([1, 2], "segments"), ([2], "segments")):
        filt = SearchFilters(user_id=7, document_ids=docs, min_score=None)
        a, b = (run(HipSearchEngine(m).semantic_search(strat, filt, 8)) for m in (pooled, plain))
        assert pool.index.last_path == path
        assert len(a) == 8 and all(r.chunk_id for r in a)
        assert [(r.document_id, r.chunk_id) for r in a] == [(r.document_id, r.chunk_id) for r in b]
    run(plain.cleanup())
    run(pooled.cleanup())
