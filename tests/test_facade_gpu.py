"""The drop-in surface (storage manager / search optimizer / search engine) on the
GPU against the oracle's restatement of the reference's wrapper rules."""
import asyncio

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(coro):
    return asyncio.get_event_loop().run_until_complete(coro)


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.fixture()
def manager():
    from rag_amd.storage_manager import VectorStorageManager
    m = VectorStorageManager()
    yield m
    run(m.cleanup())


def _fill(manager, rng, doc, n, d=384, user=7):
    content, context = _unit(rng, n, d), _unit(rng, n, d)
    ids = [f"doc_{doc}_chunk_{i}" for i in range(n)]
    meta = [{"document_id": doc, "user_id": user, "start_char": 10 * i, "end_char": 10 * i + 9,
             "text": f"text {doc}/{i}", "page": i % 5} for i in range(n)]
    assert run(manager.create_index(f"doc_{doc}", d, user_id=user, document_id=doc))
    assert run(manager.add_vectors(f"doc_{doc}", content.tolist(), context.tolist(), meta, ids)) == ids
    return content, context, ids


def test_search_vectors_rules(manager):
    from oracle import flat
    rng = np.random.default_rng(3)
    content, context, ids = _fill(manager, rng, 1, 500)
    q = _unit(rng, 1, 384)[0]
    hits = run(manager.search_vectors("doc_1", q.tolist(), "content", limit=10))
    want = flat.search_vectors_rules(content, q, 10, None)          # threshold None -> 0.0
    assert [h["metadata"]["chunk_id"] for h in hits] == [ids[i] for i, _ in want]
    assert np.allclose([h["score"] for h in hits], [s for _, s in want], rtol=1e-5, atol=1e-6)
    assert all(h["score"] >= 0.0 for h in hits)
    h0 = hits[0]
    assert h0["content"].startswith("text 1/") and h0["metadata"]["document_id"] == 1
    assert h0["metadata"]["vector_type"] == "content" and h0["metadata"]["index_name"] == "doc_1"
    # context collection, threshold, filters
    hits = run(manager.search_vectors("doc_1", q.tolist(), "context", limit=5, score_threshold=0.05))
    want = flat.search_vectors_rules(context, q, 5, 0.05)
    assert [h["metadata"]["chunk_id"] for h in hits] == [ids[i] for i, _ in want]
    hits = run(manager.search_vectors("doc_1", q.tolist(), "content", limit=10, metadata_filters={"page": 2}))
    mask = np.array([i % 5 == 2 for i in range(500)], dtype=np.uint8)
    want = flat.search_vectors_rules(content, q, 10, None, row_mask=mask)
    assert [h["metadata"]["chunk_id"] for h in hits] == [ids[i] for i, _ in want]
    hits = run(manager.search_vectors("doc_1", q.tolist(), "content", limit=10,
                                      metadata_filters={"user_id": 8}))
    assert hits == []
    hits = run(manager.search_vectors("doc_1", q.tolist(), "content", limit=10,
                                      metadata_filters={"start_char": {"range": {"gte": 100, "lt": 200}}}))
    assert all(100 <= h["metadata"]["start_char"] < 200 for h in hits) and hits
    # unknown index: the manager logs and answers [] (storage_manager.py:509-511)
    assert run(manager.search_vectors("nope", q.tolist(), "content")) == []
    assert run(manager.search_vectors("nope", q.tolist(), "context")) == []


def test_contextual_search_matches_oracle(manager):
    from oracle import flat, fusion as ofusion
    rng = np.random.default_rng(5)
    content, context, ids = _fill(manager, rng, 2, 400)
    q = _unit(rng, 1, 384)[0]
    for limit, thr in ((10, None), (5, 0.03), (50, None)):
        got = run(manager.contextual_search("doc_2", q.tolist(), limit=limit, score_threshold=thr))
        c_hits = [(ids[i], s) for i, s in flat.search_vectors_rules(content, q, 2 * limit, None)]
        x_hits = [(ids[i], s) for i, s in flat.search_vectors_rules(context, q, 2 * limit, None)]
        want = ofusion.contextual_fuse(c_hits, x_hits, limit, 0.7, 0.3, thr)
        assert [g["metadata"]["chunk_id"] for g in got] == [w[0] for w in want]
        assert np.allclose([g["score"] for g in got], [w[1] for w in want], rtol=1e-5, atol=1e-6)
        for g in got:
            assert g["score"] == g["content_score"] * 0.7 + g["context_score"] * 0.3


def test_get_vector_soft_delete_upsert(manager):
    rng = np.random.default_rng(8)
    content, context, ids = _fill(manager, rng, 3, 50)
    v = run(manager.get_vector_by_chunk_id(ids[17]))
    assert v["source"] == "doc_3_content" and v["collection"] == "doc_3_content"
    assert np.allclose(v["vector"], content[17], atol=1e-6)
    assert run(manager.get_vector_by_chunk_id("missing")) is None
    q = content[17]
    assert run(manager.search_vectors("doc_3", q.tolist(), limit=1))[0]["metadata"]["chunk_id"] == ids[17]
    assert run(manager.soft_delete_index("doc_3"))
    assert run(manager.search_vectors("doc_3", q.tolist(), limit=1)) == []
    assert run(manager.restore_index("doc_3"))
    assert run(manager.search_vectors("doc_3", q.tolist(), limit=1))[0]["metadata"]["chunk_id"] == ids[17]
    # upsert of an existing chunk id replaces the point
    new = _unit(rng, 1, 384)
    run(manager.add_vectors("doc_3", new.tolist(), new.tolist(), [{"document_id": 3, "text": "new"}], [ids[17]]))
    hit = run(manager.search_vectors("doc_3", new[0].tolist(), limit=1))[0]
    assert hit["metadata"]["chunk_id"] == ids[17] and hit["content"] == "new"
    hits = run(manager.search_vectors("doc_3", q.tolist(), limit=50))
    assert [h["metadata"]["chunk_id"] for h in hits].count(ids[17]) <= 1
    stats = run(manager.get_index_stats("doc_3"))
    assert stats["collections"]["content"]["vectors_count"] == 50
    assert run(manager.delete_index("doc_3")) and "doc_3_content" not in manager.search_optimizers


def test_health_check_smoke_operation(manager):
    # vector_db/health_checker.py:228-249
    assert run(manager.create_index("health_test", 384))
    ids = run(manager.add_vectors("health_test", [[0.1] * 384, [0.2] * 384], [[0.1] * 384, [0.2] * 384],
                                  [{"test": "health_check_1"}, {"test": "health_check_2"}], ["a", "b"]))
    assert len(ids) == 2
    res = run(manager.search_vectors("health_test", [0.15] * 384, limit=2))
    assert len(res) == 2 and abs(res[0]["score"] - 1.0) < 1e-6


def test_optimizer_batch_search_is_one_device_call(manager):
    from oracle import flat
    from rag_amd.schemas import MetricType, SearchConfig
    from rag_amd.search_optimizer import HipSearchOptimizer
    rng = np.random.default_rng(12)
    n, d = 20_000, 384
    X = _unit(rng, n, d)
    opt = HipSearchOptimizer(SearchConfig(dimension=d, metric=MetricType.COSINE, collection_name="big"))
    opt.upsert(X, [{"chunk_id": f"c{i}", "document_id": 1} for i in range(n)], [f"c{i}" for i in range(n)])
    Q = _unit(rng, 64, d)
    res = run(opt.batch_search(Q.tolist(), limit=10))
    assert opt.index.last_path == "gemm"
    Dr, Ir = flat.flat_ip_search(X, Q, 10, min_score=0.0)
    for qi in range(64):
        assert [c.chunk_id for c, _ in res[qi]] == [f"c{i}" for i in Ir[qi] if i >= 0]
    one = run(opt.search(Q[0].tolist(), limit=10))
    assert [c.chunk_id for c, _ in one] == [c.chunk_id for c, _ in res[0]]
    assert run(opt.get_vector_count()) == n and opt.validate_health()["status"] == "healthy"
    run(opt.delete_collection())


def test_engine_semantic_hybrid_and_mmr(manager):
    from oracle import flat, fusion as ofusion, mmr as ommr
    from rag_amd.schemas import SearchFilters, SearchResult
    from rag_amd.search_engine import HipSearchEngine
    rng = np.random.default_rng(21)
    data = {doc: _fill(manager, rng, doc, 300) for doc in (11, 12)}
    eng = HipSearchEngine(manager)
    qs = _unit(rng, 3, 384)
    strat = [("full_query", qs[0].tolist(), 0.4), ("concepts", qs[1].tolist(), 0.6), ("expanded", qs[2].tolist(), 0.3)]
    filt = SearchFilters(user_id=7, document_ids=[11, 12], min_score=None)
    got = run(eng.semantic_search(strat, filt, 10))
    per_doc = []
    for doc in (11, 12):
        content, _, ids = data[doc]
        per_doc.append((doc, [(name, w, [(ids[i], s) for i, s in flat.search_vectors_rules(content, np.array(v, np.float32), 20, 0.1)])
                              for name, v, w in strat]))
    want = ofusion.multi_embedding_fuse(per_doc, 10)
    assert [f"{r.document_id}_{r.chunk_id}" for r in got] == [w[0] for w in want]
    assert np.allclose([r.score for r in got], [w[1] for w in want], rtol=1e-5, atol=1e-6)
    assert [r.metadata["search_strategy"] for r in got] == [w[2] for w in want]
    # hybrid fusion on SearchResult objects
    sem = run(eng.contextual_semantic(qs[0].tolist(), filt, 10))
    bm = [SearchResult(document_id=r.document_id, chunk_id=r.chunk_id, text=r.text, score=float(s))
          for r, s in zip(sem[::3], rng.uniform(0, 1, len(sem[::3])))]
    bm.append(SearchResult(document_id=11, chunk_id="only_bm25", text="x", score=0.9))
    fused = eng.combine_semantic_and_bm25_contextual(sem, bm, limit=10)
    want = ofusion.hybrid_rank([(f"{r.document_id}_{r.chunk_id}", r.score) for r in sem],
                               [(f"{r.document_id}_{r.chunk_id}", r.score) for r in bm], 10)
    assert [f"{r.document_id}_{r.chunk_id}" for r in fused] == [w[0] for w in want]
    assert [r.score for r in fused] == [w[1] for w in want]
    # MMR over results whose vectors are fetched from the storage manager
    filt2 = SearchFilters(enable_mmr=True, mmr_lambda=0.5, mmr_similarity_threshold=0.1, mmr_max_results=5)
    cand = [SearchResult(document_id=r.document_id, chunk_id=r.chunk_id, text=r.text, score=r.score) for r in sem]
    cand.insert(2, SearchResult(document_id=1, chunk_id="no_vector", text="", score=0.99))
    out = run(eng.apply_mmr_diversification(cand, filt2))
    with_vec = [r for r in cand if r.chunk_id != "no_vector"]
    embs = [np.array(run(manager.get_vector_by_chunk_id(r.chunk_id))["vector"]) for r in with_vec]
    want_idx = ommr.mmr_select(embs, [r.score for r in with_vec], 0.5, 5, 0.1, "cosine")
    assert [r.chunk_id for r in out] == [with_vec[i].chunk_id for i in want_idx]
    assert [r.metadata["mmr_rank"] for r in out] == [1, 2, 3, 4, 5]


def test_backup_restore_roundtrip(tmp_path):
    from rag_amd.schemas import MetricType, SearchConfig
    from rag_amd.search_optimizer import HipSearchOptimizer
    rng = np.random.default_rng(33)
    n, d = 3000, 100                      # d % 4 == 0 but rows padded? 100 -> ld 100; also try 90 below
    for d in (100, 90):
        X = (rng.standard_normal((n, d)) * 2.0).astype(np.float32)
        cfg = SearchConfig(dimension=d, metric=MetricType.COSINE, collection_name="bk")
        opt = HipSearchOptimizer(cfg)
        ids = [f"c{i}" for i in range(n)]
        opt.upsert(X, [{"chunk_id": c, "document_id": 5, "text": c} for c in ids], ids)
        opt.upsert(X[:3] * -1.0, [{"chunk_id": c, "document_id": 5, "text": "v2"} for c in ids[:3]], ids[:3])
        q = X[10] + 0.05
        before = run(opt.search(q.tolist(), limit=15))
        path = str(tmp_path / f"snap{d}.bin")
        info = opt.backup(path)
        assert info["points_count"] == n
        run(opt.delete_collection())
        opt2 = HipSearchOptimizer.restore(path, cfg)
        after = run(opt2.search(q.tolist(), limit=15))
        assert [(c.chunk_id, s) for c, s in after] == [(c.chunk_id, s) for c, s in before]
        assert run(opt2.get_vector_count()) == n
        assert opt2.get_vector("c0")["payload"]["text"] == "v2"
        run(opt2.delete_collection())


def test_duplicate_ids_in_one_batch_and_oversized_limit():
    """The last point of a batch that names an id twice wins (Qdrant upsert semantics,
    vector_db/qdrant_client.py:207-246); a limit beyond the device bound is an error, not a shorter answer."""
    from rag_amd.schemas import MetricType, SearchConfig, SearchError
    from rag_amd.search_optimizer import HipSearchOptimizer
    rng = np.random.default_rng(8)
    d = 64
    opt = HipSearchOptimizer(SearchConfig(dimension=d, metric=MetricType.COSINE, collection_name="dups"))
    base = _unit(rng, 40, d)
    opt.upsert(base, [{"chunk_id": f"b{i}", "document_id": 1, "text": "base"} for i in range(40)], [f"b{i}" for i in range(40)])
    v = _unit(rng, 3, d)
    opt.upsert(v, [{"chunk_id": "x", "document_id": 1, "text": "first"}, {"chunk_id": "y", "document_id": 1, "text": "y"},
                   {"chunk_id": "x", "document_id": 1, "text": "second"}], ["x", "y", "x"])
    assert run(opt.get_vector_count()) == 42
    for q in (v[0], v[2]):
        hits = run(opt.search(q.tolist(), limit=42, min_score=-1.0))
        texts = [c.text for c, _s in hits if c.chunk_id == "x"]
        assert texts == ["second"]
    assert opt.get_vector("x")["payload"]["text"] == "second"
    with pytest.raises(SearchError):
        run(opt.search(v[0].tolist(), limit=2000))
    run(opt.delete_collection())
