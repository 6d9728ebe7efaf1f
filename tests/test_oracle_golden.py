"""The CPU oracle against the outputs of the reference's own Python
(tests/golden/*.json, written by oracle/gen_golden.py).  Floating-point values
are compared exactly (same IEEE double operations in the same order) unless a
case says otherwise."""
import numpy as np
import pytest

from oracle import bm25, flat, fusion, mmr


def test_contextual_fuse_matches_reference(golden):
    data = golden("fusion_contextual.json")
    assert len(data["cases"]) >= 20
    for c in data["cases"]:
        # the reference asks each side for 2*limit hits (storage_manager.py:639,648)
        over = 2 * c["limit"]
        assert [call[1] for call in c["inner_calls"]] == [over, over]
        got = fusion.contextual_fuse([tuple(x) for x in c["content"]][:over],
                                     [tuple(x) for x in c["context"]][:over],
                                     c["limit"], c["content_weight"], c["context_weight"],
                                     c["score_threshold"])
        assert [[g[0], g[1], g[2], g[3]] for g in got] == c["expected"], c["note"]


def test_hybrid_fuse_matches_reference(golden):
    data = golden("fusion_hybrid.json")
    for c in data["cases"]:
        sem = [(f"{d}_{k}", s) for d, k, s in c["semantic"]]
        bm = [(f"{d}_{k}", s) for d, k, s in c["bm25"]]
        got = fusion.hybrid_fuse(sem, bm)
        if c["fallback"]:
            assert got is None, c["note"]
            continue
        assert got is not None, c["note"]
        assert len(got) == c["expected_len"], c["note"]
        for key, comb, sn, bn, hs, hb in got:
            assert c["expected"][key] == [comb, sn, bn, hs, hb], (c["note"], key)


def test_multi_embedding_fuse_matches_reference(golden):
    data = golden("fusion_multi_embedding.json")
    for c in data["cases"]:
        per_doc = []
        for doc in sorted(int(k) for k in c["doc_hits"]):
            strategies = []
            for name, w in c["strategies"]:
                hits = [tuple(h) for h in c["doc_hits"][str(doc)].get(name, [])][: 2 * c["limit"]]
                strategies.append((name, w, hits))
            per_doc.append((doc, strategies))
        got = fusion.multi_embedding_fuse(per_doc, c["limit"])
        assert [[k, s, n] for k, s, n in got] == c["expected"], c["note"]
        # the wrapper rules around the inner search (search_manager.py:887-894)
        for call in c["inner_calls"]:
            assert call[2] == "content" and call[3] == 2 * c["limit"]
            assert call[4] == (c["min_score"] or 0.1)


def test_bm25_matches_reference(golden):
    data = golden("bm25.json")
    for q, exp in data["preprocess"]:
        assert bm25.preprocess_query(q) == exp
    st = bm25.corpus_stats(data["corpus"])
    assert st["total_docs"] == data["stats"]["total_docs"]
    assert st["avg_doc_length"] == data["stats"]["avg_doc_length"]
    assert dict(st["idf"]) == data["stats"]["idf"]
    for c in data["scores"]:
        stats = c["stats"] if c["stats"] is not None else st
        assert bm25.bm25_score(c["terms"], c["text"], stats) == c["expected"], c
    assert data["scores"][0]["expected"] == 2.414285714285714   # survey known answer
    for c in data["normalize"]:
        assert bm25.normalize_scores(c["scores"], c["method"]) == c["expected"], c


def _mmr_embeddings(c):
    if "embeddings" in c:
        return c["embeddings"]
    r = np.random.default_rng(c["seed"])
    if c["gen"] == "random_unit":
        big = r.standard_normal((200, 768))
    else:
        cent = r.standard_normal((12, 768))
        lab = r.integers(0, 12, 200)
        big = cent[lab] + 0.25 * r.standard_normal((200, 768))
    big = (big / np.linalg.norm(big, axis=1, keepdims=True)).astype(np.float32)
    assert float(np.sum(big.astype(np.float64))) == c["checksum"], "numpy RNG stream changed"
    return [row for row in big]


def mmr_cases(golden):
    return golden("mmr.json")["cases"]


def test_mmr_matches_reference(golden):
    cases = mmr_cases(golden)
    assert len(cases) >= 20
    for c in cases:
        emb = _mmr_embeddings(c)
        got = mmr.mmr_select(emb, c["scores"], c["lambda"], c["max_results"], c["threshold"], c["metric"])
        assert got == c["expected"], c["note"]


def test_mmr_fast_equals_loop(golden):
    for c in mmr_cases(golden):
        emb = _mmr_embeddings(c)
        if any(e is None or len(e) == 0 for e in emb):
            continue
        metric = c["metric"] if c["metric"] in ("cosine", "dot_product", "euclidean") else "cosine"
        got = mmr.mmr_select_fast(np.asarray(emb, dtype=np.float64), c["scores"], c["lambda"],
                                  c["max_results"], c["threshold"], metric)
        assert got == c["expected"], c["note"]


def test_similarity_matrix_and_diversity_match_reference(golden):
    """MMRDiversifier.compute_similarity_matrix / analyze_diversity.  Cosine and dot products agree to
    the last bits of an fp64 matrix product; the reference's euclidean values come out of sklearn's
    |x|^2 - 2 x.y + |y|^2 expansion, whose cancellation leaves ~1e-7 in a distance (its value for two
    identical rows is 0.99999992, not 1), so that metric is compared to 1e-6."""
    data = golden("mmr_analysis.json")
    for c in data["similarity"]:
        E = [np.array(r, dtype=c["dtype"]) for r in data["embedding_sets"][c["set"]]]
        S = mmr.similarity_matrix(E, c["metric"])
        want = np.array(c["expected"])
        assert S.shape == want.shape, c["note"]
        if "expected_shape" in c:
            assert list(S.shape) == c["expected_shape"]
        tol = 1e-6 if c["metric"] == "euclidean" else 1e-12
        assert np.allclose(S, want, rtol=0, atol=tol * max(1.0, float(np.abs(want).max()) if want.size else 1.0)), (c["note"], c["metric"])
    for c in data["diversity"]:
        assert mmr.analyze_diversity(c["results"]) == c["expected"], c["note"]


def test_cosine_scan_matches_reference_and_flat_ip(golden):
    data = golden("cosine_scan.json")
    for c in data["cases"]:
        got = flat.embedded_cosine_scan(c["vectors"], c["query"], c["limit"])
        assert [[i, s] for i, s in got] == c["expected"]
        # on unit vectors the inner-product oracle returns the same ids, scores within fp32 rounding
        X = flat.l2_normalize_rows(np.asarray(c["vectors"], dtype=np.float32))
        q = flat.l2_normalize_rows(np.asarray([c["query"]], dtype=np.float32))
        if not np.any(q):
            continue
        k = min(c["limit"], len(c["vectors"]))
        D, I = flat.flat_ip_search(X, q, k)
        ref_I = np.array([[i for i, _ in c["expected"][:k]]])
        ref_D = np.array([[s for _, s in c["expected"][:k]]])
        assert flat.compare_topk(D, I, ref_D, ref_I, rtol=1e-5, atol=1e-6) == []


def test_flat_ip_semantics():
    rng = np.random.default_rng(5)
    X = flat.l2_normalize_rows(rng.standard_normal((5000, 48)))
    Q = flat.l2_normalize_rows(rng.standard_normal((7, 48)))
    D, I = flat.flat_ip_search(X, Q, 10, block=777)
    S = (Q.astype(np.float64) @ X.astype(np.float64).T)
    ref = np.argsort(-S, axis=1, kind="stable")[:, :10]
    assert np.array_equal(I, ref)
    assert np.allclose(D, np.take_along_axis(S, ref, 1), rtol=1e-5, atol=1e-6)
    assert D.dtype == np.float32 and I.dtype == np.int64
    # padding: k > n
    D2, I2 = flat.flat_ip_search(X[:3], Q, 5)
    assert np.all(I2[:, 3:] == -1) and np.all(D2[:, 3:] == -flat.FLT_MAX)
    # ties -> lower id first
    Xt = np.tile(X[:4], (3, 1))
    D3, I3 = flat.flat_ip_search(Xt, Q[:1], 12, block=5)
    for j in range(0, 12, 3):
        assert I3[0, j] < I3[0, j + 1] < I3[0, j + 2]
        assert I3[0, j] % 4 == I3[0, j + 1] % 4 == I3[0, j + 2] % 4
    # mask and min_score
    mask = np.ones(5000, dtype=np.uint8)
    mask[I[0, :3]] = 0
    D4, I4 = flat.flat_ip_search(X, Q[:1], 10, row_mask=mask)
    assert np.array_equal(I4[0, :7], I[0, 3:])
    D5, I5 = flat.flat_ip_search(X, Q[:1], 10, min_score=float(D[0, 4]))
    assert np.array_equal(I5[0, :5], I[0, :5]) and np.all(I5[0, 5:] == -1)
    assert flat.compare_topk(D, I, D, I) == []
    Ibad = I.copy()
    Ibad[0, 0] = 4999 if I[0, 0] != 4999 else 4998
    assert flat.compare_topk(D, Ibad, D, I) != []


def test_flat_l2_semantics():
    """flat_l2_search against sums of squared differences: ascending squared distances, lower id first
    among equal ones, (-1, FLT_MAX) padding, limit, mask and ids.  Parity unpinned (no FAISS / Qdrant here)."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((400, 12)).astype(np.float32)
    Q = rng.standard_normal((4, 12)).astype(np.float32)
    X[300] = X[20]
    Q[0] = X[20]
    ref = ((Q[:, None, :].astype(np.float64) - X[None].astype(np.float64)) ** 2).sum(-1)
    order = np.argsort(ref, axis=1, kind="stable")
    D, I = flat.flat_l2_search(X, Q, 7)
    assert np.array_equal(I, order[:, :7]) and np.allclose(D, np.take_along_axis(ref, order[:, :7], 1), rtol=1e-6, atol=1e-6)
    assert I[0, 0] == 20 and I[0, 1] == 300 and D[0, 0] == 0.0 and D[0, 1] == 0.0
    lim = float(D[1, 3])
    D2, I2 = flat.flat_l2_search(X, Q[1:2], 7, max_dist2=lim)
    assert I2[0, :4].tolist() == I[1, :4].tolist() and np.all(I2[0, 4:] == -1) and np.all(D2[0, 4:] == flat.FLT_MAX)
    mask = np.ones(400, np.uint8)
    mask[order[2, 0]] = 0
    ids = np.arange(400, dtype=np.int64) * 3 + 1
    D3, I3 = flat.flat_l2_search(X, Q[2:3], 3, row_mask=mask, ids=ids)
    assert I3[0].tolist() == (order[2, 1:4] * 3 + 1).tolist()
    D4, I4 = flat.flat_l2_search(X[:2], Q, 5)
    assert np.all(I4[:, 2:] == -1) and np.all(D4[:, 2:] == flat.FLT_MAX)
    assert flat.compare_topk_l2(D, I, D, I) == [] and flat.compare_topk_l2(D, I, D + 1.0, I) != []
