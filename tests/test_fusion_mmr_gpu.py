"""GPU fusion / MMR kernels (through the C ABI) against the reference's own
outputs (tests/golden, produced by oracle/gen_golden.py) and the CPU oracle.
Fused scores are compared bit for bit (fp64, same operation order); MMR and
fusion orderings must be identical."""
import asyncio

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_contextual_fuse_golden(golden):
    from rag_amd import fusion
    for c in golden("fusion_contextual.json")["cases"]:
        over = 2 * c["limit"]
        got = fusion.contextual_fuse([tuple(x) for x in c["content"]][:over], [tuple(x) for x in c["context"]][:over],
                                     c["limit"], c["content_weight"], c["context_weight"], c["score_threshold"])
        assert [[g[0], g[1], g[2], g[3]] for g in got] == c["expected"], c["note"]


def test_hybrid_fuse_golden(golden):
    from oracle import fusion as ofusion
    from rag_amd import fusion
    for c in golden("fusion_hybrid.json")["cases"]:
        sem = [(f"{d}_{k}", s) for d, k, s in c["semantic"]]
        bm = [(f"{d}_{k}", s) for d, k, s in c["bm25"]]
        got = fusion.hybrid_fuse(sem, bm, limit=10_000)
        if c["fallback"]:
            assert got is None, c["note"]
            continue
        assert len(got) == c["expected_len"], c["note"]
        for key, comb, sn, bn, hs, hb in got:
            assert c["expected"][key] == [comb, sn, bn, hs, hb], (c["note"], key)
        # ranking = the caller's stable sort + cut (search_manager.py:969-970)
        for limit in (1, 5, 10_000):
            want = ofusion.hybrid_rank(sem, bm, limit)
            have = fusion.hybrid_fuse(sem, bm, limit)
            assert [h[0] for h in have] == [w[0] for w in want], c["note"]


def test_multi_embedding_fuse_golden(golden):
    from rag_amd import fusion
    for c in golden("fusion_multi_embedding.json")["cases"]:
        per_doc = []
        for doc in sorted(int(k) for k in c["doc_hits"]):
            per_doc.append((doc, [(name, w, [tuple(h) for h in c["doc_hits"][str(doc)].get(name, [])][: 2 * c["limit"]])
                                  for name, w in c["strategies"]]))
        got = fusion.multi_embedding_fuse(per_doc, c["limit"])
        assert [[k, s, n] for k, s, n in got] == c["expected"], c["note"]


def test_fuse_edge_cases():
    from oracle import fusion as ofusion
    from rag_amd import fusion
    rng = np.random.default_rng(1)
    # many candidates, heavy ties, negative zero, threshold exactly on a score
    keys = [f"c{i}" for i in range(3000)]
    cs = np.round(rng.uniform(0, 1, 3000), 1)
    xs = np.round(rng.uniform(0, 1, 3000), 1)
    content = list(zip(keys, cs.tolist()))
    context = list(zip(keys[::-1], xs.tolist()))
    want = ofusion.contextual_fuse(content, context, 50, 0.7, 0.3, 0.37)
    got = fusion.contextual_fuse(content, context, 50, 0.7, 0.3, 0.37)
    assert got == want
    assert fusion.contextual_fuse([("a", -0.0)], [("b", 0.0)], 5) == ofusion.contextual_fuse([("a", -0.0)], [("b", 0.0)], 5)
    with pytest.raises(ValueError):
        fusion.fuse_weighted(np.zeros(5000), np.zeros(5000), 1, 1, None, 5)


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
    assert float(np.sum(big.astype(np.float64))) == c["checksum"]
    return [row for row in big]


def test_mmr_golden(golden):
    from rag_amd.mmr_diversifier import MMRDiversifier
    cases = golden("mmr.json")["cases"]
    for c in cases:
        emb = _mmr_embeddings(c)
        results = []
        for i, (e, s) in enumerate(zip(emb, c["scores"])):
            r = {"score": s, "_i": i}
            if e is not None:
                r["embedding"] = [float(v) for v in e]
            results.append(r)
        m = MMRDiversifier(lambda_param=c["lambda"], similarity_threshold=c["threshold"], max_results=None,
                           similarity_metric=c["metric"])
        out = asyncio.get_event_loop().run_until_complete(
            m.diversify(results, lambda_param=c["lambda"], max_results=c["max_results"]))
        assert [r["_i"] for r in out] == c["expected"], c["note"]


def test_mmr_rows_of_index_and_oracle_at_config_size():
    """Config 'top-200 -> MMR(lambda 0.5) -> 10' with clustered candidates so the
    penalty branch fires; embeddings taken from index rows on the device."""
    from oracle import mmr as omrr
    from rag_amd import DeviceIndex
    from rag_amd.mmr_diversifier import mmr_select_device, mmr_select_rows, normalise_relevance
    rng = np.random.default_rng(7)
    n, d = 5000, 768
    cent = rng.standard_normal((40, d))
    X = cent[rng.integers(0, 40, n)] + 0.2 * rng.standard_normal((n, d))
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    q = (cent[3] / np.linalg.norm(cent[3])).astype(np.float32)
    D, I = idx.search(q, 200)
    rel = normalise_relevance([float(s) for s in D[0]])
    for lam, thr, metric in ((0.5, 0.8, "cosine"), (0.3, 0.5, "cosine"), (0.5, 0.0, "dot_product"), (0.7, 0.4, "euclidean")):
        want = omrr.mmr_select_fast(X[I[0]].astype(np.float64), [float(s) for s in D[0]], lam, 10, thr, metric)
        got_rows = mmr_select_rows(idx, I[0], rel, lam, thr, metric, 10)
        got_host = mmr_select_device(X[I[0]], rel, lam, thr, metric, 10)
        assert got_rows == want and got_host == want, (lam, thr, metric)
    idx.close()


def test_similarity_matrix_and_diversity_golden(golden):
    """MMRDiversifier.compute_similarity_matrix on the device against the reference's outputs
    (vector_db/mmr_diversifier.py:291-319) and analyze_diversity (:321-366) on dicts and result objects.
    The device computes in fp64 whatever the input type (as the MMR selection does); the reference's
    result for fp32 rows is an fp32 matrix and its euclidean values carry sklearn's ~1e-7 cancellation
    noise: those are compared to 1e-6, everything else to 1e-12 of the largest entry."""
    from oracle import mmr as ommr
    from rag_amd.mmr_diversifier import MMRDiversifier
    from rag_amd.schemas import SearchResult
    data = golden("mmr_analysis.json")
    for c in data["similarity"]:
        E = [np.array(r, dtype=c["dtype"]) for r in data["embedding_sets"][c["set"]]]
        m = MMRDiversifier(similarity_metric=c["metric"])
        S = m.compute_similarity_matrix(E)
        want = np.array(c["expected"])
        assert S.shape == want.shape and S.dtype == np.float64, (c["note"], S.shape, want.shape)
        if not want.size:
            continue
        loose = c["metric"] == "euclidean" or c["dtype"] == "float32"
        tol = (1e-6 if loose else 1e-12) * max(1.0, float(np.abs(want).max()))
        assert np.abs(S - want).max() <= tol, (c["note"], c["metric"], float(np.abs(S - want).max()))
        assert np.array_equal(S, S.T)                                   # symmetric bit for bit
        if c["metric"] == "euclidean":
            assert np.all(np.diag(S) == 1.0)                            # a row is at distance 0 from itself
    for c in data["diversity"]:
        given = [SearchResult(**r) for r in c["results"]] if c["as_objects"] else c["results"]
        got = MMRDiversifier().analyze_diversity(given)
        assert got == c["expected"], (c["note"], got)
    # sizes past one tile, rows wider than the register path, fp64 oracle on the same rows
    rng = np.random.default_rng(3)
    for n, d in ((1, 5), (9, 64), (130, 384), (257, 768), (40, 1100), (600, 33)):
        E = rng.standard_normal((n, d))
        E[n // 2] = E[0]
        for metric in ("cosine", "dot_product", "euclidean"):
            S = MMRDiversifier(similarity_metric=metric).compute_similarity_matrix(list(E))
            want = ommr.similarity_matrix(list(E), metric)
            # the expansion's cancellation error in a distance grows with the rows' norms (~sqrt(eps) * |x|)
            tol = 1e-7 * float(np.linalg.norm(E, axis=1).max()) if metric == "euclidean" else 1e-12 * max(1.0, float(np.abs(want).max()))
            assert S.shape == (n, n) and np.abs(S - want).max() <= tol, (n, d, metric, float(np.abs(S - want).max()))
            assert np.array_equal(S, S.T)
            if metric == "euclidean":                                   # against sums of squared differences: no cancellation
                direct = np.stack([1.0 / (1.0 + np.sqrt(((E - E[i]) ** 2).sum(axis=1))) for i in range(n)])
                assert np.abs(S - direct).max() <= 1e-12, (n, d, float(np.abs(S - direct).max()))


def test_rerank_blend_matches_reference_formula():
    """(1-w)*clip(orig) + w*clip(rerank), threshold on the raw rerank score, stable sort, top_k:
    bit-exact against the oracle's restatement of base_reranker.py:256-280 and the ordering of
    huggingface_reranker.py:225-264."""
    from oracle import fusion as of
    from rag_amd import fusion
    rng = np.random.default_rng(77)
    for n, w, thr, top_k in ((50, 0.5, None, None), (200, 0.3, 0.1, 10), (7, 0.9, 2.0, 3), (64, 0.0, None, 5)):
        orig = rng.uniform(-0.2, 1.3, n)
        rr = rng.uniform(-0.5, 1.5, n)
        rr[::7] = rr[0]                       # ties on the rerank side
        orig[::7] = orig[0]
        order, blend = fusion.rerank_blend(orig, rr, w, thr, top_k)
        kept = [i for i in range(n) if thr is None or not (rr[i] < thr)]
        want = {i: of.rerank_blend(float(orig[i]), float(rr[i]), w) for i in kept}
        ref = sorted(kept, key=lambda i: want[i], reverse=True)        # stable, like list.sort(reverse=True)
        if top_k:
            ref = ref[:top_k]
        assert order == ref
        for i in kept:
            assert blend[i] == want[i]
