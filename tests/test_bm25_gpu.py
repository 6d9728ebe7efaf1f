"""Keyword side on the device (rag_bm25_*): raw BM25, min-max normalisation, filter and ranking
against the reference's own outputs (tests/golden/bm25.json, produced by
vector_db/search_manager.py:44-263 CorpusStatsManager) and, on random corpora, against the
host restatement those fixtures pin.  Everything is compared bit for bit (fp64)."""
import math
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORDS = ("dense retrieval inner product search keyword matrix cores accelerate quick brown fox lazy dog "
         "embedding vector index shard stripe page posting series hbm mfma wave lane kernel launch "
         "fusion rerank diversity relevance corpus token query document chunk score").split()


def _texts(rng, n, lo=0, hi=40):
    out = []
    for _ in range(n):
        k = rng.randint(lo, hi)
        out.append(" ".join(rng.choice(WORDS) for _ in range(k)))
    return out


def test_golden_scores_on_the_device(golden):
    from rag_amd.bm25 import CorpusStatsManager, DeviceKeywordIndex
    data = golden("bm25.json")
    host = CorpusStatsManager()
    corpus_stats = host.compute_corpus_stats(data["corpus"])     # a case without statistics uses the golden corpus's
    checked = 0
    for case in data["scores"]:
        stats = case["stats"] if case["stats"] is not None else corpus_stats
        kw = DeviceKeywordIndex()
        kw.add_texts(["filler text before", case["text"], "filler text after the document"])
        if float(stats.get("avg_doc_length", 100)) == 0.0 and case["terms"]:
            with pytest.raises(ValueError):
                kw.score(case["terms"], [1], stats)
        else:
            got = kw.score(case["terms"], [1], stats)
            assert got == [case["expected"]], (case, got)
            checked += 1
        kw.close()
    assert checked >= 5
    # the statistics of the golden corpus from the stored token ids
    kw = DeviceKeywordIndex()
    kw.add_texts(data["corpus"])
    st = kw.corpus_stats()
    assert st["total_docs"] == data["stats"]["total_docs"] and st["avg_doc_length"] == data["stats"]["avg_doc_length"]
    assert st["idf"] == data["stats"]["idf"]
    kw.close()


def test_golden_data_carriers_and_host_rules(golden):
    """The host-side rules of the keyword path and the result types, run in the GPU session as well."""
    from rag_amd.bm25 import CorpusStatsManager
    from rag_amd.mmr_diversifier import normalise_relevance
    from rag_amd.schemas import SearchFilters, SearchResult
    data = golden("bm25.json")
    c = CorpusStatsManager()
    for q, exp in data["preprocess"]:
        assert c.preprocess_query(q) == exp
    for case in data["normalize"]:
        assert c.normalize_scores(case["scores"], case["method"]) == case["expected"], case
    st = c.compute_corpus_stats(data["corpus"])
    assert dict(st["idf"]) == data["stats"]["idf"] and st["avg_doc_length"] == data["stats"]["avg_doc_length"]
    for case in data["scores"]:
        stats = case["stats"] if case["stats"] is not None else st
        assert c.calculate_bm25_score(case["terms"], case["text"], stats) == case["expected"]
    assert normalise_relevance([0.5, 2.0, 1.25]) == [0.0, 1.0, 0.5]
    r = SearchResult(document_id=1, chunk_id="c", text="t", score=0.5, metadata={"a": 1})
    assert SearchResult.from_dict(r.to_dict()) == r
    f = SearchFilters()
    assert (f.mmr_lambda, f.mmr_similarity_threshold, f.mmr_similarity_metric) == (0.6, 0.8, "cosine")


def test_scores_and_ranking_equal_the_host_manager_bitwise():
    from rag_amd.bm25 import CorpusStatsManager, DeviceKeywordIndex
    rng = random.Random(7)
    host = CorpusStatsManager()
    for trial in range(12):
        n = rng.choice([1, 2, 3, 17, 100, 700])
        texts = _texts(rng, n, 0, rng.choice([5, 40, 200]))
        if n > 3:
            texts[rng.randrange(n)] = ""            # an empty text scores 0.0
            texts[rng.randrange(n)] = "the a an"    # only stop words: the first three words are kept
        kw = DeviceKeywordIndex()
        kw.add_texts(texts[: n // 2])
        kw.add_texts(texts[n // 2:])                 # appended in two steps
        assert len(kw) == n
        stats = host.compute_corpus_stats(texts)
        assert kw.corpus_stats() == stats
        for _ in range(6):
            query = " ".join(rng.choice(WORDS + ["unseenword", "zzz"]) for _ in range(rng.randint(1, 9)))
            terms = host.preprocess_query(query)
            cand = None if rng.random() < 0.4 else [rng.randrange(n) for _ in range(rng.randint(1, min(100, 2 * n)))]
            ctexts = texts if cand is None else [texts[i] for i in cand]
            raw = kw.score(terms, cand, stats)
            assert raw == host.calculate_bm25_scores(terms, ctexts, stats)
            assert raw == [host.calculate_bm25_score(terms, t, stats) for t in ctexts]
            limit = rng.choice([1, 3, 10, 50])
            min_score = rng.choice([None, 0.0, 0.25, 0.9])
            want = host.keyword_rank(query, ctexts, texts, limit, min_score)
            got = kw.rank(query, cand, limit, min_score, stats)
            want = [((i if cand is None else cand[i]), nv, rv) for i, nv, rv in want]
            assert got == want, (trial, query, cand, limit, min_score)
        kw.close()


def test_single_candidate_equal_scores_and_limits():
    from rag_amd import RagHipError
    from rag_amd.bm25 import CorpusStatsManager, DeviceKeywordIndex
    host = CorpusStatsManager()
    texts = ["dense retrieval search", "dense retrieval search", "quick brown fox", "dense dense dense retrieval"]
    kw = DeviceKeywordIndex()
    kw.add_texts(texts)
    stats = host.compute_corpus_stats(texts)
    # one candidate: the sigmoid of its raw score (search_manager.py:195-197), with libm's exp
    (doc, nv, rv), = kw.rank("dense retrieval", [3], 5, None, stats)
    assert doc == 3 and rv == host.calculate_bm25_score(["dense", "retrieval"], texts[3], stats)
    assert nv == 1.0 / (1.0 + math.exp(-2.0 * rv))
    # equal scores: 1.0 each (0.0 each when the common score is zero), first-come order
    assert kw.rank("dense retrieval", [0, 1], 5, None, stats) == [(0, 1.0, kw.score(["dense", "retrieval"], [0], stats)[0])] * 0 + \
        [(0, 1.0, kw.score(["dense", "retrieval"], [0], stats)[0]), (1, 1.0, kw.score(["dense", "retrieval"], [1], stats)[0])]
    assert kw.rank("zebra", [0, 2], 5, None, stats) == [(0, 0.0, 0.0), (2, 0.0, 0.0)]
    assert kw.rank("zebra", [0, 2], 5, 0.5, stats) == []
    # duplicate query terms count twice (the reference loops over the term list), unknown documents score 0.0
    assert kw.score(["dense", "dense"], [0], stats) == [host.calculate_bm25_score(["dense", "dense"], texts[0], stats)]
    assert kw.score(["dense"], [99, -1], stats) == [0.0, 0.0]
    assert kw.rank("", None, 5) == [] and kw.score([], None, stats) == [0.0] * 4
    with pytest.raises(ValueError):
        kw.score(["w%d" % i for i in range(65)], None, stats)
    with pytest.raises(RagHipError):
        kw.rank("dense", None, 5000, None, stats)
    kw.close()


def test_device_ranking_feeds_the_hybrid_pipeline():
    """rank_device -> hybrid_mmr with every buffer on the GPU == the host lists through the same pipeline."""
    import torch
    from rag_amd import DeviceIndex
    from rag_amd.bm25 import CorpusStatsManager, DeviceKeywordIndex
    from rag_amd.index import hybrid_mmr
    rng = random.Random(3)
    n, d = 3000, 128
    g = np.random.default_rng(3)
    X = g.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    texts = _texts(rng, n, 3, 30)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    kw = DeviceKeywordIndex()
    kw.add_texts(texts)
    host = CorpusStatsManager()
    stats = host.compute_corpus_stats(texts)
    q = X[5] + 0.1 * g.standard_normal(d).astype(np.float32)
    query = "dense retrieval kernel fusion"
    cand = sorted(rng.sample(range(n), 100))
    want = kw.rank(query, cand, 30, None, stats)
    assert want == [(cand[i], nv, rv) for i, nv, rv in host.keyword_rank(query, [texts[i] for i in cand], texts, 30, None)]
    dev = torch.device("cuda", 0)
    rows, norm, raw, cnt = kw.rank_device(query, torch.tensor(cand, dtype=torch.int64, device=dev), 30, None, stats)
    assert int(cnt.item()) == len(want)
    assert rows.cpu().tolist()[: len(want)] == [w[0] for w in want] and set(rows.cpu().tolist()[len(want):]) <= {-1}
    assert norm.cpu().tolist()[: len(want)] == [w[1] for w in want]
    assert raw.cpu().tolist()[: len(want)] == [w[2] for w in want]
    kwargs = dict(fuse_limit=40, max_results=10, semantic_weight=0.65, bm25_weight=0.35, mmr_lambda=0.6, mmr_threshold=0.8,
                  mmr_metric="cosine", want_fused=True)
    host_out = hybrid_mmr(idx, q, 50, np.asarray([w[0] for w in want], np.int64), np.asarray([w[1] for w in want], np.float64), **kwargs)
    dev_out = hybrid_mmr(idx, torch.from_numpy(q).to(dev), 50, rows, norm, **kwargs)
    torch.cuda.synchronize()
    n_out = int(dev_out["n_out"].item())
    assert n_out == len(host_out["ids"]) and n_out > 0
    assert int(dev_out["status"].item()) == host_out["status"] == 0
    assert np.array_equal(dev_out["ids"].cpu().numpy()[:n_out], host_out["ids"])
    assert np.array_equal(dev_out["scores"].cpu().numpy()[:n_out], host_out["scores"])
    n_fused = int(dev_out["n_fused"].item())
    assert n_fused == host_out["n_fused"]
    for key in ("fused_rows", "fused_scores", "fused_sem", "fused_bm"):
        assert np.array_equal(dev_out[key].cpu().numpy()[:n_fused], host_out[key]), key
    assert np.any(host_out["fused_bm"] > 0)          # keyword hits took part in the fusion
    kw.close()
    idx.close()
