"""CPU-only checks: the C ABI library loads and exports every declared symbol,
host-side logic (BM25, relevance normalisation, sharding) matches the golden
vectors, and the product path refuses to run without a GPU instead of falling
back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rag_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rag_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    from rag_amd import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rag_version() == 1


def test_no_oracle_import_in_product_code():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rag_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rag_amd import DeviceIndex, RagHipError
    with pytest.raises(RagHipError):
        DeviceIndex(16, metric="ip")
    from rag_amd import fusion
    with pytest.raises(RagHipError):
        fusion.fuse_weighted([0.5], [0.25], 0.7, 0.3, None, 1)
    from rag_amd.mmr_diversifier import MMRDiversifier
    with pytest.raises(RagHipError):
        MMRDiversifier().diversify_sync([{"score": 1.0, "embedding": [1.0, 0.0]}, {"score": 0.5, "embedding": [0.0, 1.0]}])


def test_bm25_host_logic_matches_reference(golden):
    from rag_amd.bm25 import CorpusStatsManager
    data = golden("bm25.json")
    c = CorpusStatsManager()
    for q, exp in data["preprocess"]:
        assert c.preprocess_query(q) == exp
    st = c.compute_corpus_stats(data["corpus"])
    assert st["total_docs"] == data["stats"]["total_docs"]
    assert st["avg_doc_length"] == data["stats"]["avg_doc_length"]
    assert dict(st["idf"]) == data["stats"]["idf"]
    for case in data["scores"]:
        stats = case["stats"] if case["stats"] is not None else st
        assert c.calculate_bm25_score(case["terms"], case["text"], stats) == case["expected"]
    for case in data["normalize"]:
        assert c.normalize_scores(case["scores"], case["method"]) == case["expected"]
    from oracle import bm25 as obm
    texts = data["corpus"]
    assert c.keyword_rank("dense retrieval search", texts, texts, 3) == obm.keyword_rank("dense retrieval search", texts, texts, 3)


def test_relevance_normalisation_and_schemas():
    from rag_amd.mmr_diversifier import MMRDiversifier, create_balanced_mmr, normalise_relevance
    from rag_amd.schemas import SearchFilters, SearchResult
    assert normalise_relevance([0.5, 2.0, 1.25]) == [0.0, 1.0, 0.5]
    assert normalise_relevance([3.0, 3.0]) == [1.0, 1.0] and normalise_relevance([]) == []
    with pytest.raises(ValueError):
        MMRDiversifier(lambda_param=1.5)
    assert create_balanced_mmr().lambda_param == 0.6
    f = SearchFilters()
    assert (f.mmr_lambda, f.mmr_similarity_threshold, f.mmr_similarity_metric) == (0.6, 0.8, "cosine")
    with pytest.raises(ValueError):
        SearchFilters(mmr_similarity_metric="manhattan")
    r = SearchResult(document_id=1, chunk_id="c", text="t", score=0.5, metadata={"a": 1})
    assert SearchResult.from_dict(r.to_dict()) == r
    # no-embedding path answers the first results without touching the GPU (mmr_diversifier.py:170-172)
    assert MMRDiversifier().diversify_sync([{"score": 0.1}, {"score": 0.9}], max_results=1) == [{"score": 0.1}]


def test_shard_range_covers_rows_once():
    from rag_amd.sharded import shard_range
    for n, w in ((10_000_000, 8), (100, 3), (7, 8), (0, 2), (12_345_678, 4)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and a <= b


def test_bm25_batch_scores_equal_scalar_scores_bitwise():
    """The array form of calculate_bm25_score (one pass over the candidate texts) gives the
    same float64 bits as the per-document loop of the reference (search_manager.py:142-180)."""
    import random
    from rag_amd.bm25 import CorpusStatsManager
    rnd = random.Random(5)
    vocab = ["alpha", "beta", "gamma", "delta", "vector", "search", "index", "kernel", "the", "and", "gpu", "row"]
    texts = [" ".join(rnd.choice(vocab) for _ in range(rnd.randint(0, 60))) for _ in range(120)]
    texts[3] = ""
    texts[4] = "   "
    texts[5] = None
    m = CorpusStatsManager()
    stats = m.compute_corpus_stats([t for t in texts if t])
    for query in ("alpha vector search", "gpu gpu kernel unknownterm", "the and", "row"):
        terms = m.preprocess_query(query)
        batch = m.calculate_bm25_scores(terms, texts, stats)
        scalar = [m.calculate_bm25_score(terms, t or "", stats) for t in texts]
        assert batch == scalar


def test_payload_filters_follow_the_reference_conditions():
    """vector_db/search_optimizer.py:173-201: all conditions must hold; a list value is one
    equality condition per element; {'range': ...} is numeric; a list payload matches by membership."""
    from rag_amd.filters import payload_passes
    p = {"user_id": 7, "tags": ["a", "b"], "page": 12, "flag": True, "name": "x"}
    assert payload_passes(p, {"user_id": 7})
    assert not payload_passes(p, {"user_id": 8})
    assert not payload_passes(p, {"missing": 1})
    assert payload_passes(p, {"tags": "a"}) and not payload_passes(p, {"tags": "c"})
    assert payload_passes(p, {"tags": ["a", "b"]}) and not payload_passes(p, {"tags": ["a", "c"]})
    assert payload_passes(p, {"page": {"range": {"gte": 12, "lte": 12}}})
    assert not payload_passes(p, {"page": {"range": {"gt": 12}}})
    assert not payload_passes(p, {"page": {"range": {"lt": 12}}})
    assert not payload_passes(p, {"flag": {"range": {"gte": 0}}})          # booleans are not numbers here
    assert not payload_passes(p, {"name": {"range": {"gte": 0}}})
    assert payload_passes(p, {"user_id": 7, "page": {"range": {"gte": 1}}, "tags": "b"})
    assert not payload_passes(p, {"user_id": 7, "page": {"range": {"gte": 13}}})
