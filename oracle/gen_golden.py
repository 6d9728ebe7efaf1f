#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the reference's own Python.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs
/root/reference); the fixtures it writes are data (inputs + the reference's
outputs), never the reference's source.  Usage:

    python oracle/gen_golden.py            # rewrites tests/golden/
    python oracle/gen_golden.py mmr_analysis.json      # only the named files

How the reference is loaded: its packages cannot be imported whole
(``api/__init__.py`` and ``database/__init__.py`` pull in FastAPI routes and
``pydantic_settings``, which is not installed), so single files are loaded with
``importlib.util.spec_from_file_location`` after registering empty namespace
packages and ``MagicMock`` stand-ins for the third-party / unrelated modules they
import (qdrant_client, database.models, config, redis manager, reranker, ...).
Only pure functions of those files are exercised.
"""
from __future__ import annotations

import asyncio
import importlib.util
import json
import os
import sys
import tempfile
import time
import types
from unittest.mock import MagicMock

import numpy as np

REF = os.environ.get("RAG_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _ns(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    for pkg in ("api", "api.schemas", "vector_db", "database", "utils", "utils.caching",
                "utils.search"):
        _ns(pkg)
    for stub in ("qdrant_client", "qdrant_client.models", "qdrant_client.http",
                 "qdrant_client.http.exceptions", "database.models", "database.connection",
                 "vector_db.table_search_enhancer", "vector_db.reranker",
                 "vector_db.reranker.base_reranker", "utils.caching.redis_manager",
                 "fakeredis", "utils.search.query_processor"):
        sys.modules[stub] = MagicMock(name=stub)
    cfg = types.ModuleType("config")
    cfg.get_settings = lambda: MagicMock(name="settings")
    sys.modules["config"] = cfg
    mods = {}
    mods["schemas"] = _load("api.schemas.search_schemas", "api/schemas/search_schemas.py")
    mods["chunking"] = _load("vector_db.chunking", "vector_db/chunking.py")
    mods["ctx"] = _load("vector_db.search_context_processor", "vector_db/search_context_processor.py")
    mods["mmr"] = _load("vector_db.mmr_diversifier", "vector_db/mmr_diversifier.py")
    mods["opt"] = _load("vector_db.search_optimizer", "vector_db/search_optimizer.py")
    mods["fb"] = _load("utils.service_fallbacks", "utils/service_fallbacks.py")
    mods["qc"] = _load("vector_db.qdrant_client", "vector_db/qdrant_client.py")
    mods["sm"] = _load("vector_db.storage_manager", "vector_db/storage_manager.py")
    mods["se"] = _load("vector_db.search_manager", "vector_db/search_manager.py")
    return mods


def run(coro):
    return asyncio.get_event_loop().run_until_complete(coro)


# --------------------------------------------------------------------------- fusion
def gen_contextual(mods, rng):
    VSM = mods["sm"].VectorStorageManager

    class Fake(VSM):
        def __init__(self, content, context):
            super().__init__()
            self._c, self._x = content, context
            self.calls = []

        async def search_vectors(self, index_name, query_vector, vector_type="content", limit=10,
                                 score_threshold=None, metadata_filters=None, use_faiss=None):
            self.calls.append((vector_type, limit, score_threshold))
            src = self._c if vector_type == "content" else self._x
            return [{"score": s, "content": "", "metadata": {"chunk_id": cid}, "source": "qdrant"}
                    for cid, s in src][:limit]

    cases = []

    def add(content, context, limit, cw=0.7, xw=0.3, thr=None, note=""):
        f = Fake(content, context)
        res = run(f.contextual_search("idx", [0.0], limit=limit, content_weight=cw,
                                      context_weight=xw, score_threshold=thr))
        cases.append({
            "note": note, "content": content, "context": context, "limit": limit,
            "content_weight": cw, "context_weight": xw, "score_threshold": thr,
            "inner_calls": [list(c) for c in f.calls],
            "expected": [[r["metadata"]["chunk_id"], r["score"], r["content_score"], r["context_score"]]
                         for r in res],
        })

    add([["a", 0.9], ["b", 0.5]], [["b", 0.8], ["c", 0.4]], 10, note="survey known answer")
    add([], [], 5, note="both empty")
    add([["a", 0.5]], [], 5, note="context empty")
    add([], [["z", 0.25]], 5, note="content empty")
    add([["a", 0.5], ["b", 0.5], ["c", 0.5]], [["c", 0.5], ["d", 0.5]], 3, note="exact ties keep insertion order")
    add([["a", 0.9], ["b", 0.2]], [["b", 0.1], ["c", 0.9]], 10, thr=0.27, note="threshold drops")
    add([["a", 0.9], ["", 0.95], ["b", 0.2]], [["b", 0.1], [None, 0.9]], 10, note="missing chunk ids skipped")
    add([["a", 0.9], ["a", 0.1]], [["a", 0.3], ["a", 0.6]], 10, note="repeated id keeps later score")
    add([["a", 0.9], ["b", 0.2]], [["b", 0.1], ["c", 0.9]], 10, thr=0.0, note="threshold 0.0 is not None")
    for t in range(12):
        nc = int(rng.integers(1, 40))
        nx = int(rng.integers(0, 40))
        limit = int(rng.integers(1, 25))
        pool = [f"doc_{int(rng.integers(1, 4))}_chunk_{i}" for i in range(50)]
        cids = list(rng.choice(pool, size=nc, replace=False))
        xids = list(rng.choice(pool, size=nx, replace=False))
        cs = np.sort(rng.uniform(0.0, 1.0, nc))[::-1]
        xs = np.sort(rng.uniform(0.0, 1.0, nx))[::-1]
        # the reference over-fetches 2*limit from each side (storage_manager.py:639,648)
        content = [[c, float(s)] for c, s in zip(cids, cs)]
        context = [[c, float(s)] for c, s in zip(xids, xs)]
        thr = None if t % 3 else float(rng.uniform(0.1, 0.5))
        cw, xw = (0.7, 0.3) if t % 2 else (float(rng.uniform(0, 1)), float(rng.uniform(0, 1)))
        add(content, context, limit, cw, xw, thr, note=f"random {t}")
    return {"source": "vector_db/storage_manager.py:604-716 VectorStorageManager.contextual_search",
            "cases": cases}


def gen_hybrid(mods, rng):
    SE = mods["se"].EnhancedSearchEngine
    SR = mods["schemas"].SearchResult
    cases = []

    def mk(rows):
        return [SR(document_id=d, chunk_id=c, text="t", score=s, metadata={}) for d, c, s in rows]

    def add(sem, bm, note=""):
        sem_objs = mk(sem)
        out = SE._combine_semantic_and_bm25_contextual(None, sem_objs, mk(bm))
        fell_back = out is sem_objs
        # fallback returns the semantic list object unchanged (search_manager.py:1098-1101)
        exp = {f"{r.document_id}_{r.chunk_id}": [r.score, r.metadata.get("semantic_score"),
                                                r.metadata.get("bm25_score"),
                                                r.metadata.get("has_semantic"),
                                                r.metadata.get("has_bm25")] for r in out}
        cases.append({"note": note, "semantic": sem, "bm25": bm, "fallback": bool(fell_back),
                      "expected": exp, "expected_len": len(out)})

    add([[1, "a", 0.9], [1, "b", 0.6]], [[1, "b", 1.0], [1, "c", 0.5]], "survey known answer")
    add([], [], "both empty")
    add([[1, "a", 0.4]], [], "bm25 empty divides by 1.0")
    add([], [[2, "z", 0.7]], "semantic empty divides by 1.0")
    add([[1, "a", 0.0], [1, "b", 0.0]], [[1, "b", 1.0]], "zero semantic max -> fallback")
    add([[1, "a", 0.5]], [[1, "b", 0.0], [1, "c", 0.0]], "zero bm25 max -> fallback")
    add([[1, "a", -0.5], [1, "b", -0.25]], [[1, "b", 1.0]], "negative semantic max flips sign")
    add([[1, "a", 0.9], [1, "a", 0.3]], [[1, "a", 0.2], [1, "a", 0.8]], "repeated key keeps later entry, max over all")
    add([[1, "1_a", 0.9]], [[11, "a", 0.5]], "keys are f'{doc}_{chunk}' strings")
    for t in range(10):
        ns, nb = int(rng.integers(1, 60)), int(rng.integers(1, 40))
        pool = [(int(rng.integers(1, 5)), f"chunk_{i}") for i in range(80)]
        si = rng.choice(len(pool), size=ns, replace=False)
        bi = rng.choice(len(pool), size=nb, replace=False)
        sem = [[pool[i][0], pool[i][1], float(rng.uniform(0.05, 1.0))] for i in si]
        bm = [[pool[i][0], pool[i][1], float(rng.uniform(0.0, 1.0))] for i in bi]
        add(sem, bm, f"random {t}")
    return {"source": "vector_db/search_manager.py:1023-1101 _combine_semantic_and_bm25_contextual",
            "cases": cases}


def gen_multi_embedding(mods, rng):
    """Drive the real ``_semantic_search`` (search_manager.py:834-936) with canned
    hits; embedding manager, query processor and the SQL lookups are mocks."""
    se_mod = mods["se"]
    SE = se_mod.EnhancedSearchEngine
    SR = mods["schemas"].SearchResult
    SF = mods["schemas"].SearchFilters
    cases = []

    def add(doc_hits, strategies, limit, min_score=None, note=""):
        # doc_hits: {doc_id: {strategy_name: [[chunk_id, score], ...]}}
        names = [s[0] for s in strategies]
        eng = SE.__new__(SE)
        calls = []

        class Emb:
            def __init__(self):
                self.i = 0

            async def generate_embeddings(self, texts):
                v = [float(self.i)]
                self.i += 1
                return [v]

        class Store:
            async def search_vectors(self, index_name, query_vector, vector_type, limit,
                                     score_threshold, metadata_filters):
                calls.append([index_name, int(query_vector[0]), vector_type, limit, score_threshold])
                doc = int(index_name.split("_")[1])
                name = names[int(query_vector[0])]
                return [{"score": s, "metadata": {"chunk_id": c, "document_id": doc}}
                        for c, s in doc_hits.get(doc, {}).get(name, [])][:limit]

        async def create(self_, vr, db):
            md = vr["metadata"]
            return SR(document_id=md["document_id"], chunk_id=md["chunk_id"], text="t",
                      score=vr["score"], metadata={})

        eng.storage_manager = Store()
        eng.embedding_manager = Emb()
        eng._create_search_result = types.MethodType(create, eng)
        pq = types.SimpleNamespace(search_strategy="s",
                                   primary_terms=["x"] if "concepts" in names else [],
                                   secondary_terms=[],
                                   confidence=0.5 if "expanded" in names else 0.95)
        qp = MagicMock()
        qp.process_query.return_value = pq
        qp.expand_query.return_value = types.SimpleNamespace(primary_terms=["expanded", "text"])
        sys.modules["utils.search.query_processor"].get_query_processor = lambda: qp
        filt = SF(document_ids=sorted(doc_hits.keys()), user_id=7, min_score=min_score)
        res = run(eng._semantic_search("the query", filt, limit, None))
        cases.append({
            "note": note, "limit": limit, "min_score": min_score,
            "strategies": strategies,
            "doc_hits": {str(k): v for k, v in doc_hits.items()},
            "inner_calls": calls,
            "expected": [[f"{r.document_id}_{r.chunk_id}", r.score, r.metadata.get("search_strategy")]
                         for r in res],
        })

    S3 = [["full_query", 0.4], ["concepts", 0.6], ["expanded", 0.3]]
    S2 = [["full_query", 0.4], ["concepts", 0.6]]
    S1 = [["full_query", 0.4]]
    add({1: {"full_query": [["a", 0.9], ["b", 0.5]], "concepts": [["a", 0.6], ["c", 0.7]],
             "expanded": [["b", 0.9]]}}, S3, 5, note="three strategies one doc")
    add({1: {"full_query": [["a", 0.9]], "concepts": [["a", 0.6]]}}, S2, 5,
        note="0.9*0.4 == 0.6*0.6 exact tie: first strategy wins")
    add({1: {"full_query": [["a", 0.5]]}, 2: {"full_query": [["a", 0.5], ["b", 0.75]]}}, S1, 5,
        min_score=0.3, note="two docs, same chunk id different doc")
    add({}, S1, 5, note="no docs")
    for t in range(8):
        docs = {}
        for d in range(1, int(rng.integers(2, 5))):
            docs[d] = {}
            for name, _ in S3:
                n = int(rng.integers(0, 12))
                ids = rng.choice(15, size=n, replace=False)
                sc = np.sort(rng.uniform(0.1, 1.0, n))[::-1]
                docs[d][name] = [[f"doc_{d}_chunk_{int(i)}", float(s)] for i, s in zip(ids, sc)]
        add(docs, S3 if t % 2 else S2, int(rng.integers(1, 12)), note=f"random {t}")
    return {"source": "vector_db/search_manager.py:834-936 _semantic_search (fusion lines 877-932)",
            "cases": cases}


# ----------------------------------------------------------------------------- bm25
def gen_bm25(mods, rng):
    C = mods["se"].CorpusStatsManager()
    out = {"source": "vector_db/search_manager.py:44-263 CorpusStatsManager"}
    queries = ["The quick brown fox jumps over the lazy dog in AI", "", "to be or not to be",
               "GPU kernels for dense retrieval: MI355X, HBM3E & MFMA!", "a an the", "x yz abc",
               "Ünïcode naïve café tokens_with_underscores 123 4567", "is it ok"]
    out["preprocess"] = [[q, C.preprocess_query(q)] for q in queries]
    corpus = [
        "quick quick fox jumps high",
        "the lazy dog sleeps all day long",
        "dense retrieval with inner product search over embeddings",
        "fox and dog are friends; the fox is quick",
        "",
        "retrieval augmented generation uses dense retrieval and keyword search",
        "matrix cores accelerate inner product search",
    ]
    db = MagicMock()
    rows = [types.SimpleNamespace(text=t) for t in corpus]
    db.query.return_value.filter.return_value.all.return_value = rows
    stats = run(C.get_corpus_stats([1, 2, 3], db))
    out["corpus"] = corpus
    out["stats"] = {"total_docs": stats["total_docs"], "avg_doc_length": stats["avg_doc_length"],
                    "idf": dict(stats["idf"])}
    sc = []
    known = {"total_docs": 3, "avg_doc_length": 5.0, "idf": {"quick": 1.2, "fox": 0.7}}
    sc.append({"terms": ["quick", "fox"], "text": "quick quick fox jumps high", "stats": known,
               "expected": C.calculate_bm25_score(["quick", "fox"], "quick quick fox jumps high", known)})
    for q in ["quick fox", "dense retrieval search", "inner product", "unseen words only", "dog dog dog",
              "retrieval"]:
        terms = C.preprocess_query(q)
        for text in corpus + ["   ", "inner inner inner product product", "zebra"]:
            sc.append({"terms": terms, "text": text, "stats": None,
                       "expected": C.calculate_bm25_score(terms, text, stats)})
    zero = {"total_docs": 1, "avg_doc_length": 0, "idf": {}}
    sc.append({"terms": ["quick", "zebra"], "text": "Quick brown", "stats": zero,
               "expected": C.calculate_bm25_score(["quick", "zebra"], "Quick brown", zero)})
    out["scores"] = sc
    norm = []
    for method in ("minmax", "sigmoid", "softmax", "bogus"):
        for s in ([], [0.5], [-3.0], [0.5, 2.0, 1.25], [0.0, 0.0, 0.0], [2.0, 2.0], [1e3, -1e3, 0.0],
                  [float(x) for x in rng.uniform(0, 5, 9)]):
            norm.append({"method": method, "scores": s, "expected": C.normalize_scores(list(s), method)})
    out["normalize"] = norm
    return out


# ------------------------------------------------------------------------------ mmr
def _clustered(rng, n, d, n_clusters, noise):
    cent = rng.standard_normal((n_clusters, d))
    lab = rng.integers(0, n_clusters, n)
    e = cent[lab] + noise * rng.standard_normal((n, d))
    return e


def gen_mmr(mods, rng):
    M = mods["mmr"].MMRDiversifier
    cases = []

    def add(emb, scores, lam, thr, metric, max_results, note="", store=True, seed=None, gen=None,
            as_float32=False):
        results = []
        for e, s in zip(emb, scores):
            r = {"score": s}
            if e is not None:
                r["embedding"] = [float(v) for v in e]
            results.append(r)
        for i, r in enumerate(results):
            r["_i"] = i
        m = M(lambda_param=lam, similarity_threshold=thr, max_results=None, similarity_metric=metric)
        t0 = time.perf_counter()
        out = run(m.diversify(results, lambda_param=lam, max_results=max_results))
        dt = time.perf_counter() - t0
        c = {"note": note, "lambda": lam, "threshold": thr, "metric": metric,
             "max_results": max_results, "scores": [float(s) for s in scores],
             "expected": [r["_i"] for r in out], "reference_seconds": dt}
        if store:
            c["embeddings"] = [None if e is None else [float(v) for v in e] for e in emb]
        else:
            c["gen"] = gen
            c["seed"] = seed
            c["checksum"] = float(np.sum(np.asarray([e for e in emb], dtype=np.float64)))
        cases.append(c)

    d = 16
    e = _clustered(rng, 24, d, 4, 0.05)
    s = rng.uniform(0.2, 0.95, 24)
    for lam in (0.0, 0.3, 0.5, 0.7, 1.0):
        add(e, s, lam, 0.8, "cosine", 10, f"clustered cosine lambda={lam}")
    add(e, s, 0.5, 0.0, "cosine", 10, "threshold 0: always penalise")
    add(e, s, 0.5, 1.0, "cosine", 10, "threshold 1.0")
    add(e, s, 0.5, 0.8, "dot_product", 10, "dot_product metric (unnormalised rows)")
    add(e, s, 0.5, 0.1, "euclidean", 10, "euclidean metric 1/(1+dist)")
    add(e, s, 0.5, 0.8, "manhattan", 10, "unknown metric falls back to cosine")
    add(e, s, 0.5, 0.8, "cosine", None, "max_results None -> all")
    add(e, s, 0.5, 0.8, "cosine", 100, "max_results above n")
    add(e, [0.5] * 24, 0.5, 0.8, "cosine", 8, "all-equal scores -> relevance 1.0")
    ez = e.copy()
    ez[3] = 0.0
    ez[10] = 0.0
    add(ez, s, 0.4, 0.0, "cosine", 12, "zero vectors stay zero")
    emiss = [None if i in (0, 5, 6) else row for i, row in enumerate(e)]
    emiss[9] = []
    add(emiss, s, 0.5, 0.8, "cosine", 10, "entries without an embedding are not candidates")
    add([None] * 5, [0.9, 0.1, 0.5, 0.3, 0.2], 0.5, 0.8, "cosine", 3, "no embeddings at all")
    dup = np.vstack([e[:6], e[:6], e[:6]])
    sdup = np.concatenate([s[:6], s[:6], s[:6]])
    add(dup, sdup, 0.5, 0.8, "cosine", 12, "exact duplicates and exactly tied scores")
    add(e[:1], s[:1], 0.5, 0.8, "cosine", 5, "single candidate")
    en = e / np.linalg.norm(e, axis=1, keepdims=True)
    add(en.astype(np.float32), s, 0.6, 0.8, "cosine", 10, "unit fp32 vectors, defaults of SearchFilters")
    # the configuration the target names: 200 candidates, d=768, lambda 0.5, thr 0.8 -> 10
    for seed, gen in ((0, "random_unit"), (1, "clustered_unit")):
        r = np.random.default_rng(seed)
        if gen == "random_unit":
            big = r.standard_normal((200, 768))
        else:
            big = _clustered(r, 200, 768, 12, 0.25)
        big = (big / np.linalg.norm(big, axis=1, keepdims=True)).astype(np.float32)
        sc = np.sort(r.uniform(0.3, 0.9, 200))[::-1]
        add(big, sc, 0.5, 0.8, "cosine", 10, f"200x768 {gen}", store=False, seed=seed, gen=gen)
    return {"source": "vector_db/mmr_diversifier.py:65-289 MMRDiversifier.diversify", "cases": cases}


def gen_mmr_analysis(mods, rng):
    """compute_similarity_matrix and analyze_diversity of the reference class."""
    M = mods["mmr"].MMRDiversifier
    SR = mods["schemas"].SearchResult
    sims, sets = [], {"none": []}
    e = _clustered(rng, 14, 16, 3, 0.1)
    e[4] = 0.0                                   # a zero row stays zero under cosine
    e[9] = e[2]                                  # an exact duplicate
    wide = _clustered(rng, 9, 300, 2, 0.3)       # rows wider than a wave's 64 lanes x 4
    unit32 = (e[:6] / np.maximum(np.linalg.norm(e[:6], axis=1, keepdims=True), 1e-30)).astype(np.float32)
    for name, emb in (("clustered 14x16 with a zero row and a duplicate", e), ("9x300", wide), ("fp32 rows", unit32),
                      ("one row", e[:1])):
        for metric in ("cosine", "dot_product", "euclidean", "manhattan"):
            m = M(similarity_metric=metric)
            S = m.compute_similarity_matrix([np.asarray(r) for r in emb])
            sims.append({"note": name, "metric": metric, "dtype": str(emb.dtype), "set": name,
                         "expected": [[float(v) for v in r] for r in S]})
        sets[name] = [[float(v) for v in r] for r in emb]
    empty = M().compute_similarity_matrix([])
    sims.append({"note": "no embeddings", "metric": "cosine", "dtype": "float64", "set": "none",
                 "expected": [float(v) for v in empty], "expected_shape": list(empty.shape)})
    div = []
    words = "dense retrieval inner product search keyword matrix cores".split()

    def results(n, docs):
        return [{"document_id": docs[i % len(docs)], "chunk_id": f"c{i}", "score": float(rng.uniform()),
                 "text": " ".join(rng.choice(words, int(rng.integers(0, 12))))} for i in range(n)]

    for note, rs, as_objects in (("dicts, three documents", results(10, [1, 2, 3]), False),
                                 ("SearchResult objects, str and int ids", results(7, ["a", 5, "b", 5]), True),
                                 ("one result", results(1, [9]), False),
                                 ("dicts without text or document_id", [{"score": 0.5}, {"score": 0.1, "text": "abc"}], False)):
        given = [SR(**r) for r in rs] if as_objects else rs
        out = M().analyze_diversity(given)
        div.append({"note": note, "as_objects": as_objects, "results": rs,
                    "expected": {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in out.items()}})
    div.append({"note": "no results", "as_objects": False, "results": [], "expected": M().analyze_diversity([])})
    return {"source": "vector_db/mmr_diversifier.py:291-366 MMRDiversifier.compute_similarity_matrix / analyze_diversity",
            "embedding_sets": sets, "similarity": sims, "diversity": div}


# --------------------------------------------------------------------- cosine scan
def gen_cosine_scan(mods, rng):
    E = mods["fb"].EmbeddedVectorDB
    cases = []
    with tempfile.TemporaryDirectory() as td:
        for t, (n, d, limit) in enumerate([(40, 8, 10), (64, 32, 5), (30, 16, 30), (3, 4, 10)]):
            db = E(os.path.join(td, f"v{t}.db"))
            db.create_collection("c", d)
            v = rng.standard_normal((n, d)).astype(np.float32)
            if t == 1:
                v[7] = 0.0
                v[9] = v[3]
            q = rng.standard_normal(d).astype(np.float32)
            if t == 3:
                q[:] = 0.0
            ids = [f"p{i:03d}" for i in range(n)]
            db.upsert_vectors("c", [list(map(float, r)) for r in v], [{"i": i} for i in range(n)], ids)
            res = db.search_vectors("c", list(map(float, q)), limit)
            cases.append({"vectors": [list(map(float, r)) for r in v], "query": list(map(float, q)),
                          "limit": limit,
                          "expected": [[r["payload"]["i"], r["score"]] for r in res]})
    # the health-check smoke vectors (vector_db/health_checker.py:228-249)
    return {"source": "utils/service_fallbacks.py:101-144 EmbeddedVectorDB.search_vectors",
            "cases": cases}


def main():
    os.makedirs(OUT, exist_ok=True)
    mods = load_reference()
    gens = {
        "fusion_contextual.json": gen_contextual,
        "fusion_hybrid.json": gen_hybrid,
        "fusion_multi_embedding.json": gen_multi_embedding,
        "bm25.json": gen_bm25,
        "mmr.json": gen_mmr,
        "cosine_scan.json": gen_cosine_scan,
        "mmr_analysis.json": gen_mmr_analysis,
    }
    only = set(sys.argv[1:])                     # file names to (re)write; none = all
    for name, fn in gens.items():
        if only and name not in only:
            continue
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        data = fn(mods, rng)
        data["generated_by"] = "oracle/gen_golden.py"
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(data, f, indent=None, separators=(",", ":"))
        print(name, "ok", os.path.getsize(os.path.join(OUT, name)), "bytes")


if __name__ == "__main__":
    main()
