"""Keyword side of hybrid search: query preprocessing, corpus statistics, BM25 and
score normalisation with the behaviour of the reference's ``CorpusStatsManager``
(vector_db/search_manager.py:44-263).  Text tokenisation is host work (regex over
Python strings) in the reference and here; only the (candidate, score) pairs go
on to the GPU fusion."""
from __future__ import annotations

import logging
import math
import re
from collections import Counter, defaultdict
from typing import Any, Dict, List, Optional, Sequence, Set

logger = logging.getLogger(__name__)

_WORD = re.compile(r"\b\w+\b")


class CorpusStatsManager:
    def __init__(self):
        self.stats_cache: Dict[str, Dict] = {}
        self.stopwords: Set[str] = set(
            "the a an and or but in on at to for of with by is are was were be been being have has had do does "
            "did will would could should may might must can this that these those".split())

    def preprocess_query(self, query: str) -> List[str]:
        """:57-66."""
        tokens = _WORD.findall(query.lower())
        kept = [t for t in tokens if t not in self.stopwords and len(t) > 2]
        return kept if kept else tokens[:3]

    def _get_default_stats(self) -> Dict[str, Any]:
        """:133-140."""
        return {"total_docs": 1, "avg_doc_length": 100, "idf": defaultdict(lambda: 1.0)}

    def compute_corpus_stats(self, texts: Sequence[Optional[str]], cache_key: Optional[str] = None) -> Dict[str, Any]:
        """:87-127 over the chunk texts (the reference reads them with one SQL query, :83-85)."""
        if cache_key is not None and cache_key in self.stats_cache and len(self.stats_cache) < 100:
            return self.stats_cache[cache_key]
        if not texts:
            return self._get_default_stats()
        lengths, df = [], defaultdict(int)
        for text in texts:
            tokens = self.preprocess_query(text or "")
            lengths.append(len(tokens))
            for term in set(tokens):
                df[term] += 1
        n = len(texts)
        stats = {"total_docs": n, "avg_doc_length": sum(lengths) / len(lengths) if lengths else 0,
                 "idf": {t: max(math.log((n - f + 0.5) / (f + 0.5)), 0.01) for t, f in df.items()}}
        if cache_key is not None:
            self.stats_cache[cache_key] = stats
        return stats

    def calculate_bm25_score(self, query_terms: List[str], document_text: str, corpus_stats: Dict[str, Any],
                             k1: float = 1.5, b: float = 0.75) -> float:
        """:142-180."""
        try:
            if not query_terms or not document_text.strip():
                return 0.0
            doc_tokens = self.preprocess_query(document_text)
            dl = len(doc_tokens)
            avgdl = corpus_stats.get("avg_doc_length", 100)
            idf = corpus_stats.get("idf", {})
            tf = Counter(doc_tokens)
            score = 0.0
            for term in query_terms:
                f = tf.get(term, 0)
                if f == 0:
                    continue
                score += idf.get(term, 0.5) * ((f * (k1 + 1)) / (f + k1 * (1 - b + b * (dl / avgdl))))
            return max(score, 0.0)
        except Exception as e:
            logger.error(f"BM25 calculation failed: {e}")
            return sum(1 for t in query_terms if t in document_text.lower()) / len(query_terms)

    def calculate_bm25_scores(self, query_terms: List[str], document_texts: Sequence[Optional[str]],
                              corpus_stats: Dict[str, Any], k1: float = 1.5, b: float = 0.75) -> List[float]:
        """``calculate_bm25_score`` (:142-180) for a list of candidate texts at once: every text is
        tokenised once, then the terms are accumulated in query order with one array operation
        per term.  The per-document operation order is the scalar one, so the float64 results
        are bit-identical (a term that does not occur adds exactly 0.0)."""
        import numpy as np
        n = len(document_texts)
        if not query_terms or n == 0:
            return [0.0] * n
        avgdl = corpus_stats.get("avg_doc_length", 100)
        idf = corpus_stats.get("idf", {})
        dl = np.zeros(n, dtype=np.float64)
        tf = np.zeros((len(query_terms), n), dtype=np.float64)
        live = np.zeros(n, dtype=bool)
        for j, text in enumerate(document_texts):
            if not text or not text.strip():
                continue
            live[j] = True
            counts = Counter(self.preprocess_query(text))
            dl[j] = sum(counts.values())
            for t, term in enumerate(query_terms):
                tf[t, j] = counts.get(term, 0)
        denom_len = k1 * (1 - b + b * (dl / avgdl))
        score = np.zeros(n, dtype=np.float64)
        for t, term in enumerate(query_terms):
            f = tf[t]
            with np.errstate(invalid="ignore", divide="ignore"):
                part = idf.get(term, 0.5) * ((f * (k1 + 1)) / (f + denom_len))
            score = score + np.where(f > 0, part, 0.0)
        score = np.where(live, np.maximum(score, 0.0), 0.0)
        return score.tolist()

    @staticmethod
    def _sigmoid_normalize(score: float, scale: float = 2.0) -> float:
        try:
            return 1.0 / (1.0 + math.exp(-scale * score))
        except OverflowError:
            return 0.0 if score < 0 else 1.0

    def normalize_scores(self, scores: List[float], method: str = "minmax") -> List[float]:
        """:182-263."""
        if not scores:
            return []
        if len(scores) == 1:
            return [self._sigmoid_normalize(scores[0])]
        if method == "sigmoid":
            return [self._sigmoid_normalize(s) for s in scores]
        if method == "softmax":
            m = max(scores)
            ex = []
            for s in scores:
                try:
                    ex.append(math.exp(s - m))
                except OverflowError:
                    ex.append(1e-10)
            total = sum(ex)
            return [1.0 / len(scores)] * len(scores) if total == 0 else [e / total for e in ex]
        if method != "minmax":
            logger.warning(f"Unknown normalization method: {method}, using minmax")
        lo, hi = min(scores), max(scores)
        if hi == lo:
            return [0.0 if hi == 0.0 else 1.0] * len(scores)
        return [(s - lo) / (hi - lo) for s in scores]

    def keyword_rank(self, query: str, candidate_texts: Sequence[str], corpus_texts: Sequence[str], limit: int,
                     min_score: Optional[float] = None):
        """:1121-1219 without the SQL prefilter: [(candidate index, normalised, raw)] best first."""
        terms = self.preprocess_query(query)
        if not terms or not candidate_texts:
            return []
        stats = self.compute_corpus_stats(corpus_texts)
        raw = self.calculate_bm25_scores(terms, candidate_texts, stats)
        norm = self.normalize_scores(raw, "minmax")
        rows = [(i, n, r) for i, (n, r) in enumerate(zip(norm, raw)) if min_score is None or n >= min_score]
        rows.sort(key=lambda t: t[1], reverse=True)
        return rows[:limit]


class DeviceKeywordIndex:
    """The keyword corpus on the GPU.  A text is tokenised once, when it is added
    (``preprocess_query``, vector_db/search_manager.py:57-66); its kept tokens go to HBM as
    vocabulary ids.  A query then costs one small launch: term frequencies, BM25
    (:142-180), min-max normalisation (:182-263), the ``min_score`` filter and the stable
    descending sort of ``_keyword_search`` (:1159-1213) run on the device with the
    reference's fp64 operation order, so scores are bit-identical to ``CorpusStatsManager``.
    Document numbers are positions in the order of ``add_texts``: use the same order as the
    rows of the vector index and the ranked hits feed ``hybrid_mmr`` as they are."""

    MAX_TERMS = 64
    MAX_RANK = 4096

    def __init__(self, device: int = 0, stats_manager: Optional[CorpusStatsManager] = None):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        self.manager = stats_manager or CorpusStatsManager()
        self.device = device
        self.vocab: Dict[str, int] = {}
        self._doc_ids: List[Any] = []          # per document: numpy int32 token ids (host copy for the statistics)
        h = C.c_void_p()
        _lib.check(_lib.lib.rag_bm25_create(device, C.byref(h)))
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.lib.rag_bm25_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return len(self._doc_ids)

    def add_texts(self, texts: Sequence[Optional[str]]) -> range:
        import numpy as np
        first = len(self._doc_ids)
        docs = []
        for text in texts:
            ids = [self.vocab.setdefault(t, len(self.vocab)) for t in self.manager.preprocess_query(text or "")]
            docs.append(np.asarray(ids, dtype=np.int32))
        if docs:
            lens = np.asarray([d.size for d in docs], dtype=np.int64)
            flat = np.ascontiguousarray(np.concatenate(docs) if lens.sum() else np.zeros(0, np.int32), dtype=np.int32)
            self._lib.check(self._lib.lib.rag_bm25_add(self._h, len(docs), lens.ctypes.data, flat.ctypes.data if flat.size else None))
            self._doc_ids.extend(docs)
        return range(first, len(self._doc_ids))

    def corpus_stats(self, docs: Optional[Sequence[int]] = None) -> Dict[str, Any]:
        """``compute_corpus_stats`` (:87-127) over the stored documents (or the given subset)."""
        import numpy as np
        chosen = self._doc_ids if docs is None else [self._doc_ids[i] for i in docs]
        if not chosen:
            return self.manager._get_default_stats()
        n = len(chosen)
        lengths = [int(d.size) for d in chosen]
        uniq = [np.unique(d) for d in chosen]
        df = np.bincount(np.concatenate(uniq), minlength=len(self.vocab)) if sum(u.size for u in uniq) else np.zeros(0, np.int64)
        terms = list(self.vocab)
        return {"total_docs": n, "avg_doc_length": sum(lengths) / len(lengths),
                "idf": {terms[t]: max(math.log((n - int(f) + 0.5) / (int(f) + 0.5)), 0.01) for t, f in enumerate(df) if f}}

    def _terms(self, query_terms: Sequence[str], stats: Dict[str, Any]):
        import numpy as np
        if len(query_terms) > self.MAX_TERMS:
            raise ValueError(f"at most {self.MAX_TERMS} query terms")
        idf = stats.get("idf", {})
        ids = np.asarray([self.vocab.get(t, -1) for t in query_terms], dtype=np.int32)
        w = np.asarray([idf.get(t, 0.5) for t in query_terms], dtype=np.float64)
        avgdl = float(stats.get("avg_doc_length", 100))
        if avgdl == 0.0:
            raise ValueError("avg_doc_length == 0: the reference's substring fallback needs the texts (CorpusStatsManager)")
        return ids, w, avgdl

    def score(self, query_terms: Sequence[str], candidates: Optional[Sequence[int]] = None,
              stats: Optional[Dict[str, Any]] = None, k1: float = 1.5, b: float = 0.75) -> List[float]:
        """``calculate_bm25_score`` for every candidate document (default: all), one launch."""
        import numpy as np
        m = len(self) if candidates is None else len(candidates)
        if not query_terms or m == 0:
            return [0.0] * m
        stats = stats if stats is not None else self.corpus_stats()
        ids, w, avgdl = self._terms(query_terms, stats)
        cand = None if candidates is None else np.ascontiguousarray(candidates, dtype=np.int64)
        out = np.empty(m, dtype=np.float64)
        self._lib.check(self._lib.lib.rag_bm25_score(self._h, m, None if cand is None else cand.ctypes.data, len(ids),
                                                     ids.ctypes.data, w.ctypes.data, avgdl, k1, b, out.ctypes.data))
        return out.tolist()

    def rank(self, query: str, candidates: Optional[Sequence[int]] = None, limit: int = 10,
             min_score: Optional[float] = None, stats: Optional[Dict[str, Any]] = None,
             k1: float = 1.5, b: float = 0.75):
        """``_keyword_search`` after its SQL prefilter: [(document, normalised, raw)] best first."""
        import numpy as np
        terms = self.manager.preprocess_query(query)
        m = len(self) if candidates is None else len(candidates)
        if not terms or m == 0:
            return []
        stats = stats if stats is not None else self.corpus_stats()
        ids, w, avgdl = self._terms(terms, stats)
        cand = None if candidates is None else np.ascontiguousarray(candidates, dtype=np.int64)
        rows = np.empty(limit, dtype=np.int64)
        norm = np.empty(limit, dtype=np.float64)
        raw = np.empty(limit, dtype=np.float64)
        n = self._C.c_int(0)
        self._lib.check(self._lib.lib.rag_bm25_rank(
            self._h, m, None if cand is None else cand.ctypes.data, len(ids), ids.ctypes.data, w.ctypes.data, avgdl, k1, b,
            0 if min_score is None else 1, 0.0 if min_score is None else float(min_score), limit,
            rows.ctypes.data, norm.ctypes.data, raw.ctypes.data, self._C.byref(n)))
        return [(int(rows[i]), float(norm[i]), float(raw[i])) for i in range(n.value)]

    def rank_device(self, query: str, candidates=None, limit: int = 10, min_score: Optional[float] = None,
                    stats: Optional[Dict[str, Any]] = None, k1: float = 1.5, b: float = 0.75):
        """The same ranking with torch CUDA tensors in and out and no synchronisation:
        (rows int64[limit] padded with -1, normalised float64[limit], raw float64[limit], count int32[1]) --
        ``rows`` / ``normalised`` are what ``hybrid_mmr`` takes as ``bm25_rows`` / ``bm25_scores``.
        ``candidates``: int64 CUDA tensor of document numbers, or None for every document (at most 4096)."""
        import torch
        dev = torch.device("cuda", self.device)
        terms = self.manager.preprocess_query(query)
        m = len(self) if candidates is None else int(candidates.numel())
        rows = torch.full((limit,), -1, dtype=torch.int64, device=dev)
        norm = torch.zeros(limit, dtype=torch.float64, device=dev)
        raw = torch.zeros(limit, dtype=torch.float64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        if not terms or m == 0:
            return rows, norm, raw, cnt
        stats = stats if stats is not None else self.corpus_stats()
        ids, w, avgdl = self._terms(terms, stats)
        ids_d = torch.from_numpy(ids).to(dev, non_blocking=True)
        w_d = torch.from_numpy(w).to(dev, non_blocking=True)
        if candidates is not None and (candidates.dtype != torch.int64 or not candidates.is_cuda or not candidates.is_contiguous()):
            raise ValueError("candidates must be a contiguous int64 CUDA tensor")
        scratch = torch.empty(m, dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        self._lib.check(self._lib.lib.rag_bm25_rank_device(
            self._h, m, None if candidates is None else candidates.data_ptr(), len(ids), ids_d.data_ptr(), w_d.data_ptr(),
            avgdl, k1, b, 0 if min_score is None else 1, 0.0 if min_score is None else float(min_score), limit,
            scratch.data_ptr(), rows.data_ptr(), norm.data_ptr(), raw.data_ptr(), cnt.data_ptr(), stream))
        # the staging tensors must outlive the enqueued work: tie them to the stream
        for t in (ids_d, w_d, scratch):
            t.record_stream(torch.cuda.current_stream(dev))
        return rows, norm, raw, cnt
