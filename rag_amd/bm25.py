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
