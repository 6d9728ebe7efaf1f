"""CPU oracle: the reference's BM25 scoring and score normalisation
(vector_db/search_manager.py:44-263 ``CorpusStatsManager``).  TEST
INFRASTRUCTURE ONLY.  Pinned by tests/golden/bm25.json.
"""
from __future__ import annotations

import math
import re
from collections import Counter, defaultdict
from typing import Dict, List, Sequence

# vector_db/search_manager.py:50-55
STOPWORDS = {
    'the', 'a', 'an', 'and', 'or', 'but', 'in', 'on', 'at', 'to', 'for',
    'of', 'with', 'by', 'is', 'are', 'was', 'were', 'be', 'been', 'being',
    'have', 'has', 'had', 'do', 'does', 'did', 'will', 'would', 'could',
    'should', 'may', 'might', 'must', 'can', 'this', 'that', 'these', 'those',
}

_TOKEN = re.compile(r'\b\w+\b')


def preprocess_query(text: str) -> List[str]:
    """:57-66 lowercase, ``\\b\\w+\\b`` tokens, drop stopwords and tokens of length
    <= 2; if nothing is left fall back to the first three raw tokens."""
    tokens = _TOKEN.findall(text.lower())
    kept = [t for t in tokens if t not in STOPWORDS and len(t) > 2]
    return kept if kept else tokens[:3]


def corpus_stats(texts: Sequence[str]) -> Dict:
    """:87-120 every chunk text is one BM25 document; df counts chunks holding a
    term at least once; idf = max(ln((N-df+.5)/(df+.5)), 0.01); no chunks ->
    default stats (:133-140: N=1, avgdl=100, idf 1.0 for every term)."""
    if not texts:
        return {'total_docs': 1, 'avg_doc_length': 100, 'idf': defaultdict(lambda: 1.0)}
    lengths = []
    df = defaultdict(int)
    for t in texts:
        toks = preprocess_query(t or "")
        lengths.append(len(toks))
        for term in set(toks):
            df[term] += 1
    n = len(texts)
    avg = sum(lengths) / len(lengths) if lengths else 0
    idf = {term: max(math.log((n - f + 0.5) / (f + 0.5)), 0.01) for term, f in df.items()}
    return {'total_docs': n, 'avg_doc_length': avg, 'idf': idf}


def bm25_score(query_terms: Sequence[str], document_text: str, stats: Dict,
               k1: float = 1.5, b: float = 0.75) -> float:
    """:142-180.  Terms absent from the idf table weigh 0.5 (:166); the sum runs
    over ``query_terms`` in order, repeats included; result clamped at 0.  A zero
    ``avg_doc_length`` raises ZeroDivisionError in the reference, which falls back
    to the share of query terms found as substrings (:177-180)."""
    try:
        if not query_terms or not document_text.strip():
            return 0.0
        toks = preprocess_query(document_text)
        dl = len(toks)
        avgdl = stats.get('avg_doc_length', 100)
        idf = stats.get('idf', {})
        tf = Counter(toks)
        score = 0.0
        for term in query_terms:
            f = tf.get(term, 0)
            if f == 0:
                continue
            w = idf.get(term, 0.5)
            num = f * (k1 + 1)
            den = f + k1 * (1 - b + b * (dl / avgdl))
            score += w * (num / den)
        return max(score, 0.0)
    except Exception:
        return sum(1 for t in query_terms if t in document_text.lower()) / len(query_terms)


def _sigmoid(x: float, scale: float = 2.0) -> float:
    """:235-241."""
    try:
        return 1.0 / (1.0 + math.exp(-scale * x))
    except OverflowError:
        return 0.0 if x < 0 else 1.0


def normalize_scores(scores: Sequence[float], method: str = "minmax") -> List[float]:
    """:182-263.  [] -> []; one score -> sigmoid(2x) whatever the method; minmax
    with all scores equal -> 0.0 each if that score is 0 else 1.0 each."""
    scores = list(scores)
    if not scores:
        return []
    if len(scores) == 1:
        return [_sigmoid(scores[0])]
    if method == "sigmoid":
        return [_sigmoid(s) for s in scores]
    if method == "softmax":
        m = max(scores)
        ex = []
        for s in scores:
            try:
                ex.append(math.exp(s - m))
            except OverflowError:
                ex.append(1e-10)
        tot = sum(ex)
        if tot == 0:
            return [1.0 / len(scores)] * len(scores)
        return [e / tot for e in ex]
    lo, hi = min(scores), max(scores)
    if hi == lo:
        return [0.0 if hi == 0.0 else 1.0] * len(scores)
    return [(s - lo) / (hi - lo) for s in scores]


def keyword_rank(query: str, candidate_texts: Sequence[str], corpus_texts: Sequence[str],
                 limit: int, min_score=None):
    """:1121-1219 without the SQL: query terms, BM25 per candidate, minmax
    normalise, keep ``min_score is None or s >= min_score``, stable sort
    descending, first ``limit``.  Returns [(candidate_index, normalised, raw)]."""
    terms = preprocess_query(query)
    if not terms or not candidate_texts:
        return []
    stats = corpus_stats(corpus_texts)
    raw = [bm25_score(terms, t or "", stats) for t in candidate_texts]
    norm = normalize_scores(raw, "minmax")
    rows = [(i, n, r) for i, (n, r) in enumerate(zip(norm, raw))
            if min_score is None or n >= min_score]
    rows.sort(key=lambda t: t[1], reverse=True)
    return rows[:limit]
