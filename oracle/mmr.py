"""CPU oracle: greedy Maximal-Marginal-Relevance selection as the reference does
it (vector_db/mmr_diversifier.py:151-289), in numpy float64.  TEST
INFRASTRUCTURE ONLY.  Pinned by tests/golden/mmr_*.json.
"""
from __future__ import annotations

import numpy as np


def _cosine_rows(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """sklearn.metrics.pairwise.cosine_similarity as called at
    mmr_diversifier.py:258 -- each side row-normalised by sqrt(sum(x*x)), a zero
    norm replaced by 1 so a zero row stays zero, then a plain matrix product."""
    def _norm(x):
        n = np.sqrt(np.einsum("ij,ij->i", x, x))
        n[n == 0.0] = 1.0
        return x / n[:, None]
    return _norm(a) @ _norm(b).T


def normalise_relevance(scores):
    """mmr_diversifier.py:188-200: min-max to [0,1]; all equal -> 1.0 each."""
    scores = [float(s) for s in scores]
    if not scores:
        return []
    lo, hi = min(scores), max(scores)
    if hi > lo:
        return [(s - lo) / (hi - lo) for s in scores]
    return [1.0] * len(scores)


def mmr_select(embeddings, scores, lambda_param, max_results=None,
               similarity_threshold=0.8, metric="cosine"):
    """Returns the original indices in selection order.

    ``embeddings``: list whose entries are 1-D arrays/lists or None; an entry that
    is None or empty is not a candidate (:165-168).  No valid entry at all ->
    ``range(min(max_results, n))`` (:170-172).  ``max_results`` None -> n
    (:89-92), then capped by the number of valid candidates (:211).
    Each round walks the remaining candidates in order; similarity of the
    candidate to every selected one by ``metric`` ('cosine' | 'dot_product' |
    'euclidean' = 1/(1+dist); anything else = cosine, :257-268); the largest is
    zeroed when it is below ``similarity_threshold`` (:271-273); score =
    l*rel - (1-l)*maxsim (:278); strict ``>`` keeps the first maximum (:280-282).
    """
    n = len(embeddings)
    if max_results is None:
        max_results = n
    max_results = min(max_results, n)
    valid = [i for i, e in enumerate(embeddings) if e is not None and len(e) > 0]
    if not valid:
        return list(range(min(max_results, n)))
    rel = normalise_relevance([scores[i] for i in valid])
    E = np.vstack([np.array(embeddings[i]) for i in valid]).astype(np.float64)
    max_results = min(max_results, len(valid))
    selected = []
    remaining = list(range(len(valid)))
    while len(selected) < max_results and remaining:
        best = -float("inf")
        best_pos = 0
        for pos, c in enumerate(remaining):
            if selected:
                ce = E[c].reshape(1, -1)
                se = E[selected]
                if metric == "dot_product":
                    sims = np.dot(ce, se.T)[0]
                elif metric == "euclidean":
                    sims = 1.0 / (1.0 + np.linalg.norm(ce - se, axis=1))
                else:
                    sims = _cosine_rows(ce, se)[0]
                m = np.max(sims)
                if m < similarity_threshold:
                    m = 0.0
            else:
                m = 0.0
            v = lambda_param * rel[c] - (1 - lambda_param) * m
            if v > best:
                best = v
                best_pos = pos
        selected.append(remaining.pop(best_pos))
    return [valid[i] for i in selected]


def mmr_select_fast(E, scores, lambda_param, max_results=None,
                    similarity_threshold=0.8, metric="cosine"):
    """Same selection for a dense [n, d] matrix with every row valid, one
    similarity column per round instead of one call per pair (used for the CPU
    baseline timing and for larger parity cases; checked against ``mmr_select``)."""
    E = np.asarray(E, dtype=np.float64)
    n = E.shape[0]
    if max_results is None:
        max_results = n
    max_results = min(max_results, n)
    rel = np.array(normalise_relevance(scores), dtype=np.float64)
    if metric == "dot_product":
        simcol = lambda s: E @ E[s]
    elif metric == "euclidean":
        simcol = lambda s: 1.0 / (1.0 + np.linalg.norm(E - E[s][None, :], axis=1))
    else:
        nr = np.sqrt(np.einsum("ij,ij->i", E, E))
        nr[nr == 0.0] = 1.0
        En = E / nr[:, None]
        simcol = lambda s: En @ En[s]
    maxsim = np.full(n, -np.inf)
    alive = np.ones(n, dtype=bool)
    out = []
    for _ in range(max_results):
        if out:
            m = np.where(maxsim < similarity_threshold, 0.0, maxsim)
        else:
            m = np.zeros(n)
        v = lambda_param * rel - (1 - lambda_param) * m
        v[~alive] = -np.inf
        best = int(np.argmax(v))          # first maximum
        if not alive[best]:
            break
        out.append(best)
        alive[best] = False
        maxsim = np.maximum(maxsim, simcol(best))
    return out


def similarity_matrix(embeddings, metric="cosine"):
    """mmr_diversifier.py:291-319 compute_similarity_matrix: no embeddings -> an empty 1-D array; else
    the n x n matrix of the stacked rows under 'cosine' (row-normalised product), 'dot_product'
    (plain product), 'euclidean' (1 / (1 + distance), sklearn's distances: the diagonal is exactly
    zero, the rest sqrt of the clamped |x|^2 - 2 x.y + |y|^2); anything else = cosine."""
    if not embeddings:
        return np.array([])
    E = np.vstack(embeddings)
    if metric == "dot_product":
        return np.dot(E, E.T)
    if metric == "euclidean":
        sq = np.einsum("ij,ij->i", E, E)
        d2 = sq[:, None] - 2.0 * (E @ E.T) + sq[None, :]
        np.maximum(d2, 0.0, out=d2)
        np.fill_diagonal(d2, 0.0)
        return 1.0 / (1.0 + np.sqrt(d2))
    return _cosine_rows(E, E)


def analyze_diversity(results):
    """mmr_diversifier.py:321-366 analyze_diversity on dict results ('text', 'document_id'): counts,
    the share of distinct documents, mean and variance of the text lengths."""
    if not results:
        return {"error": "No results provided"}
    texts = [r.get("text", "") for r in results]
    unique = len(set(r.get("document_id") for r in results))
    lens = [len(t) for t in texts]
    return {"total_results": len(results), "unique_documents": unique,
            "document_diversity_ratio": unique / len(results),
            "avg_text_length": float(np.mean(lens)) if texts else 0,
            "text_length_variance": float(np.var(lens)) if texts else 0}
