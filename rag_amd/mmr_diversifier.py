"""Maximal-Marginal-Relevance diversifier with the interface of the reference's
``MMRDiversifier`` (vector_db/mmr_diversifier.py:20-120, factories :384-432); the
greedy selection (:216-289) runs as one GPU kernel (``rag_mmr_select``,
rag_amd/csrc/fusion_mmr.hip) in fp64.

``diversify`` returns the same result objects, reordered / cut, and never
modifies scores, like the reference.  Results may be ``SearchResult``-like
objects (``.score``, ``.metadata['embedding']`` or ``.embedding``) or dicts
(``'embedding'`` | ``metadata.embedding`` | ``'vector'``).
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np

from . import _lib
from ._lib import check, lib

logger = logging.getLogger(__name__)

_METRICS = {"cosine": _lib.MMR_COSINE, "dot_product": _lib.MMR_DOT, "euclidean": _lib.MMR_EUCLIDEAN}


def mmr_select_device(embeddings: np.ndarray, relevance: Sequence[float], lambda_param: float,
                      similarity_threshold: float, metric: str, max_results: int, device: int = 0) -> List[int]:
    """Greedy MMR over a dense [n, d] matrix (fp32 or fp64) and normalised relevance
    scores; returns candidate positions in selection order."""
    emb = np.ascontiguousarray(embeddings)
    if emb.dtype not in (np.float32, np.float64):
        emb = emb.astype(np.float64)
    n, d = emb.shape
    rel = np.ascontiguousarray(relevance, dtype=np.float64)
    if rel.size != n:
        raise ValueError("one relevance score per embedding")
    out = np.empty(max(n, 1), dtype=np.int32)
    n_out = C.c_int32(0)
    check(lib.rag_mmr_select(n, d, emb.ctypes.data_as(C.c_void_p), 1 if emb.dtype == np.float64 else 0,
                             rel.ctypes.data_as(C.c_void_p), float(lambda_param), float(similarity_threshold),
                             _METRICS.get(metric, _lib.MMR_COSINE), int(max_results), int(device),
                             out.ctypes.data_as(C.c_void_p), C.byref(n_out)))
    return out[:n_out.value].tolist()


def similarity_matrix_device(embeddings: np.ndarray, metric: str, device: int = 0) -> np.ndarray:
    """Pair similarities of the rows of a dense [n, d] matrix (fp32 or fp64) in fp64 on the GPU: the
    values the MMR selection uses (vector_db/mmr_diversifier.py:291-319)."""
    emb = np.ascontiguousarray(embeddings)
    if emb.dtype not in (np.float32, np.float64):
        emb = emb.astype(np.float64)
    n, d = emb.shape
    out = np.empty((n, n), dtype=np.float64)
    check(lib.rag_similarity_matrix(n, d, emb.ctypes.data_as(C.c_void_p), 1 if emb.dtype == np.float64 else 0,
                                    _METRICS.get(metric, _lib.MMR_COSINE), int(device), out.ctypes.data_as(C.c_void_p)))
    return out


def mmr_select_rows(index, rows: Sequence[int], relevance: Sequence[float], lambda_param: float,
                    similarity_threshold: float, metric: str, max_results: int) -> List[int]:
    """Same selection with the embeddings taken from rows of a ``DeviceIndex``
    (nothing is copied to the host; replaces the per-result vector fetch of
    vector_db/search_manager.py:1699-1727).  ``rows`` are row positions (insertion order), not
    ids: subtract the id base / map explicit ids back before calling.  Flat and IVF indexes."""
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    rel = np.ascontiguousarray(relevance, dtype=np.float64)
    n = rows.size
    out = np.empty(max(n, 1), dtype=np.int32)
    n_out = C.c_int32(0)
    check(lib.rag_mmr_select_rows(index._h, n, rows.ctypes.data_as(C.c_void_p), rel.ctypes.data_as(C.c_void_p),
                                  float(lambda_param), float(similarity_threshold),
                                  _METRICS.get(metric, _lib.MMR_COSINE), int(max_results),
                                  out.ctypes.data_as(C.c_void_p), C.byref(n_out)))
    return out[:n_out.value].tolist()


def normalise_relevance(scores: Sequence[float]) -> List[float]:
    """vector_db/mmr_diversifier.py:188-200: min-max to [0, 1]; all equal -> 1.0."""
    scores = list(scores)
    if not scores:
        return []
    lo, hi = min(scores), max(scores)
    if hi > lo:
        return [(s - lo) / (hi - lo) for s in scores]
    return [1.0] * len(scores)


class MMRDiversifier:
    """MMR score = lambda * relevance - (1 - lambda) * max similarity to the selected."""

    def __init__(self, lambda_param: float = 0.7, similarity_threshold: float = 0.8,
                 max_results: Optional[int] = None, similarity_metric: str = "cosine", device: int = 0):
        if not 0.0 <= lambda_param <= 1.0:
            raise ValueError("lambda_param must be between 0.0 and 1.0")
        if not 0.0 <= similarity_threshold <= 1.0:
            raise ValueError("similarity_threshold must be between 0.0 and 1.0")
        self.lambda_param = lambda_param
        self.similarity_threshold = similarity_threshold
        self.max_results = max_results
        self.similarity_metric = similarity_metric
        self.device = device

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _embedding_of(result) -> Optional[np.ndarray]:
        """vector_db/mmr_diversifier.py:129-147."""
        if isinstance(result, dict):
            if "embedding" in result:
                return np.array(result["embedding"])
            if "metadata" in result and "embedding" in result["metadata"]:
                return np.array(result["metadata"]["embedding"])
            if "vector" in result:
                return np.array(result["vector"])
            return None
        emb = getattr(result, "embedding", None)
        if emb is not None and len(emb) > 0:
            return np.array(emb)
        md = getattr(result, "metadata", None)
        if md and "embedding" in md:
            return np.array(md["embedding"])
        return None

    @staticmethod
    def _score_of(result) -> float:
        if isinstance(result, dict):
            return result.get("score", 0.0)
        return getattr(result, "score", 0.0)

    def select_indices(self, results: Sequence[Any], lambda_param: float, max_results: int) -> List[int]:
        """vector_db/mmr_diversifier.py:151-214 (+ the greedy loop on the GPU)."""
        embeddings = [self._embedding_of(r) for r in results]
        valid = [i for i, e in enumerate(embeddings) if e is not None and len(e) > 0]
        if not valid:
            return list(range(min(max_results, len(results))))
        dims = {len(embeddings[i]) for i in valid}
        if len(dims) != 1:
            raise ValueError("embeddings of different lengths")      # np.vstack fails in the reference
        rel = normalise_relevance([self._score_of(results[i]) for i in valid])
        E = np.vstack([np.asarray(embeddings[i], dtype=np.float64) for i in valid])
        picked = mmr_select_device(E, rel, lambda_param, self.similarity_threshold, self.similarity_metric,
                                   min(max_results, len(valid)), self.device)
        return [valid[i] for i in picked]

    # ---------------------------------------------------------------- interface
    async def diversify(self, results: List[Union[Any, Dict[str, Any]]], lambda_param: Optional[float] = None,
                        max_results: Optional[int] = None) -> List[Union[Any, Dict[str, Any]]]:
        return self.diversify_sync(results, lambda_param, max_results)

    def diversify_sync(self, results, lambda_param: Optional[float] = None, max_results: Optional[int] = None):
        if not results:
            return results
        lam = lambda_param if lambda_param is not None else self.lambda_param
        max_res = max_results if max_results is not None else self.max_results
        max_res = len(results) if max_res is None else min(max_res, len(results))
        try:
            idx = self.select_indices(results, lam, max_res)
            return [results[i] for i in idx]
        except _lib.RagHipError:
            raise           # a missing / failing GPU library is never papered over
        except Exception as e:      # vector_db/mmr_diversifier.py:117-120
            logger.error(f"MMR diversification failed: {e}")
            return results[:max_res]

    def compute_similarity_matrix(self, embeddings: List[np.ndarray]) -> np.ndarray:
        """vector_db/mmr_diversifier.py:291-319: the n x n similarity matrix of the stacked embeddings under
        this diversifier's metric (an unknown metric counts as cosine); no embeddings -> an empty array."""
        if not embeddings:
            return np.array([])
        return similarity_matrix_device(np.vstack(embeddings), self.similarity_metric, self.device)

    def analyze_diversity(self, results: List[Union[Any, Dict[str, Any]]]) -> Dict[str, Any]:
        """vector_db/mmr_diversifier.py:321-366: counts of a result set -- results, distinct documents and
        their ratio, mean and variance of the text lengths.  Host bookkeeping; errors come back as
        ``{"error": ...}`` like the reference's."""
        try:
            if not results:
                return {"error": "No results provided"}
            texts = []
            for r in results:
                if isinstance(r, dict):
                    texts.append(r.get("text", ""))
                elif hasattr(r, "text"):
                    texts.append(r.text)
            unique_documents = len(set(r.get("document_id") if isinstance(r, dict) else r.document_id for r in results))
            lengths = [len(t) for t in texts]
            return {
                "total_results": len(results),
                "unique_documents": unique_documents,
                "document_diversity_ratio": unique_documents / len(results),
                "avg_text_length": np.mean(lengths) if texts else 0,
                "text_length_variance": np.var(lengths) if texts else 0,
            }
        except Exception as e:
            logger.error(f"Diversity analysis failed: {e}")
            return {"error": str(e)}

    async def __aenter__(self):
        return self

    async def __aexit__(self, exc_type, exc_val, exc_tb):
        return None


def create_relevance_focused_mmr(similarity_threshold: float = 0.9) -> MMRDiversifier:
    return MMRDiversifier(lambda_param=0.8, similarity_threshold=similarity_threshold)


def create_diversity_focused_mmr(similarity_threshold: float = 0.7) -> MMRDiversifier:
    return MMRDiversifier(lambda_param=0.3, similarity_threshold=similarity_threshold)


def create_balanced_mmr(similarity_threshold: float = 0.8) -> MMRDiversifier:
    return MMRDiversifier(lambda_param=0.6, similarity_threshold=similarity_threshold)


_default_mmr_diversifier: Optional[MMRDiversifier] = None


def get_default_mmr_diversifier() -> MMRDiversifier:
    global _default_mmr_diversifier
    if _default_mmr_diversifier is None:
        _default_mmr_diversifier = create_balanced_mmr()
    return _default_mmr_diversifier


def set_default_mmr_config(lambda_param: float = 0.6, similarity_threshold: float = 0.8,
                           max_results: Optional[int] = None, similarity_metric: str = "cosine") -> None:
    global _default_mmr_diversifier
    _default_mmr_diversifier = MMRDiversifier(lambda_param=lambda_param, similarity_threshold=similarity_threshold,
                                              max_results=max_results, similarity_metric=similarity_metric)
