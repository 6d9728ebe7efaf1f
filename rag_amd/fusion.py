"""The reference's score fusions over (key, score) pairs, arithmetic on the GPU.

Key bookkeeping (which chunk appeared on which side, dict insertion order,
"later entry wins") is host logic and mirrors the reference line by line; the
scores themselves -- weighting, normalising, thresholding, the stable descending
sort and the cut -- are computed by ``rag_fuse_weighted`` / ``rag_fuse_maxnorm``
(rag_amd/csrc/fusion_mmr.hip) in fp64 with the reference's operation order.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import check, lib

FUSE_MAX = 4096


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fuse_weighted(a: Sequence[float], b: Sequence[float], wa: float, wb: float,
                  threshold: Optional[float], k: int):
    """score[i] = a[i]*wa + b[i]*wb; returns (order of kept candidates by score
    descending / stable / first k, all scores)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = a.size
    if b.size != n:
        raise ValueError("a and b must have the same length")
    if n > FUSE_MAX:
        raise ValueError(f"at most {FUSE_MAX} candidates")
    order = np.empty(max(n, 1), dtype=np.int32)
    score = np.empty(max(n, 1), dtype=np.float64)
    n_out = C.c_int32(0)
    check(lib.rag_fuse_weighted(n, _ptr(a), _ptr(b), float(wa), float(wb),
                                0 if threshold is None else 1,
                                0.0 if threshold is None else float(threshold), int(k),
                                _ptr(order), _ptr(score), C.byref(n_out)))
    return order[:n_out.value].tolist(), score[:n]


def fuse_maxnorm(a, b, has_a, has_b, max_a: float, max_b: float, wa: float, wb: float, k: int):
    """score[i] = wa*(a[i]/max_a if has_a[i] else 0) + wb*(b[i]/max_b if has_b[i] else 0)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    ha = np.ascontiguousarray(has_a, dtype=np.uint8)
    hb = np.ascontiguousarray(has_b, dtype=np.uint8)
    n = a.size
    if n > FUSE_MAX:
        raise ValueError(f"at most {FUSE_MAX} candidates")
    order = np.empty(max(n, 1), dtype=np.int32)
    score = np.empty(max(n, 1), dtype=np.float64)
    sa = np.empty(max(n, 1), dtype=np.float64)
    sb = np.empty(max(n, 1), dtype=np.float64)
    n_out = C.c_int32(0)
    check(lib.rag_fuse_maxnorm(n, _ptr(a), _ptr(b), _ptr(ha), _ptr(hb), float(max_a), float(max_b),
                               float(wa), float(wb), int(k), _ptr(order), _ptr(score), _ptr(sa), _ptr(sb),
                               C.byref(n_out)))
    return order[:n_out.value].tolist(), score[:n], sa[:n], sb[:n]


def contextual_fuse(content: Sequence[Tuple[str, float]], context: Sequence[Tuple[str, float]],
                    limit: int, content_weight: float = 0.7, context_weight: float = 0.3,
                    score_threshold: Optional[float] = None):
    """vector_db/storage_manager.py:653-709 on (chunk_id, score) hit lists.
    Returns [(chunk_id, combined, content_score, context_score, source)]."""
    comb: Dict[str, list] = {}
    for cid, s in content:                       # :658-667
        if cid:
            comb[cid] = [s, 0.0, "content"]
    for cid, s in context:                       # :670-682
        if cid:
            if cid in comb:
                comb[cid][1] = s
            else:
                comb[cid] = [0.0, s, "context"]
    if not comb:
        return []
    keys = list(comb.keys())
    order, score = fuse_weighted([comb[c][0] for c in keys], [comb[c][1] for c in keys],
                                 content_weight, context_weight, score_threshold, limit)
    return [(keys[i], float(score[i]), comb[keys[i]][0], comb[keys[i]][1], comb[keys[i]][2]) for i in order]


def hybrid_fuse(semantic: Sequence[Tuple[str, float]], bm25: Sequence[Tuple[str, float]], limit: int,
                semantic_weight: float = 0.65, bm25_weight: float = 0.35):
    """vector_db/search_manager.py:1023-1101 followed by the caller's stable sort
    and cut (:969-970).  Returns [(key, combined, sem_n, bm25_n, has_sem, has_bm25)]
    best first, or ``None`` where the reference falls back to the semantic list
    (a zero maximum, :1098-1101).  Keys are walked in order of first appearance
    (semantic, then bm25-only), which fixes the order of exactly tied scores the
    reference leaves to its string hash seed (:1049)."""
    sem_by, bm_by = {}, {}
    for key, s in semantic:
        sem_by[key] = s
    for key, s in bm25:
        bm_by[key] = s
    max_sem = max([s for _, s in semantic]) if semantic else 1.0      # :1045-1046
    max_bm = max([s for _, s in bm25]) if bm25 else 1.0
    keys = list(sem_by.keys()) + [key for key in bm_by if key not in sem_by]
    if not keys:
        return []
    uses_sem = bool(sem_by)
    uses_bm = bool(bm_by)
    if (uses_sem and max_sem == 0.0) or (uses_bm and max_bm == 0.0):
        return None
    order, score, sa, sb = fuse_maxnorm([sem_by.get(key, 0.0) for key in keys],
                                        [bm_by.get(key, 0.0) for key in keys],
                                        [key in sem_by for key in keys], [key in bm_by for key in keys],
                                        max_sem if uses_sem else 1.0, max_bm if uses_bm else 1.0,
                                        semantic_weight, bm25_weight, limit)
    return [(keys[i], float(score[i]), float(sa[i]), float(sb[i]), keys[i] in sem_by, keys[i] in bm_by)
            for i in order]


def rerank_blend(original: Sequence[float], rerank: Sequence[float], weight: float = 0.5,
                 min_rerank_score: Optional[float] = None, top_k: Optional[int] = None):
    """Score blend and ordering of the reranker stage:
    vector_db/reranker/base_reranker.py:256-280 (both scores clipped to [0, 1],
    ``(1-w)*orig + w*rerank``) and vector_db/reranker/huggingface_reranker.py:225-264
    (results under ``min_rerank_score`` dropped, stable sort by the blend descending,
    first ``top_k``; ``top_k`` of None or 0 keeps all).  The rerank scores themselves
    come from a cross-encoder, which is outside this library.
    Returns (indices into the input in result order, blend per input)."""
    o = np.clip(np.ascontiguousarray(original, dtype=np.float64), 0.0, 1.0)
    r_raw = np.ascontiguousarray(rerank, dtype=np.float64)
    if o.size != r_raw.size:
        raise ValueError("original and rerank must have the same length")
    keep = np.arange(o.size) if min_rerank_score is None else np.nonzero(~(r_raw < min_rerank_score))[0]
    if keep.size == 0:
        return [], np.zeros(o.size)
    r = np.clip(r_raw, 0.0, 1.0)
    order, score = fuse_weighted(o[keep], r[keep], 1 - weight, weight, None, top_k if top_k else keep.size)
    full = np.zeros(o.size)
    full[keep] = score
    return [int(keep[i]) for i in order], full


def multi_embedding_fuse(per_doc, limit: int):
    """vector_db/search_manager.py:877-932: per document the best ``score*weight``
    per chunk over the query strategies (strict ``>``: first strategy wins ties),
    across documents the best per ``"{doc}_{chunk}"``, stable sort descending,
    first ``limit``.  ``per_doc`` = [(doc_id, [(strategy, weight, [(chunk_id, score)...])...])...].
    The ranking (sort + cut) runs through ``rag_fuse_weighted`` with weights (1, 0)."""
    all_results: Dict[str, Tuple[float, str]] = {}
    for doc_id, strategies in per_doc:
        doc_results: Dict[str, Tuple[float, str]] = {}
        for name, weight, hits in strategies:
            for cid, s in hits:
                if cid:
                    w = s * weight
                    if cid not in doc_results or w > doc_results[cid][0]:
                        doc_results[cid] = (w, name)
        for cid, (w, name) in doc_results.items():
            key = f"{doc_id}_{cid}"
            if key not in all_results or w > all_results[key][0]:
                all_results[key] = (w, name)
    if not all_results:
        return []
    keys = list(all_results.keys())
    vals = [all_results[key][0] for key in keys]
    out = []
    for lo in range(0, len(keys), FUSE_MAX):      # more candidates than one launch takes: rank per slab, merge
        order, _ = fuse_weighted(vals[lo:lo + FUSE_MAX], [0.0] * len(vals[lo:lo + FUSE_MAX]), 1.0, 0.0, None, limit)
        out.extend(lo + i for i in order)
    if len(keys) > FUSE_MAX:
        sub = out
        order, _ = fuse_weighted([vals[i] for i in sub], [0.0] * len(sub), 1.0, 0.0, None, limit)
        out = [sub[i] for i in order]
    return [(keys[i], vals[i], all_results[keys[i]][1]) for i in out]
