"""CPU oracle: the reference's three score fusions, restated on plain
(key, score) pairs in Python floats (fp64).  TEST INFRASTRUCTURE ONLY.

Pinned by tests/golden/fusion_*.json, generated from the reference's own Python
by oracle/gen_golden.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple


def contextual_fuse(content: Sequence[Tuple[str, float]],
                    context: Sequence[Tuple[str, float]],
                    limit: int,
                    content_weight: float = 0.7,
                    context_weight: float = 0.3,
                    score_threshold: Optional[float] = None):
    """vector_db/storage_manager.py:653-709.

    ``content`` / ``context`` are the (chunk_id, score) hits of the two searches
    in returned order; falsy chunk ids are skipped (:660,:672).  Union on chunk
    id in dict insertion order (content hits first, then context-only hits); a
    repeated id inside one list keeps the later score (:661,:674); combined =
    content*cw + context*xw with the missing side 0.0 (:688-691); kept when
    ``threshold is None or combined >= threshold`` (:695); stable sort descending
    (:708); first ``limit`` (:709).
    Returns [(chunk_id, combined, content_score, context_score, source)] where
    source is 'content' or 'context' (whose result dict is carried, :664,:679).
    """
    comb: Dict[str, list] = {}
    for cid, s in content:
        if cid:
            comb[cid] = [s, 0.0, "content"]
    for cid, s in context:
        if cid:
            if cid in comb:
                comb[cid][1] = s
            else:
                comb[cid] = [0.0, s, "context"]
    out = []
    for cid, (cs, xs, src) in comb.items():
        c = cs * content_weight + xs * context_weight
        if score_threshold is None or c >= score_threshold:
            out.append((cid, c, cs, xs, src))
    out.sort(key=lambda t: t[1], reverse=True)
    return out[:limit]


def multi_embedding_fuse(per_doc: Sequence[Tuple[object, Sequence[Tuple[str, float, Sequence[Tuple[str, float]]]]]],
                         limit: int):
    """vector_db/search_manager.py:877-932.

    ``per_doc`` = [(doc_id, [(strategy_name, weight, [(chunk_id, score), ...]), ...]), ...]
    in the loop order of :879 and :884.  Per document keep, per chunk id, the
    entry with the largest ``score*weight`` under strict ``>`` so the first
    strategy wins ties (:902-909); across documents keep the larger per
    ``f"{doc}_{chunk}"`` again under strict ``>`` (:923-925); stable sort
    descending; first ``limit`` (:928-932).
    Returns [(key, weighted_score, strategy)].
    """
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
    final = [(k, w, n) for k, (w, n) in all_results.items()]
    final.sort(key=lambda t: t[1], reverse=True)
    return final[:limit]


def hybrid_fuse(semantic: Sequence[Tuple[str, float]],
                bm25: Sequence[Tuple[str, float]],
                semantic_weight: float = 0.65,
                bm25_weight: float = 0.35):
    """vector_db/search_manager.py:1023-1101.

    Both sides divided by their own maximum (an empty side divides by 1.0,
    :1045-1046); over the union of keys ``0.65*sem_n + 0.35*bm25_n`` (:1072-1075).
    A repeated key inside one list keeps the later entry (:1032-1039) while the
    maximum runs over every entry (:1042-1043).  A zero maximum raises
    ZeroDivisionError in the reference, which it catches and answers with the
    semantic list unchanged (:1098-1101): returned here as ``None``.

    The reference walks ``set(keys)`` (:1049), so the order of its output list,
    and with it the order of exactly tied combined scores after the caller's
    stable sort (:969), depends on the string hash seed.  This oracle fixes the
    order of first appearance (semantic keys, then bm25-only keys).
    Returns [(key, combined, sem_n, bm25_n, has_sem, has_bm25)] (unsorted) or None.
    """
    sem_by = {}
    for k, s in semantic:
        sem_by[k] = s
    bm_by = {}
    for k, s in bm25:
        bm_by[k] = s
    max_sem = max([s for _, s in semantic]) if semantic else 1.0
    max_bm = max([s for _, s in bm25]) if bm25 else 1.0
    keys = list(sem_by.keys()) + [k for k in bm_by.keys() if k not in sem_by]
    out = []
    try:
        for k in keys:
            sn = 0.0
            bn = 0.0
            if k in sem_by:
                sn = sem_by[k] / max_sem
            if k in bm_by:
                bn = bm_by[k] / max_bm
            out.append((k, semantic_weight * sn + bm25_weight * bn, sn, bn,
                        k in sem_by, k in bm_by))
    except ZeroDivisionError:
        return None
    return out


def hybrid_rank(semantic, bm25, limit, semantic_weight=0.65, bm25_weight=0.35):
    """hybrid_fuse followed by the caller's stable sort descending and cut
    (vector_db/search_manager.py:969-970)."""
    fused = hybrid_fuse(semantic, bm25, semantic_weight, bm25_weight)
    if fused is None:
        return None
    fused = sorted(fused, key=lambda t: t[1], reverse=True)
    return fused[:limit]


def rerank_blend(orig: float, rerank: float, weight: float) -> float:
    """vector_db/reranker/base_reranker.py:256-280: both scores clipped to [0,1],
    ``(1-w)*orig + w*rerank``."""
    o = max(0.0, min(1.0, orig))
    r = max(0.0, min(1.0, rerank))
    return (1 - weight) * o + weight * r
