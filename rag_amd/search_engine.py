"""The vector-side methods of the reference's ``EnhancedSearchEngine``
(vector_db/search_manager.py) over the GPU storage manager: multi-embedding
semantic search (:876-932), contextual semantic search (:995-1017), hybrid
semantic + BM25 fusion (:1023-1101 + :969-970) and MMR diversification
(:1620-1735).  Query embedding, SQL lookups, caching and access control stay with
the caller, as they sit above the hot path."""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Sequence, Tuple

from . import fusion
from .mmr_diversifier import MMRDiversifier
from .schemas import SearchFilters, SearchResult
from .storage_manager import VectorStorageManager, get_storage_manager

logger = logging.getLogger(__name__)


def _result_from_hit(hit: Dict[str, Any]) -> Optional[SearchResult]:
    md = hit.get("metadata", {})
    cid = md.get("chunk_id")
    if not cid:
        return None
    return SearchResult(document_id=md.get("document_id"), chunk_id=cid, text=hit.get("content", ""),
                        score=hit.get("score", 0.0), metadata=dict(md))


class HipSearchEngine:
    def __init__(self, storage_manager: VectorStorageManager = None):
        self.storage_manager = storage_manager or get_storage_manager()

    async def semantic_search(self, query_embeddings: Sequence[Tuple[str, Sequence[float], float]],
                              filters: SearchFilters, limit: int) -> List[SearchResult]:
        """search_manager.py:876-932 given the (strategy, vector, weight) list that
        :856-874 builds (full_query 0.4, concepts 0.6, expanded 0.3)."""
        try:
            per_doc, hits_by_key = [], {}
            doc_ids = list(filters.document_ids or [])
            # one device call for searches, weighting, best-per-chunk and ranking when the result keys
            # "{document}_{chunk}" cannot collide across documents
            keys_safe = len(set(doc_ids)) == len(doc_ids) and (len(doc_ids) < 2 or not any("_" in str(d) for d in doc_ids))
            fused = getattr(self.storage_manager, "search_vectors_fused", None)
            if keys_safe and fused is not None and query_embeddings:
                ranked = await fused(
                    [f"doc_{doc_id}" for doc_id in doc_ids], [vector for _name, vector, _weight in query_embeddings],
                    [float(weight) for _name, _vector, weight in query_embeddings], vector_type="content",
                    per_limit=limit * 2, limit=limit, score_threshold=filters.min_score or 0.1,
                    metadata_filters={"user_id": filters.user_id} if filters.user_id else None)
                if ranked is not None:
                    out = []
                    for _ni, hit, si in ranked:
                        res = _result_from_hit(hit)
                        if res is not None:
                            res.metadata["search_strategy"] = query_embeddings[si][0]
                            out.append(res)
                    return out
            # the documents x strategies fan-out as one call (one device pass on a pooled manager)
            all_hits = await self.storage_manager.search_vectors_many(
                [f"doc_{doc_id}" for doc_id in doc_ids], [vector for _name, vector, _weight in query_embeddings],
                vector_type="content", limit=limit * 2, score_threshold=filters.min_score or 0.1,
                metadata_filters={"user_id": filters.user_id} if filters.user_id else None)
            for di, doc_id in enumerate(doc_ids):
                strategies = []
                for qi, (name, _vector, weight) in enumerate(query_embeddings):
                    pairs = []
                    for h in all_hits[qi][di]:
                        cid = h.get("metadata", {}).get("chunk_id")
                        pairs.append((cid, h.get("score", 0.0)))
                        if cid:
                            hits_by_key[(doc_id, cid, name)] = h
                    strategies.append((name, weight, pairs))
                per_doc.append((doc_id, strategies))
            ranked = fusion.multi_embedding_fuse(per_doc, limit)
            owner = {f"{doc_id}_{cid}": (doc_id, cid) for (doc_id, cid, _name) in hits_by_key}
            out = []
            for key, score, strategy in ranked:
                doc_id, cid = owner[key]
                if (doc_id, cid, strategy) not in hits_by_key:      # same key string from another document
                    doc_id, cid = next((d, c) for (d, c, s) in hits_by_key
                                       if s == strategy and f"{d}_{c}" == key)
                res = _result_from_hit(hits_by_key[(doc_id, cid, strategy)])
                if res is None:
                    continue
                res.score = score
                res.metadata["search_strategy"] = strategy
                out.append(res)
            return out
        except Exception as e:
            logger.error(f"Multi-embedding semantic search failed: {e}")
            return []

    async def contextual_semantic(self, query_vector: Sequence[float], filters: SearchFilters, limit: int
                                  ) -> List[SearchResult]:
        """search_manager.py:995-1017."""
        out = []
        doc_ids = list(filters.document_ids or [])
        per_doc = await self.storage_manager.contextual_search_many(
            [f"doc_{doc_id}" for doc_id in doc_ids], query_vector, limit=limit * 2, content_weight=0.7,
            context_weight=0.3, score_threshold=filters.min_score * 0.5 if filters.min_score else None)
        for hits in per_doc:
            for h in hits:
                res = _result_from_hit(h)
                if res is not None:
                    res.metadata["content_score"] = h.get("content_score")
                    res.metadata["context_score"] = h.get("context_score")
                    out.append(res)
        return out

    def combine_semantic_and_bm25_contextual(self, semantic_results: List[SearchResult],
                                             bm25_results: List[SearchResult], limit: Optional[int] = None
                                             ) -> List[SearchResult]:
        """search_manager.py:1023-1101 and the sort + cut of :969-970 (best first)."""
        sem_by = {f"{r.document_id}_{r.chunk_id}": r for r in semantic_results}
        bm_by = {f"{r.document_id}_{r.chunk_id}": r for r in bm25_results}
        n_keys = len(set(sem_by) | set(bm_by))
        fused = fusion.hybrid_fuse([(f"{r.document_id}_{r.chunk_id}", r.score) for r in semantic_results],
                                   [(f"{r.document_id}_{r.chunk_id}", r.score) for r in bm25_results],
                                   n_keys if limit is None else limit)
        if fused is None:            # zero maximum: the reference logs and returns the semantic list (:1098-1101)
            logger.error("Score combination failed: float division by zero")
            return semantic_results
        out = []
        for key, combined, sn, bn, has_sem, has_bm in fused:
            base = sem_by[key] if has_sem else bm_by[key]
            out.append(SearchResult(
                chunk_id=base.chunk_id, document_id=base.document_id, text=base.text, score=combined,
                metadata={**base.metadata, "search_algorithm": "enhanced_contextual", "semantic_score": sn,
                          "bm25_score": bn, "has_semantic": has_sem, "has_bm25": has_bm},
                document_metadata=base.document_metadata, highlight=base.highlight))
        return out

    async def ensure_embeddings_for_mmr(self, results: List[SearchResult]) -> List[SearchResult]:
        """search_manager.py:1683-1735: results without a retrievable vector are dropped."""
        kept = []
        for r in results:
            emb = (r.metadata or {}).get("embedding") or (r.metadata or {}).get("vector")
            if emb is None:
                data = await self.storage_manager.get_vector_by_chunk_id(r.chunk_id)
                if data and "vector" in data:
                    emb = data["vector"]
            if emb:
                r.metadata = r.metadata or {}
                r.metadata["embedding"] = emb
                kept.append(r)
        return kept

    async def apply_mmr_diversification(self, results: List[SearchResult], filters: SearchFilters
                                        ) -> List[SearchResult]:
        """search_manager.py:1620-1681."""
        if not results:
            return results
        try:
            mmr = MMRDiversifier(lambda_param=filters.mmr_lambda,
                                 similarity_threshold=filters.mmr_similarity_threshold,
                                 max_results=filters.mmr_max_results,
                                 similarity_metric=filters.mmr_similarity_metric,
                                 device=self.storage_manager.device)
            with_emb = await self.ensure_embeddings_for_mmr(results)
            if not with_emb:
                return results
            out = await mmr.diversify(with_emb, lambda_param=filters.mmr_lambda, max_results=filters.mmr_max_results)
            for i, r in enumerate(out):
                r.metadata = r.metadata or {}
                r.metadata.update({"mmr_applied": True, "mmr_lambda": filters.mmr_lambda,
                                   "mmr_similarity_threshold": filters.mmr_similarity_threshold,
                                   "mmr_rank": i + 1, "mmr_similarity_metric": filters.mmr_similarity_metric})
            return out
        except Exception as e:
            from ._lib import RagHipError
            if isinstance(e, RagHipError):
                raise
            logger.error(f"MMR diversification failed, returning original results: {e}")
            return results
