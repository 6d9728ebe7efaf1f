"""``VectorStorageManager`` with the public surface of the reference
(vector_db/storage_manager.py:19-1016) and the vectors on the GPU.

Every index is a pair of collections, ``{index}_content`` and ``{index}_context``,
each served by a ``HipSearchOptimizer`` registered in ``search_optimizers`` under
those keys -- the seam the reference's ``EnhancedSearchEngine`` calls through
(vector_db/search_manager.py:887,1002,1710).  The manager keeps the reference's
error convention: it logs and answers ``[]`` / ``False`` / ``None`` instead of
raising (storage_manager.py:509-511,600-602,714-716).  No Qdrant, no SQL: the
``db`` / ``validate_chunks`` arguments are accepted and ignored.
"""
from __future__ import annotations

import logging
import uuid
from datetime import datetime, timezone
from typing import Any, Dict, List, Optional, Tuple

from . import fusion
from .schemas import MetricType, SearchConfig, device_limit
from .pooled import PooledCollections, PooledView
from .search_optimizer import HipSearchOptimizer

logger = logging.getLogger(__name__)


class VectorStorageManager:
    def __init__(self, device: int = 0, pooled: bool = False):
        """``pooled=True`` keeps every collection of one vector type and dimension as
        row ranges of one device corpus, so that ``search_vectors_many`` answers a
        whole per-document fan-out in one device pass; the objects registered in
        ``search_optimizers`` are then views of those pools."""
        self.device = device
        self.pooled = pooled
        self._pools: Dict[Tuple[str, int], PooledCollections] = {}
        self.search_optimizers: Dict[str, HipSearchOptimizer] = {}
        self.content_collection_suffix = "_content"
        self.context_collection_suffix = "_context"
        self._is_initialized = False
        self._collections_state: Dict[str, Dict[str, Any]] = {}

    async def initialize(self) -> bool:
        self._is_initialized = True
        return True

    def _get_collection_names(self, index_name: str) -> Tuple[str, str]:
        return (f"{index_name}{self.content_collection_suffix}", f"{index_name}{self.context_collection_suffix}")

    # ------------------------------------------------------------------- create
    async def create_index(self, index_name: str, embedding_dimension: int = 384, user_id: int = None,
                           document_id: int = None, db=None) -> bool:
        """storage_manager.py:162-252: two COSINE collections and their optimizers."""
        try:
            content_collection, context_collection = self._get_collection_names(index_name)
            for key, name, vtype in ((f"{index_name}_content", content_collection, "content"),
                                     (f"{index_name}_context", context_collection, "context")):
                if key in self.search_optimizers:
                    continue
                cfg = SearchConfig(dimension=embedding_dimension, metric=MetricType.COSINE,
                                   collection_name=name, device=self.device)
                if self.pooled:
                    pool = self._pools.get((vtype, embedding_dimension))
                    if pool is None:
                        pool = self._pools[(vtype, embedding_dimension)] = PooledCollections(
                            embedding_dimension, metric="cosine", device=self.device)
                    self.search_optimizers[key] = PooledView(pool, cfg)
                else:
                    self.search_optimizers[key] = HipSearchOptimizer(config=cfg)
            self._collections_state[index_name] = {"content": content_collection, "context": context_collection,
                                                   "dimension": embedding_dimension, "user_id": user_id,
                                                   "document_id": document_id}
            self._is_initialized = True
            return True
        except Exception as e:
            logger.error(f"Failed to create vector index {index_name}: {e}")
            return False

    async def add_vectors(self, index_name: str, content_vectors: List[List[float]],
                          context_vectors: List[List[float]], metadata_list: List[Dict[str, Any]],
                          chunk_ids: List[str] = None, validate_chunks: bool = True) -> List[str]:
        """storage_manager.py:254-366."""
        try:
            if len(content_vectors) != len(context_vectors) or len(content_vectors) != len(metadata_list):
                raise ValueError("All input lists must have the same length")
            if chunk_ids is None:
                chunk_ids = [str(uuid.uuid4()) for _ in range(len(content_vectors))]
            content = self.search_optimizers.get(f"{index_name}_content")
            context = self.search_optimizers.get(f"{index_name}_context")
            if content is None or context is None:
                raise ValueError(f"index '{index_name}' does not exist")
            now = datetime.now(timezone.utc).isoformat()
            for opt, vectors, vtype in ((content, content_vectors, "content"), (context, context_vectors, "context")):
                payloads = []
                for i, metadata in enumerate(metadata_list):
                    enhanced = dict(metadata)
                    enhanced.update({"chunk_id": chunk_ids[i], "index_name": index_name,
                                     "vector_type": vtype, "added_at": now})
                    payloads.append(enhanced)
                opt.upsert(vectors, payloads, chunk_ids)
            return list(chunk_ids)
        except Exception as e:
            logger.error(f"Failed to add vectors to index {index_name}: {e}")
            return []

    # ------------------------------------------------------------------- search
    async def search_vectors(self, index_name: str, query_vector: List[float], vector_type: str = "content",
                             limit: int = 10, score_threshold: float = None,
                             metadata_filters: Dict[str, Any] = None, use_faiss: bool = None
                             ) -> List[Dict[str, Any]]:
        """storage_manager.py:427-511."""
        try:
            optimizer_key = f"{index_name}_{vector_type}"
            optimizer = self.search_optimizers.get(optimizer_key)
            if not optimizer:
                available = list(self.search_optimizers.keys())
                if vector_type == "context":
                    if f"{index_name}_content" in self.search_optimizers:
                        logger.warning(f"Context vectors not available for index '{index_name}'.")
                        return []
                    raise ValueError(f"Neither content nor context vectors found for index '{index_name}'. "
                                     f"Available optimizers: {available}")
                raise ValueError(f"Search optimizer '{optimizer_key}' not found. Available: {available}")
            results = await optimizer.search(query_vector=query_vector, limit=limit,
                                             min_score=score_threshold or 0.0, filters=metadata_filters)
            return self._format_hits(results)
        except Exception as e:
            logger.error(f"Failed to search vectors in index {index_name}: {e}")
            return []

    @staticmethod
    def _format_hits(results) -> List[Dict[str, Any]]:
        """storage_manager.py:477-505: (Chunk, score) tuples as result dictionaries."""
        formatted = []
        for chunk, score in results:
            metadata = {"chunk_id": getattr(chunk, "chunk_id", "unknown"),
                        "document_id": getattr(chunk, "document_id", None),
                        "start_char": getattr(chunk, "start_idx", None),
                        "end_char": getattr(chunk, "end_idx", None)}
            if getattr(chunk, "metadata", None):
                metadata.update(chunk.metadata)
            formatted.append({"score": float(score), "content": chunk.text or "", "metadata": metadata,
                              "source": "hip"})
        return formatted

    async def search_vectors_many(self, index_names: List[str], query_vectors: List[List[float]],
                                  vector_type: str = "content", limit: int = 10, score_threshold: float = None,
                                  metadata_filters: Dict[str, Any] = None) -> List[List[List[Dict[str, Any]]]]:
        """``out[qi][ni]`` = what ``search_vectors(index_names[ni], query_vectors[qi], ...)``
        returns, for every pair at once: the loop of vector_db/search_manager.py:879-913
        (documents x query strategies, one RPC each in the reference) as one device
        pass per pool (pooled manager) or one batched device call per index."""
        out = [[[] for _ in index_names] for _ in query_vectors]
        try:
            if not index_names or not len(query_vectors):
                return out
            min_score = score_threshold or 0.0
            by_pool: Dict[int, Tuple[PooledCollections, List[Tuple[int, PooledView]]]] = {}
            for ni, index_name in enumerate(index_names):
                optimizer = self.search_optimizers.get(f"{index_name}_{vector_type}")
                if not optimizer:
                    logger.error(f"Failed to search vectors in index {index_name}: "
                                 f"Search optimizer '{index_name}_{vector_type}' not found")
                    continue
                if isinstance(optimizer, PooledView):
                    if not getattr(optimizer, "_disabled", False):
                        by_pool.setdefault(id(optimizer.pool), (optimizer.pool, []))[1].append((ni, optimizer))
                    continue
                try:
                    lists = await optimizer.batch_search(query_vectors, limit=limit, min_score=min_score,
                                                         filters=metadata_filters)
                    for qi, results in enumerate(lists):
                        out[qi][ni] = self._format_hits(results)
                except Exception as e:
                    logger.error(f"Failed to search vectors in index {index_name}: {e}")
            for pool, members in by_pool.values():
                if limit < 1:
                    continue
                names = [view.collection_name for _ni, view in members]
                try:
                    hits = pool.search_many(names, query_vectors, device_limit(limit), min_score=min_score,
                                            row_mask=pool.filter_mask(names, metadata_filters))
                    for qi in range(len(hits)):
                        for (ni, _view), h in zip(members, hits[qi]):
                            out[qi][ni] = pool.hit_dicts(h)
                except Exception as e:
                    logger.error(f"Failed to search vectors in indexes {names[:3]}...: {e}")
            return out
        except Exception as e:
            logger.error(f"Failed to search vectors in {len(index_names)} indexes: {e}")
            return out

    async def search_vectors_fused(self, index_names: List[str], query_vectors: List[List[float]], weights: List[float],
                                   vector_type: str = "content", per_limit: int = 20, limit: int = 10,
                                   score_threshold: float = None, metadata_filters: Dict[str, Any] = None
                                   ) -> Optional[List[Tuple[int, Dict[str, Any], int]]]:
        """vector_db/search_manager.py:879-932 in one device call when every index sits in
        the same pool: the searches of ``search_vectors_many``, each hit's score times the
        weight of its query vector, the best per chunk, all documents ranked together.
        Returns [(position in index_names, result dictionary with the weighted score,
        position of the winning query vector)] best first, at most ``limit``; None when
        the device call does not apply (indexes outside a pool or in several pools,
        chunk ids repeated inside a collection, more candidates than the kernel takes)."""
        try:
            if not index_names or not len(query_vectors) or limit < 1 or per_limit < 1:
                return None
            pool, names, where = None, [], []
            for ni, index_name in enumerate(index_names):
                optimizer = self.search_optimizers.get(f"{index_name}_{vector_type}")
                if not optimizer:
                    logger.error(f"Failed to search vectors in index {index_name}: "
                                 f"Search optimizer '{index_name}_{vector_type}' not found")
                    continue
                if not isinstance(optimizer, PooledView) or (pool is not None and optimizer.pool is not pool):
                    return None
                pool = optimizer.pool
                if getattr(optimizer, "_disabled", False):
                    continue
                if not pool.chunk_ids_distinct(optimizer.collection_name):
                    return None
                names.append(optimizer.collection_name)
                where.append(ni)
            if pool is None or len(set(names)) != len(names):
                return None
            if not names:
                return []
            ranked = pool.fused_search(names, query_vectors, weights, device_limit(per_limit), limit,
                                       min_score=score_threshold or 0.0, row_mask=pool.filter_mask(names, metadata_filters))
            if ranked is None:
                return None
            out = []
            for (ci, row, score, si), hit in zip(ranked, pool.hit_dicts([(row, score) for _ci, row, score, _si in ranked])):
                out.append((where[ci], hit, si))
            return out
        except Exception as e:
            logger.error(f"Fused search over {len(index_names)} indexes failed: {e}")
            return None

    async def get_vector_by_chunk_id(self, chunk_id: str) -> Optional[Dict[str, Any]]:
        """storage_manager.py:558-602: first collection (content before context, in
        registration order) that holds the chunk.  A dictionary lookup per
        collection instead of a retrieve RPC per collection."""
        try:
            for optimizer_key in list(self.search_optimizers.keys()):
                if not optimizer_key.endswith("_content"):
                    continue
                index_name = optimizer_key[:-8]
                content_collection, context_collection = self._get_collection_names(index_name)
                for key, collection in ((f"{index_name}_content", content_collection),
                                        (f"{index_name}_context", context_collection)):
                    opt = self.search_optimizers.get(key)
                    if opt is None:
                        continue
                    data = opt.get_vector(chunk_id)
                    if data and "vector" in data:
                        return {"vector": data["vector"], "source": optimizer_key, "collection": collection}
            return None
        except Exception as e:
            logger.error(f"Error getting vector for chunk_id {chunk_id}: {e}")
            return None

    async def contextual_search(self, index_name: str, query_vector: List[float], limit: int = 10,
                                content_weight: float = 0.7, context_weight: float = 0.3,
                                score_threshold: float = None, metadata_filters: Dict[str, Any] = None
                                ) -> List[Dict[str, Any]]:
        """storage_manager.py:604-716: both collections asked for 2*limit hits, fused
        on chunk id on the GPU (rag_fuse_weighted)."""
        try:
            content_results = await self.search_vectors(index_name=index_name, query_vector=query_vector,
                                                        vector_type="content", limit=limit * 2,
                                                        metadata_filters=metadata_filters)
            context_results = await self.search_vectors(index_name=index_name, query_vector=query_vector,
                                                        vector_type="context", limit=limit * 2,
                                                        metadata_filters=metadata_filters)
            return self._fuse_contextual(content_results, context_results, limit, content_weight, context_weight,
                                         score_threshold)
        except Exception as e:
            logger.error(f"Failed to perform contextual search in index {index_name}: {e}", exc_info=True)
            return []

    @staticmethod
    def _fuse_contextual(content_results, context_results, limit, content_weight, context_weight, score_threshold):
        """storage_manager.py:653-709: union on chunk id, weighted sum, cut."""
        by_id: Dict[str, Dict[str, Any]] = {}
        for r in content_results:
            cid = r["metadata"].get("chunk_id")
            if cid:
                by_id[cid] = r
        for r in context_results:
            cid = r["metadata"].get("chunk_id")
            if cid and cid not in by_id:
                by_id[cid] = r
        fused = fusion.contextual_fuse([(r["metadata"].get("chunk_id"), r["score"]) for r in content_results],
                                       [(r["metadata"].get("chunk_id"), r["score"]) for r in context_results],
                                       limit, content_weight, context_weight, score_threshold)
        out = []
        for cid, combined, cs, xs, _src in fused:
            result = by_id[cid].copy()
            result["score"] = combined
            result["content_score"] = cs
            result["context_score"] = xs
            out.append(result)
        return out

    async def contextual_search_many(self, index_names: List[str], query_vector: List[float], limit: int = 10,
                                     content_weight: float = 0.7, context_weight: float = 0.3,
                                     score_threshold: float = None, metadata_filters: Dict[str, Any] = None
                                     ) -> List[List[Dict[str, Any]]]:
        """``out[ni]`` = ``contextual_search(index_names[ni], query_vector, ...)`` for every
        index: the per-document loop of vector_db/search_manager.py:995-1017 with the
        content searches of all documents in one device pass and the context
        searches in another."""
        try:
            content = await self.search_vectors_many(index_names, [query_vector], "content", limit * 2, None,
                                                     metadata_filters)
            present = [n for n in index_names if f"{n}_context" in self.search_optimizers]
            context = await self.search_vectors_many(present, [query_vector], "context", limit * 2, None,
                                                     metadata_filters)
            ctx_of = dict(zip(present, context[0])) if present else {}
            return [self._fuse_contextual(content[0][ni], ctx_of.get(name, []), limit, content_weight,
                                          context_weight, score_threshold)
                    for ni, name in enumerate(index_names)]
        except Exception as e:
            logger.error(f"Failed to perform contextual search in {len(index_names)} indexes: {e}", exc_info=True)
            return [[] for _ in index_names]

    # --------------------------------------------------------- delete / restore
    async def delete_index(self, index_name: str, db=None) -> bool:
        try:
            for key in (f"{index_name}_content", f"{index_name}_context"):
                opt = self.search_optimizers.pop(key, None)
                if opt is not None:
                    await opt.delete_collection()
            self._collections_state.pop(index_name, None)
            # a pooled corpus only hides the rows of a deleted collection: rebuild it once more
            # than half of it (and at least 4096 rows) is dead
            for pool in self._pools.values():
                dead = pool.dead_rows()
                if dead >= 4096 and 2 * dead > len(pool.payloads):
                    pool.compact()
            return True
        except Exception as e:
            logger.error(f"Failed to delete index {index_name}: {e}")
            return False

    async def soft_delete_index(self, index_name: str, db=None) -> bool:
        """storage_manager.py:752-792: keep the data, hide every vector."""
        try:
            found = False
            for key in (f"{index_name}_content", f"{index_name}_context"):
                opt = self.search_optimizers.get(key)
                if opt is not None:
                    opt.set_points_active(False)
                    opt._disabled = True
                    found = True
            return found
        except Exception as e:
            logger.error(f"Failed to soft delete index {index_name}: {e}")
            return False

    async def restore_index(self, index_name: str, db=None) -> bool:
        try:
            found = False
            for key in (f"{index_name}_content", f"{index_name}_context"):
                opt = self.search_optimizers.get(key)
                if opt is not None:
                    opt.set_points_active(True)
                    if hasattr(opt, "_disabled"):
                        delattr(opt, "_disabled")
                    found = True
            return found
        except Exception as e:
            logger.error(f"Failed to restore index {index_name}: {e}")
            return False

    # --------------------------------------------------------------------- info
    async def get_index_stats(self, index_name: str) -> Dict[str, Any]:
        try:
            stats = {"index_name": index_name, "timestamp": datetime.now(timezone.utc).isoformat(), "type": "hip"}
            cols = {}
            for vtype in ("content", "context"):
                opt = self.search_optimizers.get(f"{index_name}_{vtype}")
                cols[vtype] = await opt.get_collection_info() if opt else None
            stats["collections"] = cols
            return stats
        except Exception as e:
            logger.error(f"Failed to get stats for index {index_name}: {e}")
            return {"error": str(e)}

    async def get_performance_stats(self) -> Dict[str, Any]:
        from ._lib import device_count
        return {"storage_type": "hip", "search_optimizers_count": len(self.search_optimizers),
                "devices": device_count(), "collections": len(self.search_optimizers)}

    async def add_collection(self, index_name: str, embedding_dimension: int) -> bool:
        return await self.create_index(index_name, embedding_dimension)

    async def remove_collection(self, index_name: str) -> bool:
        return await self.delete_index(index_name)

    async def refresh_collections(self, force: bool = False) -> bool:
        return True

    async def cleanup(self):
        for name in list(self._collections_state.keys()):
            await self.delete_index(name)
        self.search_optimizers.clear()
        self._collections_state.clear()
        for pool in self._pools.values():
            pool.close()
        self._pools.clear()
        self._is_initialized = False


_storage_manager: Optional[VectorStorageManager] = None


def get_storage_manager() -> VectorStorageManager:
    global _storage_manager
    if _storage_manager is None:
        _storage_manager = VectorStorageManager()
    return _storage_manager


async def get_initialized_storage_manager() -> VectorStorageManager:
    manager = get_storage_manager()
    if not manager._is_initialized:
        await manager.initialize()
    return manager


async def init_storage_manager() -> VectorStorageManager:
    return await get_initialized_storage_manager()
