"""``HipSearchOptimizer``: the per-collection search backend object of the
reference (``QdrantSearchOptimizer``, vector_db/search_optimizer.py:68-406) with
the vectors in HBM and the search in librag_hip.so.

It plugs into ``VectorStorageManager.search_optimizers`` under the keys
``f"{index}_content"`` / ``f"{index}_context"`` (vector_db/storage_manager.py:139,151)
and keeps the method set and the return types of the reference:
``search`` -> ``List[Tuple[Chunk, float]]`` best first, ``batch_search`` -> one such
list per query, ``SearchError`` on failure.  Where the reference sends one RPC
per query (:303-311) ``batch_search`` is ONE device call (the MFMA path).

Payloads (chunk id, document id, offsets, text, user fields) stay in a host table
indexed by row position; kernels see fp32 rows, row positions and a byte mask.
"""
from __future__ import annotations

import logging
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .filters import payload_passes
from .index import DeviceIndex
from .schemas import Chunk, IndexType, MetricType, SearchConfig, SearchError, device_limit

_METRIC = {MetricType.COSINE: "cosine", MetricType.DOT_PRODUCT: "ip", MetricType.EUCLIDEAN: "l2"}


class HipSearchOptimizer:
    def __init__(self, config: SearchConfig, qdrant_client=None):
        # `qdrant_client` is accepted for signature compatibility (search_optimizer.py:76) and unused
        if config.metric not in _METRIC:
            raise SearchError(f"metric {config.metric} is not supported by the HIP backend")
        self.config = config
        self.client = qdrant_client
        self.collection_name = config.collection_name
        self.logger = logging.getLogger(__name__)
        kind = "ivf" if config.index_type == IndexType.IVF else "flat"
        if config.metric == MetricType.EUCLIDEAN and kind != "flat":
            raise SearchError("the euclidean metric needs a flat index on the HIP backend")
        self.index = DeviceIndex(config.dimension, metric=_METRIC[config.metric], kind=kind, device=config.device)
        if config.quantization_config is not None and kind == "flat" and config.metric != MetricType.EUCLIDEAN:
            self.index.set_prefilter(True)
        self.payloads: List[Dict[str, Any]] = []       # row position -> payload
        self.point_ids: List[Any] = []                 # row position -> point id given at upsert
        self._row_of: Dict[Any, int] = {}              # point id -> live row position
        self._inactive = 0
        self._version = 0                              # bumped by every change of the payload table
        self._mask_cache: Dict[str, Tuple[int, np.ndarray]] = {}

    # ------------------------------------------------------------------ ingest
    async def create_collection(self) -> bool:
        return True

    def upsert(self, vectors, payloads: Sequence[Dict[str, Any]], ids: Sequence[Any]) -> List[Any]:
        """vector_db/qdrant_client.py:207-246 upsert_vectors: a point whose id
        exists already is replaced (its old row is hidden)."""
        x = np.ascontiguousarray(vectors, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.config.dimension:
            raise SearchError(f"expected vectors of dimension {self.config.dimension}")
        if not (len(payloads) == len(ids) == x.shape[0]):
            raise SearchError("vectors, payloads and ids must have the same length")
        if x.shape[0] == 0:
            return []
        first = len(self.payloads)
        self.index.add(x)
        stale = []
        for j, (p, i) in enumerate(zip(payloads, ids)):
            self.payloads.append(dict(p) if p else {})
            self.point_ids.append(i)
            if i in self._row_of:                 # an earlier point with this id, of this batch too: the last one wins
                stale.append(self._row_of[i])
            self._row_of[i] = first + j
        self._version += 1
        if stale:
            self.index.set_active(stale, False)
            self._inactive += len(set(stale))
        return list(ids)

    def get_vector(self, point_id) -> Optional[Dict[str, Any]]:
        """vector_db/qdrant_client.py retrieve(with_vectors=True) for one point."""
        row = self._row_of.get(point_id)
        if row is None:
            return None
        return {"id": point_id, "vector": self.index.reconstruct([row])[0].tolist(),
                "payload": self.payloads[row], "row": row}

    def row_of(self, point_id) -> Optional[int]:
        return self._row_of.get(point_id)

    def set_points_active(self, active: bool) -> None:
        """vector_db/qdrant_client.py:826-883: hide / restore every point (is_active)."""
        rows = sorted(self._row_of.values())
        if rows:
            self.index.set_active(rows, active)

    # ------------------------------------------------------------------ filters
    def _mask(self, filters: Optional[Dict[str, Any]]):
        """search_optimizer.py:173-201: every condition must hold (Filter(must=...));
        a list value adds one equality condition per element; {'range': {...}}
        is a numeric range; anything else is equality.  One byte per row; the
        mask of a filter is kept until the collection changes, so the per-user
        filter of every search (vector_db/search_manager.py:893) is evaluated once."""
        if not filters:
            return None
        key = repr(sorted(filters.items(), key=lambda kv: kv[0]))
        hit = self._mask_cache.get(key)
        if hit is not None and hit[0] == self._version:
            return hit[1]
        mask = np.fromiter((payload_passes(p, filters) for p in self.payloads), dtype=np.uint8,
                           count=len(self.payloads))
        if isinstance(self.index, DeviceIndex) and len(self.index) == mask.size:
            # the bytes stay on the device beside the rows (the server's payload index in the reference): a
            # repeated filter is not uploaded again
            mask = self.index.make_filter(mask)
        if len(self._mask_cache) >= 64:
            self._mask_cache.clear()
        self._mask_cache[key] = (self._version, mask)
        return mask

    def _to_chunks(self, D: np.ndarray, I: np.ndarray) -> List[Tuple[Chunk, float]]:
        out = []
        for score, row in zip(D, I):
            if row < 0:
                break
            payload = self.payloads[row] or {}
            chunk_id = payload.get("chunk_id") or str(self.point_ids[row])      # :235-241
            out.append((Chunk(document_id=payload.get("document_id", 0), chunk_id=chunk_id,
                              start_idx=payload.get("start_char", 0), end_idx=payload.get("end_char", 0),
                              text=payload.get("text", ""), metadata=payload, embedding=None), float(score)))
        return out

    # ------------------------------------------------------------------- search
    def search_sync(self, query_vector, limit: int = 10, min_score: float = 0.0,
                    filters: Dict[str, Any] = None, search_params: Dict[str, Any] = None):
        return self.batch_search_sync([query_vector], limit, min_score, filters, search_params)[0]

    def batch_search_sync(self, query_vectors, limit: int = 10, min_score: float = 0.0,
                          filters: Dict[str, Any] = None, search_params: Dict[str, Any] = None):
        try:
            if getattr(self, "_disabled", False):       # storage_manager.py:776-785 soft delete flag
                return [[] for _ in query_vectors]
            q = np.ascontiguousarray(query_vectors, dtype=np.float32)
            if q.ndim != 2 or q.shape[1] != self.config.dimension:
                raise ValueError(f"query vectors must have dimension {self.config.dimension}")
            if limit < 1:
                return [[] for _ in range(q.shape[0])]
            nprobe = (search_params or {}).get("nprobe", self.config.nprobe)
            if self.config.metric == MetricType.EUCLIDEAN:
                # Distance.EUCLID (search_optimizer.py:104-105): the score is the distance, smallest first, and
                # `score_threshold` (:217, fed with min_score) is the LARGEST distance returned
                if min_score is not None and min_score < 0:
                    return [[] for _ in range(q.shape[0])]
                D, I = self.index.search(q, device_limit(limit), min_score=None if min_score is None else float(min_score) ** 2,
                                         row_mask=self._mask(filters))
                D = np.sqrt(np.where(I >= 0, D, 0.0))
            else:
                D, I = self.index.search(q, device_limit(limit), min_score=min_score, nprobe=nprobe,
                                         row_mask=self._mask(filters))
            return [self._to_chunks(D[i], I[i]) for i in range(q.shape[0])]
        except Exception as e:
            self.logger.error(f"Search operation failed: {e}")
            raise SearchError(f"Search failed: {e}")

    async def search(self, query_vector: List[float], limit: int = 10, min_score: float = 0.0,
                     filters: Dict[str, Any] = None, search_params: Dict[str, Any] = None
                     ) -> List[Tuple[Chunk, float]]:
        t0 = time.time()
        out = self.search_sync(query_vector, limit, min_score, filters, search_params)
        self.logger.debug(f"HIP search completed in {time.time() - t0:.4f} seconds, found {len(out)} results")
        return out

    async def batch_search(self, query_vectors: List[List[float]], limit: int = 10, min_score: float = 0.0,
                           filters: Dict[str, Any] = None, search_params: Dict[str, Any] = None
                           ) -> List[List[Tuple[Chunk, float]]]:
        try:
            return self.batch_search_sync(query_vectors, limit, min_score, filters, search_params)
        except SearchError as e:
            raise SearchError(f"Batch search failed: {e}")

    # --------------------------------------------------------------- persistence
    def backup(self, path: str) -> Dict[str, Any]:
        """vector_db/qdrant_client.py:475-621 backup_collection: vectors as a binary
        snapshot (`path`), payloads and point ids as a JSON sidecar (`path + ".json"`)."""
        import json
        self.index.save(path)
        live = sorted(self._row_of.values())
        with open(path + ".json", "w") as f:
            json.dump({"collection_name": self.collection_name, "dimension": self.config.dimension,
                       "metric": self.config.metric.value, "payloads": self.payloads, "point_ids": self.point_ids,
                       "live_rows": live}, f)
        return {"collection_name": self.collection_name, "points_count": len(live), "path": path}

    @classmethod
    def restore(cls, path: str, config: SearchConfig) -> "HipSearchOptimizer":
        """vector_db/qdrant_client.py:623-764 restore_collection."""
        import json
        with open(path + ".json") as f:
            meta = json.load(f)
        if meta["dimension"] != config.dimension:
            raise SearchError("snapshot dimension differs from the config")
        self = cls.__new__(cls)
        self.config = config
        self.client = None
        self.collection_name = config.collection_name
        self.logger = logging.getLogger(__name__)
        self.index = DeviceIndex.load(path, device=config.device)
        self.payloads = meta["payloads"]
        self.point_ids = meta["point_ids"]
        self._row_of = {self.point_ids[r]: r for r in meta["live_rows"]}
        self._inactive = 0
        self._version = 0
        self._mask_cache = {}
        return self

    # --------------------------------------------------------------- bookkeeping
    async def get_collection_info(self) -> Dict[str, Any]:
        live = len(self._row_of)
        return {"name": self.collection_name, "vectors_count": live, "indexed_vectors_count": live,
                "points_count": live, "segments_count": 1,
                "config": {"vector_size": self.config.dimension, "distance": self.config.metric.value,
                           "hnsw_config": None, "optimizer_config": {"backend": "hip", "path": self.index.last_path}},
                "status": "green"}

    async def get_vector_count(self) -> int:
        return len(self._row_of)

    async def delete_collection(self) -> bool:
        try:
            self.index.close()
            self.payloads, self.point_ids, self._row_of = [], [], {}
            self._version += 1
            return True
        except Exception as e:
            self.logger.error(f"Failed to delete collection: {e}")
            return False

    def validate_health(self) -> Dict[str, Any]:
        try:
            live = len(self._row_of)
            return {"status": "healthy", "collection_name": self.collection_name, "vectors_count": live,
                    "indexed_vectors_count": live, "collection_status": "green"}
        except Exception as e:
            return {"status": "error", "message": f"Health check failed: {e}", "collection_name": self.collection_name}


SearchOptimizer = HipSearchOptimizer
