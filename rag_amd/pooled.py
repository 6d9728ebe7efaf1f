"""Many small collections in ONE device corpus.

The reference keeps one collection pair per document (index ``doc_{id}``,
vector_db/document_version_manager.py:446,558) and searches them one RPC at a
time: ``for doc_id ...: for strategy ...: search_vectors(...)``
(vector_db/search_manager.py:879-913), two more per document for contextual
search (vector_db/storage_manager.py:635-650).  Here every collection of one
vector type and dimension is a set of row ranges ("segments") of a single
``DeviceIndex``, and the whole fan-out is ONE call of ``rag_index_search_segments``:
the k best of every (query vector, collection) pair.  Results per pair are the
same as a search of that collection alone.

``PooledView`` gives one collection of a pool the method set of
``HipSearchOptimizer``, so that it registers in
``VectorStorageManager.search_optimizers`` like any other backend object.
"""
from __future__ import annotations

import logging
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .filters import payload_passes
from .index import DeviceIndex
from .schemas import Chunk, SearchConfig, SearchError, device_limit


class PooledCollections:
    def __init__(self, dimension: int, metric: str = "cosine", device: int = 0):
        self.dimension = dimension
        self.index = DeviceIndex(dimension, metric=metric, device=device)
        self.ranges: Dict[str, List[List[int]]] = {}        # collection -> [[begin, end), ...]
        self.payloads: List[Dict[str, Any]] = []            # row -> payload
        self.point_ids: List[Any] = []                      # row -> point id
        self._rows: Dict[str, Dict[Any, int]] = {}          # collection -> {point id -> live row}
        self._hidden: set = set()                           # soft-deleted collections
        self._version = 0                                   # bumped by every change of rows or payloads
        self._mask_cache: Dict[Tuple[str, str], Tuple[int, np.ndarray]] = {}
        self._device_filters: Dict[Tuple[Tuple[str, ...], str], Tuple[int, Any]] = {}   # finished masks kept on the device
        self._distinct_cache: Dict[str, Tuple[int, bool]] = {}

    def close(self) -> None:
        self.index.close()

    def collections(self) -> List[str]:
        return list(self.ranges.keys())

    def count(self, name: str) -> int:
        return len(self._rows.get(name, ()))

    def row_of(self, name: str, point_id) -> Optional[int]:
        return self._rows.get(name, {}).get(point_id)

    def upsert(self, name: str, vectors, payloads: Sequence[Dict[str, Any]], ids: Sequence[Any]) -> None:
        x = np.ascontiguousarray(vectors, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.dimension:
            raise ValueError(f"expected vectors of dimension {self.dimension}")
        if not (len(payloads) == len(ids) == x.shape[0]):
            raise ValueError("vectors, payloads and ids must have the same length")
        rows = self._rows.setdefault(name, {})
        if x.shape[0] == 0:
            self.ranges.setdefault(name, [])
            return
        first = len(self.payloads)
        self.index.add(x)
        rng = self.ranges.setdefault(name, [])
        if rng and rng[-1][1] == first:
            rng[-1][1] = first + x.shape[0]                 # contiguous with the previous add
        else:
            rng.append([first, first + x.shape[0]])
        stale = []
        for j, (p, i) in enumerate(zip(payloads, ids)):
            self.payloads.append(dict(p) if p else {})
            self.point_ids.append(i)
            if i in rows:                                   # an earlier point with this id, of this batch too: the last one wins
                stale.append(rows[i])
            rows[i] = first + j
        if stale:
            self.index.set_active(stale, False)
        if name in self._hidden:                            # a hidden collection stays hidden
            self.index.set_active(list(range(first, first + x.shape[0])), False)
        self._version += 1

    def set_collection_active(self, name: str, active: bool) -> None:
        """vector_db/qdrant_client.py:826-883: hide / restore every point of a collection."""
        rows = sorted(self._rows.get(name, {}).values())
        if rows:
            self.index.set_active(rows, active)
        (self._hidden.discard if active else self._hidden.add)(name)

    def drop(self, name: str) -> None:
        """Remove a collection: its rows stay allocated but are never returned."""
        rows = list(self._rows.get(name, {}).values())
        if rows:
            self.index.set_active(rows, False)
        self._rows.pop(name, None)
        self.ranges.pop(name, None)
        self._hidden.discard(name)
        self._version += 1

    def dead_rows(self) -> int:
        """Rows that no collection refers to any more (replaced points, dropped collections)."""
        return len(self.payloads) - sum(len(r) for r in self._rows.values())

    def compact(self) -> int:
        """Rebuild the device corpus from the live rows, collection by collection (each
        collection becomes one row range again); returns the number of rows released.
        Dropped collections and replaced points only hide their rows
        (vector_db/qdrant_client.py:826-883 keeps soft-deleted points too); this is the
        vacuum step that gives their memory back."""
        dead = self.dead_rows()
        if dead == 0:
            return 0
        fresh = DeviceIndex(self.dimension, metric=self.index.metric, device=self.index.device)
        payloads, point_ids, ranges, rows_of = [], [], {}, {}
        hidden_rows = []
        for name, rows in self._rows.items():
            order = sorted(rows.items(), key=lambda kv: kv[1])          # insertion order within the collection
            first = len(payloads)
            if order:
                old = [r for _pid, r in order]
                for c0 in range(0, len(old), 65536):
                    # COSINE rows were normalised on the way in; normalising them again moves an
                    # element by one unit in the last place at most
                    fresh.add(self.index.reconstruct(old[c0:c0 + 65536]))
                for pid, r in order:
                    rows_of.setdefault(name, {})[pid] = len(payloads)
                    payloads.append(self.payloads[r])
                    point_ids.append(pid)
            rows_of.setdefault(name, {})
            ranges[name] = [[first, len(payloads)]] if len(payloads) > first else []
            if name in self._hidden:
                hidden_rows.extend(range(first, len(payloads)))
        for name in self.ranges:
            ranges.setdefault(name, [])
        if hidden_rows:
            fresh.set_active(hidden_rows, False)
        self.index.close()
        self.index, self.payloads, self.point_ids, self.ranges, self._rows = fresh, payloads, point_ids, ranges, rows_of
        self._version += 1
        self._mask_cache.clear()
        self._device_filters.clear()
        return dead

    def backup(self, path: str) -> Dict[str, Any]:
        """vector_db/qdrant_client.py:475-621 backup_collection for the whole pool: the device corpus
        as one binary snapshot (`path`), collections / payloads / point ids as a JSON sidecar."""
        import json
        self.index.save(path)
        with open(path + ".json", "w") as f:
            json.dump({"dimension": self.dimension, "metric": self.index.metric, "ranges": self.ranges,
                       "payloads": self.payloads, "point_ids": self.point_ids,
                       "rows": [[name, [[pid, r] for pid, r in rows.items()]] for name, rows in self._rows.items()],
                       "hidden": sorted(self._hidden)}, f)
        return {"collections": len(self.ranges), "points_count": sum(len(r) for r in self._rows.values()), "path": path}

    @classmethod
    def restore(cls, path: str, device: int = 0) -> "PooledCollections":
        """vector_db/qdrant_client.py:623-764 restore_collection for a pool written by ``backup``."""
        import json
        with open(path + ".json") as f:
            meta = json.load(f)
        self = cls.__new__(cls)
        self.dimension = meta["dimension"]
        self.index = DeviceIndex.load(path, device=device)
        if self.index.dim != self.dimension:
            self.index.close()
            raise SearchError("snapshot dimension differs from its sidecar")
        self.ranges = {name: [list(r) for r in rs] for name, rs in meta["ranges"].items()}
        self.payloads = meta["payloads"]
        self.point_ids = meta["point_ids"]
        self._rows = {name: {(tuple(pid) if isinstance(pid, list) else pid): r for pid, r in rows} for name, rows in meta["rows"]}
        self._hidden = set(meta["hidden"])
        self._version = 0
        self._mask_cache = {}
        self._device_filters = {}
        self._distinct_cache = {}
        return self

    def get_vector(self, name: str, point_id) -> Optional[np.ndarray]:
        row = self.row_of(name, point_id)
        return None if row is None else self.index.reconstruct([row])[0]

    def filter_mask(self, names: Sequence[str], filters: Optional[Dict[str, Any]]) -> Optional[np.ndarray]:
        """One byte per pool row: 1 where the payload passes ``filters``, decided
        for the rows of ``names`` only (other rows are never scanned by the
        caller).  The per-collection pieces are kept until the pool changes, so a
        repeated filter (the per-user filter of vector_db/search_manager.py:893)
        costs one copy."""
        if not filters:
            return None
        fkey = repr(sorted(filters.items(), key=lambda kv: kv[0]))
        # the finished mask of a (collections, filter) pair stays on the device until the pool changes: the same user
        # searching the same documents again uploads nothing
        dkey = (tuple(names), fkey)
        dev_hit = self._device_filters.get(dkey)
        if dev_hit is not None and dev_hit[0] == self._version:
            return dev_hit[1]
        mask = np.zeros(len(self.payloads), dtype=np.uint8)
        for name in names:
            hit = self._mask_cache.get((name, fkey))
            if hit is None or hit[0] != self._version:
                rows = np.fromiter((r for r in self._rows.get(name, {}).values()
                                    if payload_passes(self.payloads[r], filters)), dtype=np.int64)
                if len(self._mask_cache) >= 65536:           # bounded: start over rather than grow
                    self._mask_cache.clear()
                self._mask_cache[(name, fkey)] = hit = (self._version, rows)
            mask[hit[1]] = 1
        if mask.size != len(self.index):
            return mask
        if len(self._device_filters) >= 16:                  # bounded: each entry holds one byte per pool row on the device
            self._device_filters.clear()
        dev = self.index.make_filter(mask)
        self._device_filters[dkey] = (self._version, dev)
        return dev

    def search_many(self, names: Sequence[str], query_vectors, limit: int, min_score: Optional[float] = 0.0,
                    row_mask=None) -> List[List[List[Tuple[int, float]]]]:
        """hits[qi][ci] = [(row, score), ...] best first for query qi in collection
        names[ci] (an unknown collection gives [])."""
        q = np.ascontiguousarray(query_vectors, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        segs, owner = [], []
        for ci, name in enumerate(names):
            for b, e in self.ranges.get(name, []):
                if e > b:
                    segs.append((b, e))
                    owner.append(ci)
        out = [[[] for _ in names] for _ in range(q.shape[0])]
        if not segs or limit < 1:
            return out
        D, I = self.index.search_segments(q, segs, limit, min_score=min_score, row_mask=row_mask)
        # unfilled slots (-1) sit at the end of every list: one count per list, then plain slices
        counts = (np.asarray(I) >= 0).sum(axis=-1).tolist()
        rows, scores = np.asarray(I).tolist(), np.asarray(D).tolist()
        pieces = [0] * len(names)
        for ci in owner:
            pieces[ci] += 1
        for qi in range(q.shape[0]):
            per = out[qi]
            rq, sq, cq = rows[qi], scores[qi], counts[qi]
            for si, ci in enumerate(owner):
                c = cq[si]
                if c:
                    per[ci].extend(zip(rq[si][:c], sq[si][:c]))
            for ci, hits in enumerate(per):
                if pieces[ci] > 1:                           # collection spread over several ranges
                    hits.sort(key=lambda t: (-t[1], t[0]))
                    del hits[limit:]
        return out

    def chunk_ids_distinct(self, name: str) -> bool:
        """True when every live point of the collection shows a chunk id of its own in its result
        dictionary (kept until the pool changes).  A payload that carries an explicit empty / None
        ``chunk_id`` makes it False: such a hit has no key to fuse under and the reference drops it
        BEFORE the cut to ``limit`` (vector_db/search_manager.py:899-901), which a device-side cut
        cannot reproduce -- those collections take the list fusion on the host."""
        hit = self._distinct_cache.get(name)
        if hit is None or hit[0] != self._version:
            rows = self._rows.get(name, {})
            seen, ok = set(), True
            for r in rows.values():
                payload = self.payloads[r] or {}
                if "chunk_id" in payload and not payload["chunk_id"]:
                    ok = False
                    break
                seen.add(payload.get("chunk_id") or str(self.point_ids[r]))
            if len(self._distinct_cache) >= 65536:
                self._distinct_cache.clear()
            self._distinct_cache[name] = hit = (self._version, ok and len(seen) == len(rows))
        return hit[1]

    def fused_search(self, names: Sequence[str], query_vectors, weights: Sequence[float], per_limit: int, limit: int,
                     min_score: Optional[float] = 0.0, row_mask=None) -> Optional[List[Tuple[int, int, float, int]]]:
        """The loop of vector_db/search_manager.py:879-932 over the collections ``names``
        (documents, in order) and the query vectors (strategies, in order) as one device
        call: [(index into names, row, weighted score, index of the winning query
        vector)], best first, at most ``limit``.  None when the device call does not
        apply (a collection spread over several row ranges, too many candidates per
        collection); the caller then fuses the lists of ``search_many``."""
        q = np.ascontiguousarray(query_vectors, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[0] > 16 or q.shape[0] * per_limit > 1024 or not 1 <= limit <= 1024 or per_limit < 1:
            return None
        segs, owner = [], []
        for ci, name in enumerate(names):
            live = [(b, e) for b, e in self.ranges.get(name, []) if e > b]
            if len(live) > 1:
                return None
            if live:
                segs.append(live[0])
                owner.append(ci)
        if not segs:
            return []
        rows, scores, strategy = self.index.search_segments_fused(q, weights, segs, per_limit, limit, min_score=min_score,
                                                                  row_mask=row_mask)
        begins = np.asarray([b for b, _e in segs], dtype=np.int64)
        order = np.argsort(begins, kind="stable")
        at = order[np.searchsorted(begins[order], rows, side="right") - 1] if len(rows) else []
        return [(owner[int(si)], int(r), float(s), int(st)) for si, r, s, st in zip(at, rows, scores, strategy)]

    def chunks_of(self, hits: Sequence[Tuple[int, float]]) -> List[Tuple[Chunk, float]]:
        """(row, score) pairs as the reference's (Chunk, score) tuples
        (vector_db/search_optimizer.py:228-258)."""
        out = []
        for row, score in hits:
            payload = self.payloads[row] or {}
            chunk_id = payload.get("chunk_id") or str(self.point_ids[row])
            out.append((Chunk(document_id=payload.get("document_id", 0), chunk_id=chunk_id,
                              start_idx=payload.get("start_char", 0), end_idx=payload.get("end_char", 0),
                              text=payload.get("text", ""), metadata=payload, embedding=None), float(score)))
        return out


    def hit_dicts(self, hits: Sequence[Tuple[int, float]]) -> List[Dict[str, Any]]:
        """(row, score) pairs as the result dictionaries of
        vector_db/storage_manager.py:477-505 -- what formatting ``chunks_of(hits)``
        gives, without the intermediate objects."""
        out = []
        for row, score in hits:
            payload = self.payloads[row] or {}
            metadata = {"chunk_id": payload.get("chunk_id") or str(self.point_ids[row]),
                        "document_id": payload.get("document_id", 0),
                        "start_char": payload.get("start_char", 0), "end_char": payload.get("end_char", 0)}
            if payload:
                metadata.update(payload)
            out.append({"score": float(score), "content": payload.get("text", "") or "", "metadata": metadata,
                        "source": "hip"})
        return out


class PooledView:
    """One collection of a ``PooledCollections`` behind the interface of the
    reference's per-collection backend object (``QdrantSearchOptimizer``,
    vector_db/search_optimizer.py:68-406): ``search`` / ``batch_search`` return
    ``(Chunk, score)`` tuples best first and raise ``SearchError``."""

    def __init__(self, pool: PooledCollections, config: SearchConfig):
        if config.dimension != pool.dimension:
            raise SearchError(f"pool holds vectors of dimension {pool.dimension}")
        self.pool = pool
        self.config = config
        self.collection_name = config.collection_name
        self.logger = logging.getLogger(__name__)
        pool.ranges.setdefault(self.collection_name, [])
        pool._rows.setdefault(self.collection_name, {})

    @property
    def index(self) -> DeviceIndex:
        return self.pool.index

    async def create_collection(self) -> bool:
        return True

    def upsert(self, vectors, payloads: Sequence[Dict[str, Any]], ids: Sequence[Any]) -> List[Any]:
        try:
            self.pool.upsert(self.collection_name, vectors, payloads, ids)
        except ValueError as e:
            raise SearchError(str(e))
        return list(ids)

    def get_vector(self, point_id) -> Optional[Dict[str, Any]]:
        row = self.pool.row_of(self.collection_name, point_id)
        if row is None:
            return None
        return {"id": point_id, "vector": self.pool.index.reconstruct([row])[0].tolist(),
                "payload": self.pool.payloads[row], "row": row}

    def row_of(self, point_id) -> Optional[int]:
        return self.pool.row_of(self.collection_name, point_id)

    def set_points_active(self, active: bool) -> None:
        self.pool.set_collection_active(self.collection_name, active)

    def search_sync(self, query_vector, limit: int = 10, min_score: float = 0.0,
                    filters: Dict[str, Any] = None, search_params: Dict[str, Any] = None):
        return self.batch_search_sync([query_vector], limit, min_score, filters, search_params)[0]

    def batch_search_sync(self, query_vectors, limit: int = 10, min_score: float = 0.0,
                          filters: Dict[str, Any] = None, search_params: Dict[str, Any] = None):
        try:
            if getattr(self, "_disabled", False):
                return [[] for _ in query_vectors]
            q = np.ascontiguousarray(query_vectors, dtype=np.float32)
            if q.ndim != 2 or q.shape[1] != self.config.dimension:
                raise ValueError(f"query vectors must have dimension {self.config.dimension}")
            if limit < 1:
                return [[] for _ in range(q.shape[0])]
            names = [self.collection_name]
            hits = self.pool.search_many(names, q, device_limit(limit), min_score=min_score,
                                         row_mask=self.pool.filter_mask(names, filters))
            return [self.pool.chunks_of(hits[i][0]) for i in range(q.shape[0])]
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

    async def get_collection_info(self) -> Dict[str, Any]:
        live = self.pool.count(self.collection_name)
        return {"name": self.collection_name, "vectors_count": live, "indexed_vectors_count": live,
                "points_count": live, "segments_count": len(self.pool.ranges.get(self.collection_name, [])),
                "config": {"vector_size": self.config.dimension, "distance": self.config.metric.value,
                           "hnsw_config": None, "optimizer_config": {"backend": "hip", "path": "pooled-scan"}},
                "status": "green"}

    async def get_vector_count(self) -> int:
        return self.pool.count(self.collection_name)

    async def delete_collection(self) -> bool:
        try:
            self.pool.drop(self.collection_name)
            return True
        except Exception as e:
            self.logger.error(f"Failed to delete collection: {e}")
            return False

    def validate_health(self) -> Dict[str, Any]:
        live = self.pool.count(self.collection_name)
        return {"status": "healthy", "collection_name": self.collection_name, "vectors_count": live,
                "indexed_vectors_count": live, "collection_status": "green"}
