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
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .index import DeviceIndex


class PooledCollections:
    def __init__(self, dimension: int, metric: str = "cosine", device: int = 0):
        self.dimension = dimension
        self.index = DeviceIndex(dimension, metric=metric, device=device)
        self.ranges: Dict[str, List[List[int]]] = {}        # collection -> [[begin, end), ...]
        self.payloads: List[Dict[str, Any]] = []            # row -> payload
        self.point_ids: List[Any] = []                      # row -> point id
        self._row_of: Dict[Tuple[str, Any], int] = {}       # (collection, point id) -> live row

    def close(self) -> None:
        self.index.close()

    def collections(self) -> List[str]:
        return list(self.ranges.keys())

    def count(self, name: str) -> int:
        return sum(1 for (c, _i) in self._row_of if c == name)

    def upsert(self, name: str, vectors, payloads: Sequence[Dict[str, Any]], ids: Sequence[Any]) -> None:
        x = np.ascontiguousarray(vectors, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.dimension:
            raise ValueError(f"expected vectors of dimension {self.dimension}")
        if not (len(payloads) == len(ids) == x.shape[0]):
            raise ValueError("vectors, payloads and ids must have the same length")
        if x.shape[0] == 0:
            self.ranges.setdefault(name, [])
            return
        first = len(self.payloads)
        stale = [self._row_of[(name, i)] for i in ids if (name, i) in self._row_of]
        self.index.add(x)
        rng = self.ranges.setdefault(name, [])
        if rng and rng[-1][1] == first:
            rng[-1][1] = first + x.shape[0]                 # contiguous with the previous add
        else:
            rng.append([first, first + x.shape[0]])
        for j, (p, i) in enumerate(zip(payloads, ids)):
            self.payloads.append(dict(p) if p else {})
            self.point_ids.append(i)
            self._row_of[(name, i)] = first + j
        if stale:
            self.index.set_active(stale, False)

    def drop(self, name: str) -> None:
        """Remove a collection: its rows stay allocated but are never returned."""
        rows = [r for (c, _i), r in self._row_of.items() if c == name]
        if rows:
            self.index.set_active(rows, False)
        for key in [key for key in self._row_of if key[0] == name]:
            del self._row_of[key]
        self.ranges.pop(name, None)

    def get_vector(self, name: str, point_id) -> Optional[np.ndarray]:
        row = self._row_of.get((name, point_id))
        return None if row is None else self.index.reconstruct([row])[0]

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
        for qi in range(q.shape[0]):
            per = [[] for _ in names]
            for si, ci in enumerate(owner):
                for s, r in zip(D[qi, si], I[qi, si]):
                    if r < 0:
                        break
                    per[ci].append((int(r), float(s)))
            for ci, hits in enumerate(per):
                if sum(1 for c in owner if c == ci) > 1:     # collection spread over several ranges
                    hits.sort(key=lambda t: (-t[1], t[0]))
                out[qi][ci] = hits[:limit]
        return out
