"""CPU oracle: inverted-file search with flat lists (FAISS ``IndexIVFFlat`` semantics
with an inner-product flat quantiser, as the target names them).  TEST
INFRASTRUCTURE ONLY.

add: a row goes to the list of the centroid with the largest inner product.
search: the ``nprobe`` best lists of the query by the same inner product, then an
exact inner-product top-k over the rows of those lists.  Ties (lists or rows) go
to the lower index, the library-wide rule.  FAISS is not in the reference tree
nor installed: **parity unpinned** against FAISS itself; id parity is defined
given identical centroids (k-means training is out of scope).
"""
from __future__ import annotations

import numpy as np

from . import flat


def ivf_assign(X, centroids):
    S = (np.asarray(X, np.float32).astype(np.float64) @ np.asarray(centroids, np.float32).astype(np.float64).T)
    return np.argmax(S.astype(np.float32), axis=1).astype(np.int64)


def ivf_search(X, centroids, Q, k, nprobe, assign=None, min_score=None, row_mask=None, ids=None):
    X = np.ascontiguousarray(X, np.float32)
    Q = np.ascontiguousarray(Q, np.float32)
    if Q.ndim == 1:
        Q = Q[None, :]
    C = np.ascontiguousarray(centroids, np.float32)
    if assign is None:
        assign = ivf_assign(X, C)
    nprobe = min(nprobe, C.shape[0])
    _, lists = flat.flat_ip_search(C, Q, nprobe)
    D = np.empty((Q.shape[0], k), np.float32)
    I = np.empty((Q.shape[0], k), np.int64)
    for qi in range(Q.shape[0]):
        m = np.isin(assign, lists[qi][lists[qi] >= 0])
        if row_mask is not None:
            m &= np.asarray(row_mask).astype(bool)
        D[qi], I[qi] = [a[0] for a in flat.flat_ip_search(X, Q[qi:qi + 1], k, min_score=min_score,
                                                            row_mask=m.astype(np.uint8), ids=ids)]
    return D, I
