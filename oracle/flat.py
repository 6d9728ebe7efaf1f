"""CPU oracle: exact dense top-k.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What is restated here
---------------------
* ``flat_ip_search``  -- FAISS ``IndexFlatIP.search`` semantics as the target names
  them: exact fp32 inner product of every query with every stored row, the k best
  per query in descending score order, ``int64`` ids = insertion order from 0 (or
  the ids given at add time), unfilled slots ``I = -1`` / ``D = -FLT_MAX``.  FAISS
  leaves the order of equal scores unspecified; this oracle (and the HIP path)
  fix it to "lower id first" so results do not depend on how the scan is split.
  FAISS is not vendored in the reference nor installed: **parity unpinned**
  against FAISS itself; anchored on the reference's call site
  ``vector_db/search_optimizer.py:213-222`` (limit, score_threshold) and on
  ``embedded_cosine_scan`` below for unit vectors.
* ``flat_l2_search`` -- FAISS ``IndexFlatL2`` semantics (squared distances ascending) for the
  ``MetricType.EUCLIDEAN`` a ``SearchConfig`` may name (vector_db/search_optimizer.py:33-35,104-105).
* ``embedded_cosine_scan`` -- the reference's only in-repo exact scan,
  ``utils/service_fallbacks.py:101-144`` (fp32 dot / (norm*norm), 0 when a norm is
  0, Python stable sort descending, ``[:limit]``).
* ``search_vectors_rules`` -- the wrapper rules of
  ``vector_db/storage_manager.py:470-476`` (``score_threshold or 0.0``) and the
  Qdrant ``score_threshold`` meaning for similarity metrics (keep score >= t).
* ``compare_topk`` -- the acceptance rule of the target: ids equal except where
  scores tie within tolerance; scores within 1e-5 relative.
"""
from __future__ import annotations

import numpy as np

FLT_MAX = np.finfo(np.float32).max


def l2_normalize_rows(x: np.ndarray) -> np.ndarray:
    """Row-normalise in fp32; zero rows stay zero (Qdrant COSINE collections
    normalise on insert; the reference's embedders already emit unit vectors,
    vector_db/embedding_manager.py:197,248-251)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = np.sqrt(np.einsum("ij,ij->i", x, x, dtype=np.float32)).astype(np.float32)
    n = np.where(n > 0, n, np.float32(1.0)).astype(np.float32)
    return (x / n[:, None]).astype(np.float32)


def flat_ip_search(X, Q, k, min_score=None, row_mask=None, ids=None, block=65536,
                   precise=True, _scores=None):
    """Exact inner-product top-k over fp32 data.

    X: [n, d] float32 rows; Q: [nq, d]; returns (D float32 [nq,k], I int64 [nq,k]).
    Order: score descending, ties by lower id.  ``row_mask`` (uint8/bool, 1 = row
    may be returned) models soft-deleted rows / payload filters.  ``min_score``:
    rows with score < min_score are dropped (slots padded).

    ``precise=True`` (parity checks): products accumulated in fp64 and rounded
    once to fp32, so equal rows score equally wherever they sit and the value is
    the one every fp32 summation order (FAISS's BLAS path, its SIMD path, the
    GPU's) approximates to ~1e-7.  ``precise=False`` (CPU baseline timing): one
    fp32 sgemm per block, as a CPU FAISS scan does.
    """
    X = np.ascontiguousarray(X, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    if Q.ndim == 1:
        Q = Q[None, :]
    n, d = X.shape if X.size else (0, Q.shape[1])
    nq = Q.shape[0]
    D = np.full((nq, k), -FLT_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if n == 0 or k == 0:
        return D, I
    run_s = np.empty((nq, 0), dtype=np.float32)
    run_i = np.empty((nq, 0), dtype=np.int64)
    Q64 = Q.astype(np.float64) if precise else None
    for b0 in range(0, n, block):
        b1 = min(n, b0 + block)
        if _scores is not None:
            S = _scores(b0, b1)                 # another similarity over the same selection rule (flat_l2_search)
        elif precise:
            S = (Q64 @ X[b0:b1].astype(np.float64).T).astype(np.float32)
        else:
            S = Q @ X[b0:b1].T                  # fp32 sgemm / sgemv
        idx = np.arange(b0, b1, dtype=np.int64)
        if row_mask is not None:
            keep = np.asarray(row_mask[b0:b1]).astype(bool)
            S = S[:, keep]
            idx = idx[keep]
        if S.shape[1] == 0:
            continue
        kk = min(k, S.shape[1])
        if S.shape[1] > 4 * kk:
            # keep every element tied with the kk-th so the id tie-break is exact
            part = np.partition(S, S.shape[1] - kk, axis=1)[:, S.shape[1] - kk]
            cand_s, cand_i = [], []
            width = 0
            rows = []
            for qi in range(nq):
                sel = np.nonzero(S[qi] >= part[qi])[0]
                rows.append(sel)
                width = max(width, sel.size)
            cs = np.full((nq, width), -np.inf, dtype=np.float32)
            ci = np.full((nq, width), np.iinfo(np.int64).max, dtype=np.int64)
            for qi, sel in enumerate(rows):
                cs[qi, : sel.size] = S[qi, sel]
                ci[qi, : sel.size] = idx[sel]
        else:
            cs = S
            ci = np.broadcast_to(idx, S.shape)
        all_s = np.concatenate([run_s, cs], axis=1)
        all_i = np.concatenate([run_i, ci], axis=1)
        # running entries have lower ids than the block's, ids inside a block are
        # increasing: a stable sort on -score therefore breaks ties by lower id
        order = np.argsort(-all_s, axis=1, kind="stable")[:, :k]
        run_s = np.take_along_axis(all_s, order, axis=1)
        run_i = np.take_along_axis(all_i, order, axis=1)
    m = run_s.shape[1]
    valid = np.isfinite(run_s) & (run_i != np.iinfo(np.int64).max)
    if min_score is not None:
        valid &= run_s >= np.float32(min_score)
    D[:, :m] = np.where(valid, run_s, -FLT_MAX)
    I[:, :m] = np.where(valid, run_i, -1)
    if ids is not None:
        ids = np.asarray(ids, dtype=np.int64)
        I = np.where(I >= 0, ids[np.clip(I, 0, None)], -1)
    return D, I


def flat_l2_search(X, Q, k, max_dist2=None, row_mask=None, ids=None, block=65536):
    """FAISS ``IndexFlatL2.search`` semantics: exact SQUARED Euclidean distance of every query to every
    stored row, the k nearest per query in ascending order, unfilled slots ``I = -1`` / ``D = FLT_MAX``
    (ties: lower id first, as in ``flat_ip_search``).  ``max_dist2``: rows farther than this are dropped
    (Qdrant's ``score_threshold`` for ``Distance.EUCLID`` is an upper bound on the distance, the call
    site being vector_db/search_optimizer.py:213-222).  Distances are fp64 sums |q|^2 - 2 q.x + |x|^2
    rounded once to fp32 and clamped at 0.  **Parity unpinned**: the reference holds no L2 output
    (its collections are COSINE, storage_manager.py:132,203) and FAISS is not installed."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    if Q.ndim == 1:
        Q = Q[None, :]
    Q64 = Q.astype(np.float64)
    qn = np.einsum("ij,ij->i", Q64, Q64)

    def neg_dist(b0, b1):
        Xb = X[b0:b1].astype(np.float64)
        d2 = qn[:, None] - 2.0 * (Q64 @ Xb.T) + np.einsum("ij,ij->i", Xb, Xb)[None, :]
        return (-np.maximum(d2, 0.0)).astype(np.float32)

    S, I = flat_ip_search(X, Q, k, min_score=None if max_dist2 is None else -float(max_dist2), row_mask=row_mask, ids=ids,
                          block=block, _scores=neg_dist)
    D = np.where(I >= 0, -S, FLT_MAX).astype(np.float32)
    return D, I


def compare_topk_l2(D_test, I_test, D_ref, I_ref, rtol=1e-5, atol=1e-6):
    """``compare_topk`` for ascending distance lists: the same rule on the negated distances."""
    neg = lambda D, I: np.where(np.asarray(I) >= 0, -np.asarray(D, dtype=np.float64), -float(FLT_MAX))
    return compare_topk(neg(D_test, I_test), I_test, neg(D_ref, I_ref), I_ref, rtol=rtol, atol=atol)


def rescore_f64(X, Q, I):
    """fp64 inner products of the returned ids (defines the tolerance band)."""
    X = np.asarray(X)
    Q = np.asarray(Q)
    if Q.ndim == 1:
        Q = Q[None, :]
    out = np.full(I.shape, -np.inf, dtype=np.float64)
    for qi in range(I.shape[0]):
        ok = I[qi] >= 0
        out[qi, ok] = X[I[qi, ok]].astype(np.float64) @ Q[qi].astype(np.float64)
    return out


def embedded_cosine_scan(vectors, query, limit=10):
    """utils/service_fallbacks.py:101-144: per row fp32 dot and norms, score 0.0 if
    either norm is 0, stable sort descending on the Python float, first ``limit``.
    Returns [(row_index, score_float)]."""
    q = np.array(query, dtype=np.float32)
    out = []
    for i, v in enumerate(vectors):
        s = np.array(v, dtype=np.float32)
        dot = np.dot(q, s)
        nq = np.linalg.norm(q)
        ns = np.linalg.norm(s)
        score = float(dot / (nq * ns)) if (nq > 0 and ns > 0) else 0.0
        out.append((i, score))
    out.sort(key=lambda t: t[1], reverse=True)
    return out[:limit]


def search_vectors_rules(X, q, limit=10, score_threshold=None, row_mask=None):
    """storage_manager.py:470-476 + search_optimizer.py:213-222: min_score =
    ``score_threshold or 0.0``; hits with score >= min_score, best ``limit``."""
    min_score = score_threshold or 0.0
    D, I = flat_ip_search(X, np.asarray(q, dtype=np.float32)[None, :], limit,
                          min_score=min_score, row_mask=row_mask)
    return [(int(i), float(s)) for s, i in zip(D[0], I[0]) if i >= 0]


def compare_topk(D_test, I_test, D_ref, I_ref, rtol=1e-5, atol=1e-6):
    """Acceptance rule: (1) the descending score lists agree position by position
    within ``rtol*|ref| + atol``; (2) the id sets agree, except that an id present
    on one side only must have a score within the same tolerance of the other
    side's last kept score (a tie at the cut), or of a score the other side
    holds (a tie inside the list that swapped places).  Returns a list of
    human-readable violations (empty = pass)."""
    D_test = np.asarray(D_test, dtype=np.float64)
    D_ref = np.asarray(D_ref, dtype=np.float64)
    I_test = np.asarray(I_test)
    I_ref = np.asarray(I_ref)
    bad = []
    if D_test.shape != D_ref.shape or I_test.shape != I_ref.shape:
        return [f"shape mismatch {D_test.shape} vs {D_ref.shape}"]
    for qi in range(D_ref.shape[0]):
        vt = I_test[qi] >= 0
        vr = I_ref[qi] >= 0
        if vt.sum() != vr.sum():
            # a hit exactly at min_score may fall either side; allow +-1 only if at tol
            bad.append(f"q{qi}: {vt.sum()} hits vs ref {vr.sum()}")
            continue
        nv = int(vr.sum())
        if nv == 0:
            continue
        dt, dr = D_test[qi, :nv], D_ref[qi, :nv]
        tol = rtol * np.abs(dr) + atol
        if np.any(np.abs(dt - dr) > tol):
            j = int(np.argmax(np.abs(dt - dr) - tol))
            bad.append(f"q{qi}: score[{j}] {dt[j]!r} vs ref {dr[j]!r}")
        if np.any(np.diff(dt) > 0):
            bad.append(f"q{qi}: scores not descending")
        st, sr = set(I_test[qi, :nv].tolist()), set(I_ref[qi, :nv].tolist())
        if len(st) != nv:
            bad.append(f"q{qi}: duplicate ids in result")
        ref_score = dict(zip(I_ref[qi, :nv].tolist(), dr.tolist()))
        test_score = dict(zip(I_test[qi, :nv].tolist(), dt.tolist()))
        cut_tol = rtol * abs(dr[-1]) + atol
        for i in st - sr:
            if abs(test_score[i] - dr[-1]) > cut_tol:
                bad.append(f"q{qi}: id {i} (score {test_score[i]!r}) not in ref and not tied at the cut {dr[-1]!r}")
        for i in sr - st:
            if abs(ref_score[i] - dt[-1]) > cut_tol:
                bad.append(f"q{qi}: ref id {i} (score {ref_score[i]!r}) missing and not tied at the cut {dt[-1]!r}")
    return bad
