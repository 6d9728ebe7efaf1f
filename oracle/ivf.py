"""CPU oracle: inverted-file search with flat lists (FAISS ``IndexIVFFlat`` semantics
with an inner-product flat quantiser, as the target names them).  TEST
INFRASTRUCTURE ONLY.

add: a row goes to the list of the centroid with the largest inner product.
search: the ``nprobe`` best lists of the query by the same inner product, then an
exact inner-product top-k over the rows of those lists.  Ties (lists or rows) go
to the lower index, the library-wide rule.  FAISS is not in the reference tree
nor installed: **parity unpinned** against FAISS itself; id parity is defined
given identical centroids (k-means training is out of scope).

Coarse ties.  Which list a row lands in, and which lists a query probes, are
arg-max / top-nprobe decisions over fp32 inner products.  Two implementations that
sum in different orders agree on them except where the two scores involved lie
within fp32 rounding of each other; FAISS's own BLAS and SIMD paths differ the same
way.  ``ivf_check`` therefore accepts a result that differs from this oracle's ONLY
through rows whose membership in the probed set hinges on such a tie (the row's two
best centroid scores, or the query's nprobe-th and next list scores, within the
stated tolerance), and reports how many rows it excused.
"""
from __future__ import annotations

import numpy as np

from . import flat


def ivf_assign(X, centroids):
    S = (np.asarray(X, np.float32).astype(np.float64) @ np.asarray(centroids, np.float32).astype(np.float64).T)
    return np.argmax(S.astype(np.float32), axis=1).astype(np.int64)


def ivf_assign_detail(X, centroids, chunk=4096):
    """(a1, a2, gap): the best and second-best list of every row and the fp64 difference of
    their scores.  The three best lists are found with an fp32 product per chunk, then
    re-scored in fp64 (rounded once to fp32 for the arg-max, like ``ivf_assign``; ties to
    the lower list)."""
    X = np.ascontiguousarray(X, np.float32)
    C = np.ascontiguousarray(centroids, np.float32)
    n, nlist = X.shape[0], C.shape[0]
    a1 = np.zeros(n, np.int64)
    a2 = np.zeros(n, np.int64)
    gap = np.full(n, np.inf)
    if nlist == 1:
        return a1, a2, gap
    top = min(3, nlist)
    C64 = C.astype(np.float64)
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        S = X[r0:r1] @ C.T
        cand = np.empty((r1 - r0, top), np.int64)
        rows = np.arange(r1 - r0)
        for t in range(top):
            cand[:, t] = np.argmax(S, axis=1)
            S[rows, cand[:, t]] = -np.inf
        cand.sort(axis=1)                                   # lower list first: the tie rule
        s64 = np.einsum("ntd,nd->nt", C64[cand], X[r0:r1].astype(np.float64))
        s32 = s64.astype(np.float32)
        order = np.argsort(-s32, axis=1, kind="stable")     # stable on lists in increasing order
        best, second = order[:, 0], order[:, 1]
        a1[r0:r1] = cand[rows, best]
        a2[r0:r1] = cand[rows, second]
        gap[r0:r1] = s64[rows, best] - s64[rows, second]
    return a1, a2, gap


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


def _topk_rows(scores, rows, k, min_score):
    """k best (score, row) of the candidate rows: score descending, lower row first."""
    s = scores[rows]
    order = np.lexsort((rows, -s.astype(np.float64)))[:k]
    rows, s = rows[order], s[order]
    if min_score is not None:
        keep = s >= np.float32(min_score)
        rows, s = rows[keep], s[keep]
    return s, rows


def ivf_check(D, I, X, centroids, Q, k, nprobe, detail=None, min_score=None, row_mask=None, ids=None,
              rtol=1e-5, atol=1e-6):
    """Compare an IVF result with this oracle.  Returns (violations, excused_rows, ambiguous_queries).

    A query passes outright when (D, I) meets ``flat.compare_topk`` against the oracle's own
    answer.  Otherwise every difference must be explained by a coarse tie:
      * a row is *ambiguous* when its two best centroid scores differ by at most
        ``rtol*|best| + atol`` -- it may sit in either list;
      * a list is *ambiguous for a query* when its coarse score lies within that tolerance of
        the query's nprobe-th best while a list on the other side of the cut does too.
    U- = rows scanned whatever way the ties fall, U+ = rows scanned under some way.  The result
    must (1) only contain rows of U+ (that pass the mask), with scores equal to an fp64 re-score
    within tolerance and in descending order, no duplicates; (2) contain every row of U- that
    scores more than the tolerance above its last kept score; (3) hold between |top-k(U-)| and
    |top-k(U+)| hits.  ``excused_rows`` counts result rows outside the oracle's own answer (or
    oracle rows missing from the result) that are ambiguous in this sense.
    """
    X = np.ascontiguousarray(X, np.float32)
    C = np.ascontiguousarray(centroids, np.float32)
    Q = np.ascontiguousarray(Q, np.float32)
    if Q.ndim == 1:
        Q = Q[None, :]
    D = np.asarray(D)
    I = np.asarray(I)
    n, nlist = X.shape[0], C.shape[0]
    nprobe = min(nprobe, nlist)
    a1, a2, gap = detail if detail is not None else ivf_assign_detail(X, C)
    X64 = X.astype(np.float64)
    best_score = np.einsum("nd,nd->n", X64, C[a1].astype(np.float64))
    row_amb = gap <= rtol * np.abs(best_score) + atol
    allowed = np.ones(n, bool) if row_mask is None else np.asarray(row_mask).astype(bool)
    back = None
    if ids is not None:
        ids = np.asarray(ids, np.int64)
        back = {int(v): r for r, v in enumerate(ids)}
    coarse = (Q.astype(np.float64) @ C.astype(np.float64).T)             # [nq, nlist]
    violations, excused, amb_queries = [], 0, 0
    QB = 64
    S_blk = None
    for qi in range(Q.shape[0]):
        if qi % QB == 0:
            S_blk = X64 @ Q[qi:qi + QB].astype(np.float64).T            # [n, <= QB] exact scores of every row
        sc = coarse[qi]
        order = np.lexsort((np.arange(nlist), -sc.astype(np.float32).astype(np.float64)))
        probed = order[:nprobe]
        s_b = sc[probed[-1]]
        tol_b = rtol * abs(s_b) + atol
        near = np.abs(sc - s_b) <= tol_b
        outside_near = near.copy()
        outside_near[probed] = False
        amb_lists = np.nonzero(near)[0] if (nprobe < nlist and outside_near.any()) else np.zeros(0, np.int64)
        definite = np.setdiff1d(probed, amb_lists)
        possible = np.union1d(probed, amb_lists)
        in_def1, in_def2 = np.isin(a1, definite), np.isin(a2, definite)
        in_pos1, in_pos2 = np.isin(a1, possible), np.isin(a2, possible)
        u_minus = allowed & np.where(row_amb, in_def1 & in_def2, in_def1)
        u_plus = allowed & np.where(row_amb, in_pos1 | in_pos2, in_pos1)
        own = allowed & np.isin(a1, probed)                 # the oracle's own universe
        scores64 = np.ascontiguousarray(S_blk[:, qi % QB])
        scores = scores64.astype(np.float32)
        Dr = np.full(k, -flat.FLT_MAX, np.float32)
        Ir = np.full(k, -1, np.int64)
        s_own, r_own = _topk_rows(scores, np.nonzero(own)[0], k, min_score)
        Dr[: s_own.size] = s_own
        Ir[: r_own.size] = r_own if ids is None else ids[r_own]
        if not flat.compare_topk(D[qi:qi + 1], I[qi:qi + 1], Dr[None, :], Ir[None, :], rtol=rtol, atol=atol):
            continue
        # differs from the oracle's own answer: only coarse ties may explain it
        amb_queries += 1
        valid = I[qi] >= 0
        got_ids = I[qi][valid]
        got_rows = got_ids if back is None else np.array([back.get(int(v), -1) for v in got_ids], np.int64)
        got_s = D[qi][valid].astype(np.float64)
        tag = f"q{qi}"
        if np.any(got_rows < 0) or np.any(got_rows >= n):
            violations.append(f"{tag}: unknown id in result")
            continue
        if len(set(got_rows.tolist())) != got_rows.size:
            violations.append(f"{tag}: duplicate ids")
        if np.any(np.diff(got_s) > 0):
            violations.append(f"{tag}: scores not descending")
        ref_s = scores64[got_rows]
        if np.any(np.abs(got_s - ref_s) > rtol * np.abs(ref_s) + atol):
            violations.append(f"{tag}: a returned score differs from the fp64 re-score")
        outside = got_rows[~u_plus[got_rows]]
        if outside.size:
            violations.append(f"{tag}: rows {outside[:5].tolist()} are in no list the query may probe (or masked out)")
        s_minus, r_minus = _topk_rows(scores, np.nonzero(u_minus)[0], k, min_score)
        s_plus, r_plus = _topk_rows(scores, np.nonzero(u_plus)[0], k, min_score)
        if not (r_minus.size <= got_rows.size <= r_plus.size):
            violations.append(f"{tag}: {got_rows.size} hits, expected {r_minus.size}..{r_plus.size}")
        last = got_s[-1] if got_rows.size == k else -np.inf
        cut_tol = rtol * abs(last) + atol if np.isfinite(last) else 0.0
        must = r_minus[s_minus.astype(np.float64) > last + cut_tol]
        missing = np.setdiff1d(must, got_rows)
        if missing.size:
            violations.append(f"{tag}: rows {missing[:5].tolist()} are scanned whatever way the ties fall and beat the last hit")
        # how many rows of the difference the ties excuse (each of them is in U+ \ U-, or tied at the cut)
        diff = np.setxor1d(got_rows, r_own)
        excused += int(np.sum(u_plus[diff] & ~u_minus[diff]))
    return violations, excused, amb_queries
