"""Few queries over a corpus of at most 32768 rows (path "direct": every row scored, then one selection
workgroup per query that holds the scores in registers) against the CPU oracle -- small collections and
the coarse step of an IVF search take this path.  Same tolerance as the exhaustive search."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _check(D, I, X, Q, k, **kw):
    from oracle import flat
    bad = flat.compare_topk(D, I, *flat.flat_ip_search(X, Q, k, **kw), rtol=RTOL, atol=ATOL)
    assert bad == [], bad[:5]


@pytest.mark.parametrize("n,d", [(1024, 36), (5_000, 100), (32_768, 384), (30_001, 768), (16_384, 1024), (4_097, 3072),
                                 (32_767, 64)])
def test_direct_path_matches_oracle(n, d):
    from paths import flat_fp32_path
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(n + d)
    X = _unit(rng, n, d) * rng.uniform(0.5, 1.5, (n, 1)).astype(np.float32)
    Q = _unit(rng, 4, d)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X[: n // 3])
    idx.add(X[n // 3:])
    for nq in (1, 2, 3, 4):
        for k in (1, 10, 64, 200, 1024):
            D, I = idx.search(Q[:nq], k)
            assert idx.last_path == flat_fp32_path(n, d, nq, k) == "direct"
            _check(D, I, X, Q[:nq], k)
    mask = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    D, I = idx.search(Q, 10, row_mask=mask, min_score=0.02)
    assert idx.last_path == "direct"
    _check(D, I, X, Q, 10, row_mask=mask, min_score=0.02)
    assert mask[I[I >= 0]].all()
    D, I = idx.search(np.concatenate([Q, Q[:1]]), 10)         # five queries: the general path
    assert idx.last_path != "direct"
    _check(D, I, X, np.concatenate([Q, Q[:1]]), 10)
    idx.close()


def test_direct_path_ties_ids_soft_deletes_and_few_rows():
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(77)
    n, d, k = 20_000, 128, 50
    X = _unit(rng, n, d)
    X[3000:9000] = X[3000]                                    # 6000 exactly equal scores around every cut
    ids = rng.permutation(n).astype(np.int64) * 7 + 3
    idx = DeviceIndex(d, metric="cosine")
    idx.add(X * 3.0, ids=ids)                                 # cosine: rows are normalised on the way in
    Q = np.stack([X[3000], -X[3000], X[10], X[3000] + 0.1 * X[11]]).astype(np.float32)
    for nq in (1, 2, 3, 4):
        for kk in (k, 1000):
            D, I = idx.search(Q[:nq], kk)
            assert idx.last_path == "direct"
            _check(D, I, X, Q[:nq] / np.linalg.norm(Q[:nq], axis=1, keepdims=True), kk, ids=ids)
    # equal scores come back lower row first
    D, I = idx.search(Q[:1], 100)
    assert np.array_equal(I[0], ids[3000:3100])
    # soft deletes
    dead = np.arange(3000, 3050)
    idx.set_active(dead, False)
    D, I = idx.search(Q[:1], 100)
    assert np.array_equal(I[0], ids[3050:3150])
    # fewer live rows than k: the rest of the answer is padding
    idx.set_active(np.arange(0, n - 5), False)
    D, I = idx.search(Q[:2], 10)
    assert (I[:, 5:] == -1).all() and (I[:, :5] >= 0).all() and (D[:, 5:] < -3e38).all()
    idx.close()


def test_direct_path_skips_nan_rows_and_equals_the_scan_on_ids():
    import os
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(5)
    n, d = 25_000, 384
    X = _unit(rng, n, d)
    X[123, 7] = np.nan                                        # a NaN row scores nowhere
    Q = _unit(rng, 4, d)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    D, I = idx.search(Q, 10)
    assert idx.last_path == "direct" and 123 not in I
    Xc = X.copy()
    Xc[123] = 0.0
    from oracle import flat
    Dr, Ir = flat.flat_ip_search(Xc, Q, 10)
    assert flat.compare_topk(D, I, Dr, Ir, rtol=RTOL, atol=ATOL) == []
    # the general path (five queries) returns the same ids for the same queries
    D5, I5 = idx.search(np.concatenate([Q, Q[:1]]), 10)
    assert idx.last_path != "direct" and np.array_equal(I5[:4], I) and np.array_equal(I5[4], I[0])
    idx.close()
