"""The L2 metric (RAG_METRIC_L2, MetricType.EUCLIDEAN of vector_db/search_optimizer.py:33-35,104-105):
squared distances, nearest first, against the oracle's restatement of FAISS IndexFlatL2.

Parity unpinned against FAISS / Qdrant themselves: neither is installed and the reference holds no
Euclidean output (its collections are COSINE).  The device picks the k rows by q.x - |x|^2 / 2 in fp32
(an absolute error of about 1e-7 (|q|^2 + |x|^2) in what it ranks by, so rows that close to the cut may
swap) and then computes the kept rows' distances as sums of squared differences; the comparison allows
1e-5 relative plus 1e-6 times that scale."""
import asyncio

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _run(coro):
    return asyncio.get_event_loop().run_until_complete(coro)


def _rows(rng, n, d, scale=1.0):
    return (rng.standard_normal((n, d)) * scale / np.sqrt(d)).astype(np.float32)


def _atol(X, Q):
    return 1e-6 * max(1.0, float((X.astype(np.float64) ** 2).sum(1).max() + (Q.astype(np.float64) ** 2).sum(1).max()))


@pytest.mark.parametrize("n,d,nq,k", [(1, 8, 1, 5), (700, 36, 3, 10), (20_000, 384, 1, 10), (40_000, 768, 1, 100), (30_000, 100, 7, 10),
                                       (50_000, 768, 40, 10), (25_000, 127, 300, 20), (9_000, 1024, 1030, 5), (3_000, 31, 2, 1024),
                                       (70_000, 128, 40, 100), (16_384, 64, 300, 200)])      # k above 64 in a batch: segment walk, score matrix
def test_l2_search_matches_oracle(n, d, nq, k):
    from oracle import flat
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(n + nq)
    X = _rows(rng, n, d, scale=float(rng.uniform(0.5, 3.0)))
    Q = _rows(rng, nq, d)
    if n > 10:
        X[n // 2] = X[1]                       # equal distances: the lower row first
        Q[0] = X[1]                            # distance exactly 0 to two rows
    idx = DeviceIndex(d, metric="l2")
    idx.add(X[: n // 2])
    idx.add(X[n // 2:])
    assert len(idx) == n and idx.metric == "l2"
    D, I = idx.search(Q, k)
    Dr, Ir = flat.flat_l2_search(X, Q, k)
    bad = flat.compare_topk_l2(D, I, Dr, Ir, rtol=RTOL, atol=_atol(X, Q))
    assert bad == [], (idx.last_path, bad[:3])
    valid = I >= 0
    assert np.all(D[~valid] == np.finfo(np.float32).max) and np.all(D[valid] >= 0.0)
    assert np.all(D[:, 1:] >= D[:, :-1])                                                 # nearest first, empty slots last
    if n > 10:
        assert I[0, 0] == 1 and (k < 2 or I[0, 1] == n // 2) and D[0, 0] <= _atol(X, Q)
    # the stored rows come back as given (the extra column never shows)
    rows = rng.integers(0, n, 16)
    assert np.array_equal(idx.reconstruct(rows), X[rows])
    idx.close()


def test_l2_limit_mask_ids_soft_delete_and_snapshot(tmp_path):
    from oracle import flat
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(9)
    n, d, nq, k = 12_000, 256, 5, 20
    X = _rows(rng, n, d, 2.0)
    Q = _rows(rng, nq, d, 2.0)
    ids = rng.permutation(10 * n)[:n].astype(np.int64)
    idx = DeviceIndex(d, metric="l2")
    idx.add(X, ids=ids)
    atol = _atol(X, Q)
    D0, _ = idx.search(Q, k)
    limit = float(np.median(D0[:, k // 2]))                   # cuts some lists short, keeps others whole
    mask = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    for kw, okw in (({}, {}), ({"min_score": limit}, {"max_dist2": limit}), ({"row_mask": mask}, {"row_mask": mask}),
                    ({"row_mask": mask, "min_score": limit}, {"row_mask": mask, "max_dist2": limit}), ({"min_score": 0.0}, {"max_dist2": 0.0})):
        D, I = idx.search(Q, k, **kw)
        Dr, Ir = flat.flat_l2_search(X, Q, k, ids=ids, **okw)
        assert flat.compare_topk_l2(D, I, Dr, Ir, rtol=RTOL, atol=atol) == [], kw
    back = {int(v): r for r, v in enumerate(ids)}
    hide = np.array([back[int(i)] for i in np.unique(idx.search(Q, k)[1][:, :3])])       # the three nearest of every query
    idx.set_active(hide, False)
    alive = np.ones(n, np.uint8)
    alive[hide] = 0
    D, I = idx.search(Q, k)
    assert flat.compare_topk_l2(D, I, *flat.flat_l2_search(X, Q, k, ids=ids, row_mask=alive), rtol=RTOL, atol=atol) == []
    path = str(tmp_path / "l2.snap")
    idx.save(path)
    again = DeviceIndex.load(path)
    assert again.metric == "l2" and len(again) == n
    D2, I2 = again.search(Q, k)
    assert np.array_equal(D2, D) and np.array_equal(I2, I)
    again.close()
    idx.close()
    empty = DeviceIndex(d, metric="l2")
    D, I = empty.search(Q, 3)
    assert np.all(I == -1) and np.all(D == np.finfo(np.float32).max)
    empty.close()


def test_l2_is_refused_where_answers_are_ranked_by_similarity():
    from rag_amd import DeviceIndex, MultiDeviceIndex, RagHipError
    from rag_amd.index import hybrid_mmr
    from rag_amd.sharded import ShardedIndex
    rng = np.random.default_rng(2)
    X = _rows(rng, 500, 64)
    idx = DeviceIndex(64, metric="l2")
    idx.add(X)
    with pytest.raises(RagHipError):
        DeviceIndex(64, metric="l2", kind="ivf")
    with pytest.raises(RagHipError):
        idx.set_prefilter(True)
    with pytest.raises(RagHipError):
        idx.search_segments(X[:1], [(0, 100)], 5)
    with pytest.raises(RagHipError):
        hybrid_mmr(idx, X[0], 20, np.zeros(0, np.int64), np.zeros(0))
    multi = MultiDeviceIndex(64, [0, 0], metric="l2")
    with pytest.raises(ValueError):
        multi.hybrid_mmr(X[0], 20, np.zeros(0, np.int64), np.zeros(0))
    multi.close()
    with pytest.raises(ValueError):
        ShardedIndex(idx).hybrid_mmr(X[0], 20, np.zeros(0, np.int64), np.zeros(0))
    idx.close()


@pytest.mark.parametrize("shards,n,d,nq,k", [(2, 9_000, 64, 1, 10), (3, 40_000, 384, 40, 10), (5, 20_003, 100, 300, 64)])
def test_l2_over_several_devices_equals_one_index(shards, n, d, nq, k):
    """rag_multi_* with the L2 metric: the devices' answers merged nearest first (ties to the lower id), empty
    slots as (-1, FLT_MAX); and the sharded exchange (distances travel negated through the one merge) replayed
    for the same shards on one GPU."""
    import torch
    from oracle import flat
    from rag_amd import DeviceIndex, MultiDeviceIndex
    from rag_amd.sharded import ShardedIndex, shard_range
    rng = np.random.default_rng(shards + n)
    X = _rows(rng, n, d, 2.0)
    X[n - 1] = X[2]                                           # equal distances across shards
    Q = _rows(rng, nq, d, 2.0)
    Q[0] = X[2]
    multi = MultiDeviceIndex(d, [0] * shards, metric="l2")
    multi.add(X[: n // 2])
    multi.add(X[n // 2:])
    D, I = multi.search(Q, k)
    Dr, Ir = flat.flat_l2_search(X, Q, k)
    assert flat.compare_topk_l2(D, I, Dr, Ir, rtol=RTOL, atol=_atol(X, Q)) == []
    assert I[0, 0] == 2 and I[0, 1] == n - 1 and D[0, 0] == 0.0 and D[0, 1] == 0.0
    lim = float(np.median(D[:, k // 2]))
    mask = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    D2, I2 = multi.search(Q, k, min_score=lim, row_mask=mask)
    assert flat.compare_topk_l2(D2, I2, *flat.flat_l2_search(X, Q, k, max_dist2=lim, row_mask=mask), rtol=RTOL, atol=_atol(X, Q)) == []
    multi.close()
    # the per-rank step of the sharded search; the collective is the stand-in of test_sharded_device_gpu.py, which
    # places every rank's send slot where the all-gather would
    from test_sharded_device_gpu import _Collective
    qt = torch.from_numpy(Q).cuda()
    parts = []
    for r in range(shards):
        lo, hi = shard_range(n, r, shards)
        idx = DeviceIndex(d, metric="l2")
        idx.set_id_base(lo)
        idx.add(X[lo:hi])
        parts.append(idx)
    slots = {}
    for r in range(shards):
        sh = ShardedIndex(parts[r])
        sh.dist, sh.world, sh.rank = _Collective(r, shards, {}), shards, r
        sh.search(qt, k)
        slots[r] = sh.dist.sent
    for r in range(shards):
        sh = ShardedIndex(parts[r])
        sh.dist, sh.world, sh.rank = _Collective(r, shards, {o: v for o, v in slots.items() if o != r}), shards, r
        Ds, Is = sh.search(qt, k)
        torch.cuda.synchronize()
        assert flat.compare_topk_l2(Ds.cpu().numpy(), Is.cpu().numpy(), Dr, Ir, rtol=RTOL, atol=_atol(X, Q)) == [], r
        assert Is[0, 0].item() == 2 and Is[0, 1].item() == n - 1 and Ds[0, 1].item() == 0.0
    for p in parts:
        p.close()


def test_optimizer_with_the_euclidean_metric():
    """HipSearchOptimizer(SearchConfig(metric=EUCLIDEAN)): scores are distances (not squared), nearest
    first, and min_score is the largest distance returned -- Qdrant's reading of score_threshold for
    Distance.EUCLID, which the reference feeds with min_score (search_optimizer.py:213-222)."""
    from rag_amd.schemas import IndexType, MetricType, SearchConfig, SearchError
    from rag_amd.search_optimizer import HipSearchOptimizer
    rng = np.random.default_rng(4)
    n, d = 3000, 48
    X = _rows(rng, n, d, 2.0)
    opt = HipSearchOptimizer(SearchConfig(dimension=d, metric=MetricType.EUCLIDEAN, collection_name="c"))
    opt.upsert([list(map(float, r)) for r in X], [{"chunk_id": f"c{i}", "document_id": i % 7, "text": "t"} for i in range(n)],
               [f"p{i}" for i in range(n)])
    q = X[5] + 0.01 * _rows(rng, 1, d)[0]
    dist = np.sqrt(((X.astype(np.float64) - q.astype(np.float64)) ** 2).sum(1))
    order = np.argsort(dist, kind="stable")
    hits = _run(opt.search(list(map(float, q)), limit=10, min_score=None))
    assert [c.chunk_id for c, _ in hits] == [f"c{i}" for i in order[:10]]
    assert np.allclose([s for _, s in hits], dist[order[:10]], rtol=1e-4, atol=1e-4)
    cut = float(dist[order[4]] + dist[order[5]]) / 2
    hits = _run(opt.search(list(map(float, q)), limit=10, min_score=cut))
    assert [c.chunk_id for c, _ in hits] == [f"c{i}" for i in order[:5]]
    assert _run(opt.search(list(map(float, q)), limit=10)) == []                 # the default 0.0: exact matches only
    exact = _run(opt.search(list(map(float, X[9])), limit=10))
    assert [c.chunk_id for c, _ in exact] == ["c9"] and exact[0][1] < 1e-3
    both = _run(opt.batch_search([list(map(float, q)), list(map(float, X[9]))], limit=3, min_score=None))
    assert [c.chunk_id for c, _ in both[0]] == [f"c{i}" for i in order[:3]] and both[1][0][0].chunk_id == "c9"
    filt = _run(opt.search(list(map(float, q)), limit=5, min_score=None, filters={"document_id": 3}))
    want = [i for i in order if i % 7 == 3][:5]
    assert [c.chunk_id for c, _ in filt] == [f"c{i}" for i in want]
    opt.cleanup() if hasattr(opt, "cleanup") else None
    with pytest.raises(SearchError):
        HipSearchOptimizer(SearchConfig(dimension=d, metric=MetricType.EUCLIDEAN, index_type=IndexType.IVF, collection_name="c"))
