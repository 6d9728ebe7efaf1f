"""rag_multi_*: one index over several GPUs of one process (here: several shards on one GPU).
Against the CPU oracle with the tolerance of the exhaustive search, and against a single
DeviceIndex holding the same rows under the same tolerance: a shard's row count picks its own
scoring kernel (another summation order), so a score may differ from the one-index score in the
last bit; exact ties across shards still go to the lower id."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("shards,n,d,nq,k", [(2, 30_000, 128, 1, 10), (3, 50_000, 384, 40, 10), (4, 20_003, 100, 7, 100),
                                             (8, 40_000, 768, 300, 10)])
def test_multi_device_flat_equals_oracle_and_single_index(shards, n, d, nq, k):
    from oracle import flat
    from rag_amd import DeviceIndex, MultiDeviceIndex
    rng = np.random.default_rng(shards * 1000 + nq)
    X = _unit(rng, n, d)
    X[n // 2] = X[7]                                        # an exact tie across shards: the lower row wins
    Q = _unit(rng, nq, d)
    Q[0] = X[7]
    multi = MultiDeviceIndex(d, [0] * shards, metric="ip")
    single = DeviceIndex(d, metric="ip")
    for lo, hi in ((0, n // 3), (n // 3, n // 3 + 5), (n // 3 + 5, n)):      # a 5-row add leaves some shards without a part
        multi.add(X[lo:hi])
        single.add(X[lo:hi])
    assert len(multi) == n and sum(multi.shard_rows()) == n and min(multi.shard_rows()) > 0
    D, I = multi.search(Q, k)
    assert multi.last_merge == "device"
    Ds, Is = single.search(Q, k)
    assert flat.compare_topk(D, I, Ds, Is, rtol=RTOL, atol=ATOL) == []
    assert I[0, 0] == 7 and I[0, 1] == n // 2
    bad = flat.compare_topk(D, I, *flat.flat_ip_search(X, Q, k), rtol=RTOL, atol=ATOL)
    assert bad == [], bad[:5]
    # mask, threshold, soft delete and reconstruct count rows in insertion order over all shards
    mask = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    D, I = multi.search(Q, k, min_score=0.05, row_mask=mask)
    Ds, Is = single.search(Q, k, min_score=0.05, row_mask=mask)
    assert flat.compare_topk(D, I, Ds, Is, rtol=RTOL, atol=ATOL) == []
    assert not np.isin(np.flatnonzero(mask == 0), I).any()
    # the same bytes as a filter object: cut into the shards' orders once, kept on their devices
    flt = multi.make_filter(mask)
    Df, If = multi.search(Q, k, min_score=0.05, row_mask=flt)
    assert np.array_equal(Df, D) and np.array_equal(If, I)
    hide = np.unique(I[I >= 0])[:20]
    multi.set_active(hide, False)
    single.set_active(hide, False)
    D, I = multi.search(Q, k, row_mask=mask)
    Ds, Is = single.search(Q, k, row_mask=mask)
    assert flat.compare_topk(D, I, Ds, Is, rtol=RTOL, atol=ATOL) == [] and not np.isin(hide, I).any()
    Df, If = multi.search(Q, k, row_mask=flt)               # soft deletes apply on top of a filter object as well
    assert np.array_equal(Df, D) and np.array_equal(If, I)
    rows = rng.integers(0, n, 64)
    assert np.array_equal(multi.reconstruct(rows), X[rows])
    multi.add(X[:3])                                         # the index grew: the filter no longer fits
    from rag_amd import RagHipError
    with pytest.raises(RagHipError):
        multi.search(Q, k, row_mask=flt)
    flt.close()
    multi.close()
    single.close()


def test_multi_device_ivf_with_shared_centroids_is_one_ivf_index():
    from oracle import ivf
    from rag_amd import MultiDeviceIndex
    rng = np.random.default_rng(5)
    n, d, nlist, nprobe, k = 60_000, 128, 64, 8, 10
    X = _unit(rng, n, d)
    C = X[rng.choice(n, nlist, replace=False)].copy()
    ids = np.arange(n, dtype=np.int64) * 2 + 5
    multi = MultiDeviceIndex(d, [0, 0, 0], metric="ip", kind="ivf")
    multi.train_ivf(nlist, centroids=C)
    multi.add(X[:25_000], ids=ids[:25_000])
    multi.add(X[25_000:], ids=ids[25_000:])
    for nq in (1, 300):
        Q = _unit(rng, nq, d)
        D, I = multi.search(Q, k, nprobe=nprobe)
        bad, excused, amb = ivf.ivf_check(D, I, X, C, Q, k, nprobe, ids=ids, rtol=RTOL, atol=ATOL)
        assert bad == [], bad[:5]
    multi.close()


def test_device_merge_equals_host_merge(monkeypatch):
    """The devices' lists merged by the kernel on the first device (peer copies) and by the host loop
    (what a box without peer access falls back to) are the same bits: scores, ids, padding; with ties
    across shards whose ids interleave, short lists (min_score), k up to 1024, and the L2 metric."""
    from rag_amd import MultiDeviceIndex
    rng = np.random.default_rng(77)
    n, d = 24_000, 96
    X = _unit(rng, n, d)
    X[5::1000] = X[5]                                       # 24 copies of one row, spread over every add and shard
    Q = _unit(rng, 33, d)
    Q[0] = X[5]
    for metric in ("ip", "l2"):
        got = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("RAG_MULTI_HOST_MERGE", mode)
            m = MultiDeviceIndex(d, [0, 0, 0, 0, 0], metric=metric)
            for lo in range(0, n, 6000):                    # four adds: a shard holds four separate id ranges
                m.add(X[lo:lo + 6000])
            res = []
            for k, kw in ((10, {}), (40, {"min_score": 0.38 if metric == "ip" else 1.24}), (1024, {}), (3, {"row_mask": (np.arange(n) % 3 > 0).astype(np.uint8)})):
                res.append(m.search(Q, k, **kw))
                assert m.last_merge == ("host" if mode == "1" else "device")
            got[mode] = res
            m.close()
        for (Da, Ia), (Db, Ib) in zip(got["0"], got["1"]):
            assert np.array_equal(Ia, Ib) and np.array_equal(Da, Db)
        D, I = got["0"][0]
        assert I[0, :10].tolist() == list(range(5, 10_000, 1000))      # equal scores: ids ascending
        D, I = got["0"][1]
        assert (I == -1).any() and (I >= 0).any()                       # the threshold leaves short lists, padded


def test_a_failed_add_is_taken_back_on_every_device(monkeypatch):
    """An add that fails on one device leaves no rows behind on the others: row counts, ids, searches, masks
    and the next add behave as if the failed call had not been made (flat and IVF)."""
    from oracle import flat
    from rag_amd import MultiDeviceIndex, RagHipError
    rng = np.random.default_rng(8)
    n, d = 9000, 64
    X = _unit(rng, n, d)
    Q = _unit(rng, 5, d)
    for kind in ("flat", "ivf"):
        m = MultiDeviceIndex(d, [0, 0, 0], metric="ip", kind=kind)
        if kind == "ivf":
            m.train_ivf(8, centroids=X[:8].copy())
        m.add(X[:4000])
        before = m.shard_rows()
        monkeypatch.setenv("RAG_TEST_MULTI_FAIL_ADD", "1")
        with pytest.raises(RagHipError) as e:
            m.add(_unit(rng, 3000, d))                      # shards 0 and 2 take their parts, shard 1 reports a failure
        assert "injected" in str(e.value) and "device 0" in str(e.value)
        monkeypatch.delenv("RAG_TEST_MULTI_FAIL_ADD")
        assert len(m) == 4000 and m.shard_rows() == before
        m.add(X[4000:])                                     # rows 4000.. get the ids the failed call had offered
        assert len(m) == n
        kw = {"nprobe": 8} if kind == "ivf" else {}
        mask = (rng.uniform(size=n) < 0.7).astype(np.uint8)
        D, I = m.search(Q, 10, row_mask=mask, **kw)
        bad = flat.compare_topk(D, I, *flat.flat_ip_search(X, Q, 10, row_mask=mask), rtol=RTOL, atol=ATOL)
        assert bad == [], bad[:3]
        rows = rng.integers(0, n, 50)
        assert np.array_equal(m.reconstruct(rows), X[rows])
        m.close()


def test_multi_device_errors_name_the_device():
    from rag_amd import MultiDeviceIndex, RagHipError
    with pytest.raises(RagHipError):
        MultiDeviceIndex(64, [0, 99])
    m = MultiDeviceIndex(64, [0, 0], kind="ivf")
    with pytest.raises(RagHipError) as e:
        m.add(np.zeros((10, 64), np.float32))               # not trained
    assert "device 0" in str(e.value)
    assert len(m) == 0
    m.close()


def test_thread_staging_memory_can_be_handed_back():
    """rag_thread_release frees the calling thread's staging blocks; the next call allocates them again."""
    import threading
    import torch
    from rag_amd import DeviceIndex, _lib
    rng = np.random.default_rng(9)
    d = 512
    X = _unit(rng, 8000, d)                                # 8000 queries x 512 floats: a 16 MB staging block
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    D0, I0 = idx.search(X, 5)
    out = []

    def worker():
        free0, _ = torch.cuda.mem_get_info(0)
        out.append(idx.search(X, 5))                       # this thread's staging block appears ...
        free1, _ = torch.cuda.mem_get_info(0)
        assert _lib.lib.rag_thread_release() == 0          # ... and goes away
        free2, _ = torch.cuda.mem_get_info(0)
        out.append((free0, free1, free2))
        out.append(idx.search(X, 5))                       # and the thread can go on searching

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    (D1, I1), (free0, free1, free2), (D2, I2) = out
    assert np.array_equal(I0, I1) and np.array_equal(I0, I2) and np.array_equal(D0, D2)
    assert free0 - free1 >= 8 << 20 and free2 - free1 >= 8 << 20
    idx.close()


def test_multi_device_hybrid_mmr_equals_one_index():
    """top-N -> hybrid fusion -> MMR with the corpus dealt over several shards: candidates' vectors come
    from the shards that hold them, and the answer is that of the one-index call bit for bit."""
    from rag_amd import DeviceIndex, MultiDeviceIndex
    from rag_amd.index import hybrid_mmr
    rng = np.random.default_rng(31)
    n, d = 120_000, 256
    cent = rng.standard_normal((30, d))
    X = cent[rng.integers(0, 30, n)] + 0.2 * rng.standard_normal((n, d))
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    one = DeviceIndex(d, metric="cosine")
    one.add(X)
    multi = MultiDeviceIndex(d, [0, 0, 0], metric="cosine")
    multi.add(X[:50_000])
    multi.add(X[50_000:])
    assert min(multi.shard_rows()) > 32768                     # every shard on the scan path, like the one index
    for trial in range(3):
        q = (cent[trial] / np.linalg.norm(cent[trial])).astype(np.float32)
        _, I = one.search(q, 200)
        bm_rows = np.concatenate([rng.choice(I[0], 50, replace=False), rng.integers(0, n, 50), [-1, n]])
        bm_scores = rng.gamma(2.0, 2.0, size=bm_rows.size)
        for kw in ({"mmr_lambda": 0.5, "mmr_threshold": 0.8}, {"mmr_lambda": 0.3, "mmr_threshold": 0.5, "max_results": 30},
                   {"mmr_metric": "euclidean", "mmr_threshold": 0.4, "fuse_limit": 64}):
            a = multi.hybrid_mmr(q, 200, bm_rows, bm_scores, want_fused=True, **kw)
            b = hybrid_mmr(one, q, 200, bm_rows, bm_scores, want_fused=True, **kw)
            assert a["status"] == b["status"] == 0 and a["n_fused"] == b["n_fused"] and len(a["ids"]) > 0
            for name in ("ids", "scores", "picked", "fused_rows", "fused_scores", "fused_sem", "fused_bm"):
                assert np.array_equal(np.asarray(a[name]), np.asarray(b[name])), (trial, kw, name)
    # a zero maximum on the keyword side: the semantic list, on both
    a = multi.hybrid_mmr(q, 50, [1, 2, 3], [0.0, 0.0, 0.0])
    b = hybrid_mmr(one, q, 50, [1, 2, 3], [0.0, 0.0, 0.0])
    assert a["status"] == b["status"] == 1 and np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["scores"], b["scores"])
    with_ids = MultiDeviceIndex(d, [0, 0], metric="ip")
    with_ids.add(X[:2000], ids=np.arange(2000, dtype=np.int64) + 7)
    with pytest.raises(ValueError):
        with_ids.hybrid_mmr(q, 10, [], [])
    for ix in (one, multi, with_ids):
        ix.close()
