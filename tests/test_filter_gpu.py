"""Caller filters kept on the device (rag_filter_*): a search that takes its filter from a filter object must
answer exactly what the same search answers with the filter's bytes passed as row_mask, on every entry point that
takes one; a filter made for another row count is refused; the optimizer and the pooled collections upload a
repeated filter once."""
import asyncio

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("n,d,nq,k", [(6000, 384, 1, 10), (50_000, 768, 3, 20), (40_000, 256, 24, 10), (70_000, 128, 200, 10),
                                      (3000, 100, 2, 100)])
def test_filter_object_equals_row_mask_and_the_oracle(n, d, nq, k):
    from oracle import flat
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(n + nq)
    X, Q = _unit(rng, n, d), _unit(rng, nq, d)
    mask = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    f = idx.make_filter(mask)
    assert len(f) == n and f.device_ptr != 0
    D0, I0 = idx.search(Q, k, min_score=0.0, row_mask=mask)
    path = idx.last_path
    D1, I1 = idx.search(Q, k, min_score=0.0, row_mask=f)
    assert idx.last_path == path
    assert np.array_equal(I0, I1) and np.array_equal(D0, D1)
    Dr, Ir = flat.flat_ip_search(X, Q, k, min_score=0.0, row_mask=mask)
    assert flat.compare_topk(D1, I1, Dr, Ir) == []
    assert bool((mask[I1[I1 >= 0]] == 1).all())
    # soft-deleted rows and the filter together
    gone = rng.choice(n, n // 10, replace=False)
    idx.set_active(gone, False)
    both = mask.copy()
    both[gone] = 0
    D2, I2 = idx.search(Q, k, row_mask=f)
    Dr, Ir = flat.flat_ip_search(X, Q, k, row_mask=both)
    assert flat.compare_topk(D2, I2, Dr, Ir) == []
    f.close()
    idx.close()


def test_filter_object_on_device_tensors_and_ivf():
    import torch
    from oracle import ivf as oivf
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(77)
    n, d, nlist, nprobe, k = 30_000, 128, 64, 8, 10
    X, Q = _unit(rng, n, d), _unit(rng, 40, d)
    cent = X[rng.choice(n, nlist, replace=False)].copy()
    mask = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    idx = DeviceIndex(d, metric="ip", kind="ivf")
    idx.train_ivf(nlist, centroids=cent)
    idx.add(X)
    f = idx.make_filter(mask)
    for nq in (1, 40):                                   # per-query list scan, list-major batch scan
        D0, I0 = idx.search(Q[:nq], k, nprobe=nprobe, row_mask=mask)
        D1, I1 = idx.search(Q[:nq], k, nprobe=nprobe, row_mask=f)
        assert np.array_equal(I0, I1) and np.array_equal(D0, D1)
        Qd = torch.from_numpy(Q[:nq]).cuda()
        D2, I2 = idx.search(Qd, k, nprobe=nprobe, row_mask=f)
        assert np.array_equal(I2.cpu().numpy(), I1) and np.array_equal(D2.cpu().numpy(), D1)
    D1, I1 = idx.search(Q, k, nprobe=nprobe, row_mask=f)
    bad, _excused, _amb = oivf.ivf_check(D1, I1, X, cent, Q, k, nprobe, row_mask=mask)
    assert bad == [], bad[:5]
    f.close()
    idx.close()


def test_filter_object_in_segment_searches():
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(5)
    n, d = 8000, 384
    X, Q = _unit(rng, n, d), _unit(rng, 3, d)
    cuts = np.sort(rng.choice(np.arange(1, n), 30, replace=False))
    segs = [(int(a), int(b)) for a, b in zip(np.r_[0, cuts], np.r_[cuts, n])]
    mask = (rng.uniform(size=n) < 0.6).astype(np.uint8)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    f = idx.make_filter(mask)
    D0, I0 = idx.search_segments(Q, segs, 8, min_score=0.0, row_mask=mask)
    D1, I1 = idx.search_segments(Q, segs, 8, min_score=0.0, row_mask=f)
    assert np.array_equal(I0, I1) and np.array_equal(D0, D1)
    w = [0.4, 0.6, 0.3]
    r0 = idx.search_segments_fused(Q, w, segs, 8, 10, min_score=0.0, row_mask=mask)
    r1 = idx.search_segments_fused(Q, w, segs, 8, 10, min_score=0.0, row_mask=f)
    for a, b in zip(r0, r1):
        assert np.array_equal(a, b)
    assert bool((mask[r1[0]] == 1).all())
    f.close()
    idx.close()


def test_filter_for_another_row_count_is_refused():
    from rag_amd import DeviceFilter, DeviceIndex, RagHipError
    rng = np.random.default_rng(6)
    X = _unit(rng, 2000, 64)
    idx = DeviceIndex(64, metric="ip")
    idx.add(X)
    f = idx.make_filter(np.ones(2000, dtype=np.uint8))
    idx.add(X[:10])                                       # the index grew: the filter no longer fits
    with pytest.raises(RagHipError):
        idx.search(X[:1], 5, row_mask=f)
    with pytest.raises(RagHipError):
        idx.search_segments(X[:1], [(0, 100)], 5, row_mask=f)
    with pytest.raises(ValueError):
        idx.make_filter(np.ones(5, dtype=np.uint8))
    D, I = idx.search(X[:1], 5)                           # the index itself is unharmed
    assert I[0, 0] == 0
    empty = DeviceFilter(np.zeros(0, dtype=np.uint8))
    assert len(empty) == 0
    empty.close()
    empty.close()                                         # closing twice is harmless
    f.close()
    idx.close()


def test_optimizer_and_pool_upload_a_repeated_filter_once():
    from rag_amd import DeviceFilter
    from rag_amd.storage_manager import VectorStorageManager
    run = asyncio.get_event_loop().run_until_complete
    rng = np.random.default_rng(8)
    d, n = 96, 600
    x = _unit(rng, n, d)
    for pooled in (False, True):
        mgr = VectorStorageManager(pooled=pooled)
        run(mgr.create_index("docs", d))
        ids = [f"doc_1_chunk_{i}" for i in range(n)]
        meta = [{"document_id": 1, "user_id": 7 if i % 3 else 8, "chunk_id": ids[i], "text": str(i)} for i in range(n)]
        run(mgr.add_vectors("docs", x.tolist(), x.tolist(), meta, ids))
        q = x[4].tolist()                                 # row 4: user 7
        plain = run(mgr.search_vectors("docs", q, "content", 5))
        a = run(mgr.search_vectors("docs", q, "content", 5, None, {"user_id": 7}))
        b = run(mgr.search_vectors("docs", q, "content", 5, None, {"user_id": 7}))
        other = run(mgr.search_vectors("docs", q, "content", 5, None, {"user_id": 8}))
        assert plain[0]["metadata"]["chunk_id"] == ids[4]
        assert [h["metadata"]["chunk_id"] for h in a] == [h["metadata"]["chunk_id"] for h in b]
        assert all(h["metadata"]["user_id"] == 7 for h in a) and a[0]["metadata"]["chunk_id"] == ids[4]
        assert other and all(h["metadata"]["user_id"] == 8 for h in other)
        opt = mgr.search_optimizers["docs_content"]
        if pooled:
            f1 = opt.pool.filter_mask([opt.collection_name], {"user_id": 7})
            f2 = opt.pool.filter_mask([opt.collection_name], {"user_id": 7})
        else:
            f1, f2 = opt._mask({"user_id": 7}), opt._mask({"user_id": 7})
        assert isinstance(f1, DeviceFilter) and f1 is f2
        # the collection changes: the next search makes a new filter and sees the new row
        run(mgr.add_vectors("docs", [x[4].tolist()], [x[4].tolist()], [{"document_id": 1, "user_id": 7, "chunk_id": "new", "text": "n"}],
                            ["doc_1_chunk_new"]))
        c = run(mgr.search_vectors("docs", q, "content", 5, None, {"user_id": 7}))
        assert {h["metadata"]["chunk_id"] for h in c[:2]} == {ids[4], "doc_1_chunk_new"}
        run(mgr.cleanup())
