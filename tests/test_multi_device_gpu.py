"""rag_multi_*: one index over several GPUs of one process (here: several shards on one GPU).
Against the CPU oracle with the tolerance of the exhaustive search, and against a single
DeviceIndex holding the same rows (same ids, scores equal bit for bit: a row's score does not
depend on which shard scans it)."""
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
    Ds, Is = single.search(Q, k)
    assert np.array_equal(I, Is) and np.array_equal(D, Ds)
    assert I[0, 0] == 7 and I[0, 1] == n // 2
    bad = flat.compare_topk(D, I, *flat.flat_ip_search(X, Q, k), rtol=RTOL, atol=ATOL)
    assert bad == [], bad[:5]
    # mask, threshold, soft delete and reconstruct count rows in insertion order over all shards
    mask = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    D, I = multi.search(Q, k, min_score=0.05, row_mask=mask)
    Ds, Is = single.search(Q, k, min_score=0.05, row_mask=mask)
    assert np.array_equal(I, Is) and np.array_equal(D, Ds)
    hide = np.unique(I[I >= 0])[:20]
    multi.set_active(hide, False)
    single.set_active(hide, False)
    D, I = multi.search(Q, k, row_mask=mask)
    Ds, Is = single.search(Q, k, row_mask=mask)
    assert np.array_equal(I, Is) and not np.isin(hide, I).any()
    rows = rng.integers(0, n, 64)
    assert np.array_equal(multi.reconstruct(rows), X[rows])
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
