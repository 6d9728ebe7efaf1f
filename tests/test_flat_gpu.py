"""Parity of the HIP exhaustive search (through the C ABI) with the CPU oracle.

Tolerance (stated by the target): ids equal except where scores tie within the
tolerance; scores within 1e-5 relative (+1e-6 absolute for scores near zero)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def _check(D, I, X, Q, k, **kw):
    from oracle import flat
    Dr, Ir = flat.flat_ip_search(X, Q, k, **kw)
    bad = flat.compare_topk(D, I, Dr, Ir, rtol=RTOL, atol=ATOL)
    assert bad == [], bad[:5]


@pytest.mark.parametrize("n,d", [(100_000, 384), (50_000, 768), (20_000, 1024), (30_000, 512),
                                 (10_000, 256), (9_999, 128), (5_000, 100), (3_000, 10), (777, 64),
                                 (4_321, 1536), (2_000, 36)])
def test_flat_ip_matches_oracle(n, d):
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(n + d)
    X = _unit(rng, n, d)
    Q = _unit(rng, 13, d)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X[: n // 2])
    idx.add(X[n // 2:])
    assert len(idx) == n
    for nq in (1, 2, 3, 8, 13):
        D, I = idx.search(Q[:nq], 10)
        assert D.dtype == np.float32 and I.dtype == np.int64 and D.shape == (nq, 10)
        _check(D, I, X, Q[:nq], 10)
        if nq <= 4:
            from paths import flat_fp32_path
            assert idx.last_path == flat_fp32_path(n, d, nq, 10)
    idx.close()


@pytest.mark.parametrize("k", [1, 2, 10, 33, 64, 65, 100, 200, 400, 1024])
def test_k_range(k):
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(k)
    n, d = 60_000, 384
    X = _unit(rng, n, d)
    Q = _unit(rng, 3, d)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    D, I = idx.search(Q, k)
    _check(D, I, X, Q, k)
    idx.close()


def test_small_and_empty():
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(3)
    d = 48
    idx = DeviceIndex(d, metric="ip")
    q = _unit(rng, 2, d)
    D, I = idx.search(q, 5)
    assert np.all(I == -1) and np.all(D == -np.finfo(np.float32).max)
    X = _unit(rng, 3, d)
    idx.add(X)
    D, I = idx.search(q, 5)                      # k > n: padded
    _check(D, I, X, q, 5)
    assert np.all(I[:, 3:] == -1)
    D, I = idx.search(q, 100)                    # k > 64 and > n
    _check(D, I, X, q, 100)
    idx.close()


def test_ties_lower_id_first_and_health_check_vectors():
    from rag_amd import DeviceIndex
    # vector_db/health_checker.py:228-249: two parallel vectors, both cosine 1.0
    idx = DeviceIndex(384, metric="cosine")
    idx.add(np.array([[0.1] * 384, [0.2] * 384], dtype=np.float32))
    D, I = idx.search(np.array([0.15] * 384, dtype=np.float32), 2)
    assert I.tolist() == [[0, 1]]
    assert np.allclose(D, 1.0, atol=1e-6)
    idx.close()
    rng = np.random.default_rng(9)
    base = _unit(rng, 50, 768)
    X = np.tile(base, (40, 1))                   # every row repeated 40 times
    idx = DeviceIndex(768, metric="ip")
    idx.add(X)
    D, I = idx.search(base[7], 40)
    assert I[0].tolist() == [7 + 50 * j for j in range(40)]
    assert np.all(D[0] == D[0, 0])
    D, I = idx.search(base[7], 100)
    assert I[0, :40].tolist() == [7 + 50 * j for j in range(40)]
    idx.close()


def test_mask_min_score_ids_and_soft_delete():
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(11)
    n, d = 40_000, 768
    X = _unit(rng, n, d)
    Q = _unit(rng, 4, d)
    ids = rng.permutation(10 * n)[:n].astype(np.int64) + (1 << 40)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X, ids=ids)
    D, I = idx.search(Q, 10)
    _check(D, I, X, Q, 10, ids=ids)
    mask = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    D, I = idx.search(Q, 10, row_mask=mask)
    _check(D, I, X, Q, 10, ids=ids, row_mask=mask)
    D0, I0 = idx.search(Q[:1], 10)
    thr = float(D0[0, 4]) - 1e-7
    D, I = idx.search(Q[:1], 10, min_score=thr)
    assert (I[0] >= 0).sum() == 5 and np.array_equal(I[0, :5], I0[0, :5])
    assert np.all(D[0, 5:] == -np.finfo(np.float32).max)
    # soft delete (vector_db/qdrant_client.py:826-883) then restore
    from oracle import flat
    _, Ipos = flat.flat_ip_search(X, Q[:1], 3)
    idx.set_active(Ipos[0], False)
    m2 = np.ones(n, dtype=np.uint8)
    m2[Ipos[0]] = 0
    D, I = idx.search(Q, 10)
    _check(D, I, X, Q, 10, ids=ids, row_mask=m2)
    D, I = idx.search(Q, 10, row_mask=mask)
    _check(D, I, X, Q, 10, ids=ids, row_mask=mask & m2)
    idx.set_active(Ipos[0], True)
    D, I = idx.search(Q, 10)
    _check(D, I, X, Q, 10, ids=ids)
    idx.close()


def test_cosine_metric_normalises_rows_and_queries():
    from oracle import flat
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(13)
    n, d = 20_000, 384
    X = (rng.standard_normal((n, d)) * rng.uniform(0.1, 5.0, (n, 1))).astype(np.float32)
    X[17] = 0.0
    Q = (rng.standard_normal((5, d)) * 3.0).astype(np.float32)
    idx = DeviceIndex(d, metric="cosine")
    idx.add(X)
    D, I = idx.search(Q, 10)
    Xn, Qn = flat.l2_normalize_rows(X), flat.l2_normalize_rows(Q)
    _check(D, I, Xn, Qn, 10)
    # reconstruct returns the stored (normalised) rows
    R = idx.reconstruct([0, 17, n - 1])
    assert np.allclose(R, Xn[[0, 17, n - 1]], atol=1e-6)
    # and agrees with the reference's brute-force cosine scan on a small slice
    ref = flat.embedded_cosine_scan(X[:300], Q[0], 5)
    small = DeviceIndex(d, metric="cosine")
    small.add(X[:300])
    Ds, Is = small.search(Q[0], 5)
    assert Is[0].tolist() == [i for i, _ in ref]
    assert np.allclose(Ds[0], [s for _, s in ref], rtol=RTOL, atol=ATOL)
    idx.close()
    small.close()


def test_torch_device_path_and_adopt():
    import torch
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(17)
    n, d = 30_000, 768
    X = _unit(rng, n, d)
    Q = _unit(rng, 6, d)
    xt = torch.from_numpy(X).cuda()
    qt = torch.from_numpy(Q).cuda()
    idx = DeviceIndex(d, metric="ip")
    idx.adopt(xt)
    D, I = idx.search(qt, 10)
    assert D.is_cuda and I.is_cuda
    torch.cuda.synchronize()
    _check(D.cpu().numpy(), I.cpu().numpy(), X, Q, 10)
    m = torch.from_numpy((rng.uniform(size=n) < 0.5).astype(np.uint8)).cuda()
    D, I = idx.search(qt, 10, row_mask=m)
    _check(D.cpu().numpy(), I.cpu().numpy(), X, Q, 10, row_mask=m.cpu().numpy())
    idx.close()
    idx2 = DeviceIndex(d, metric="ip")
    idx2.set_id_base(1_000_000)
    idx2.add(xt)
    D, I = idx2.search(qt[:1], 5)
    _check(D.cpu().numpy(), I.cpu().numpy() - 1_000_000, X, Q[:1], 5)
    idx2.close()


@pytest.mark.parametrize("d", [768, 100])
def test_every_small_batch_size(d):
    """Up to 4 queries are one scan pass (3 use 4 slots); from 5 the narrow MFMA shape takes
    over when the row stride allows it.  d = 100 has no MFMA path: 5..7 queries use 8 slots and
    13 queries are two scan passes, the second one padded."""
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(31 + d)
    n, k = 50_000, 10
    X = _unit(rng, n, d)
    Q = _unit(rng, 13, d)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    for nq in (1, 2, 3, 4, 5, 6, 7, 8, 13):
        D, I = idx.search(Q[:nq], k)
        from paths import flat_fp32_path
        assert idx.last_path == flat_fp32_path(n, d, nq, k)
        _check(D, I, X, Q[:nq], k)
    idx.close()


def test_searches_on_different_streams_are_ordered():
    """A device search queued on torch's stream, a host-pointer search (the index's
    own stream) and a search on a side stream share the index workspaces; the
    library orders them, so none sees the other's queries or partial lists."""
    import torch
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(23)
    n, d, k = 400_000, 768, 10
    X = _unit(rng, n, d)
    Q = _unit(rng, 9, d)
    xt = torch.from_numpy(X).cuda()
    qt = torch.from_numpy(Q).cuda()
    idx = DeviceIndex(d, metric="ip")
    idx.adopt(xt)
    side = torch.cuda.Stream()
    for _ in range(5):
        Da, Ia = idx.search(qt[:3], k)                       # async, torch's current stream
        Dh, Ih = idx.search(Q[3:6], k)                       # host API, index stream, blocking
        with torch.cuda.stream(side):
            Ds, Is = idx.search(qt[6:9], k)                  # async, side stream
        Dseg, Iseg = idx.search_segments(Q[:2], [(0, n)], k)
        torch.cuda.synchronize()
        _check(Da.cpu().numpy(), Ia.cpu().numpy(), X, Q[:3], k)
        _check(Dh, Ih, X, Q[3:6], k)
        _check(Ds.cpu().numpy(), Is.cpu().numpy(), X, Q[6:9], k)
        _check(Dseg[:, 0], Iseg[:, 0], X, Q[:2], k)
    idx.close()


def test_searches_alternating_between_streams_many_times():
    """The end of a search is marked only when a search on another stream needs it (torch's streams live on, which
    DeviceIndex tells the library): two side streams and torch's own take turns on one index, two queries at a time
    through the same workspaces, nothing synchronised in between; every answer must be the one a lone search gives."""
    import torch
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(29)
    n, d, k = 300_000, 384, 10
    X = _unit(rng, n, d)
    Q = _unit(rng, 48, d)
    xt = torch.from_numpy(X).cuda()
    qt = torch.from_numpy(Q).cuda()
    idx = DeviceIndex(d, metric="ip")
    idx.adopt(xt)
    alone = [idx.search(qt[i:i + 2], k) for i in range(0, 48, 2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()]
    got = []
    for j, i in enumerate(range(0, 48, 2)):
        with torch.cuda.stream(streams[(j * 2) % 3 if j % 4 else j % 3]):      # same stream twice in a row now and then
            got.append(idx.search(qt[i:i + 2], k))
    torch.cuda.synchronize()
    for (Da, Ia), (Dg, Ig) in zip(alone, got):
        assert torch.equal(Ia, Ig) and torch.equal(Da, Dg)
    _check(got[5][0].cpu().numpy(), got[5][1].cpu().numpy(), X, Q[10:12], k)
    idx.close()


def test_merge_topk_device():
    import torch
    from rag_amd import DeviceIndex, merge_topk
    rng = np.random.default_rng(19)
    n, d, G = 32_000, 384, 4
    X = _unit(rng, n, d)
    Q = _unit(rng, 9, d)
    parts = []
    for g in range(G):
        sh = DeviceIndex(d, metric="ip")
        lo, hi = g * n // G, (g + 1) * n // G
        sh.set_id_base(lo)
        sh.add(X[lo:hi])
        parts.append(sh.search(torch.from_numpy(Q).cuda(), 10))
        sh.close()
    D = torch.stack([p[0] for p in parts])
    I = torch.stack([p[1] for p in parts])
    Dm, Im = merge_topk(D, I)
    _check(Dm.cpu().numpy(), Im.cpu().numpy(), X, Q, 10)


def test_errors_are_reported():
    from rag_amd import DeviceIndex, RagHipError
    idx = DeviceIndex(16, metric="ip")
    with pytest.raises(ValueError):
        idx.search(np.zeros((1, 16), np.float32), 5000)
    with pytest.raises(ValueError):
        idx.add(np.zeros((3, 8), np.float32))
    with pytest.raises(RagHipError):
        idx.reconstruct([5])
    with pytest.raises(RagHipError):
        DeviceIndex(16, device=99)
    idx.close()


def test_gather_slot_layout_and_strided_merge():
    """What ShardedIndex does per step on N GPUs, replayed on one: every shard
    writes (ids | scores) into its slot of the gather buffer, the merge reads the
    slots through list strides."""
    import torch
    from rag_amd import DeviceIndex
    from rag_amd.index import merge_topk_strided
    rng = np.random.default_rng(31)
    n, d, G, k = 30_011, 768, 3, 10
    X = _unit(rng, n, d)
    for nq in (1, 5):
        Q = _unit(rng, nq, d)
        qt = torch.from_numpy(Q).cuda()
        nk = nq * k
        slot = nk + (nk + 1) // 2
        buf = torch.empty((G, slot), dtype=torch.int64, device="cuda")
        shards = []
        for g in range(G):
            from rag_amd.sharded import shard_range
            lo, hi = shard_range(n, g, G)
            sh = DeviceIndex(d, metric="ip")
            sh.set_id_base(lo)
            sh.add(X[lo:hi])
            sh.search_into(qt, k, buf[g].data_ptr() + nk * 8, buf[g].data_ptr())
            shards.append(sh)
        Dm, Im = merge_topk_strided(G, nq, k, buf.data_ptr() + nk * 8, slot * 2, buf.data_ptr(), slot, qt.device)
        torch.cuda.synchronize()
        _check(Dm.cpu().numpy(), Im.cpu().numpy(), X, Q, k)
        for sh in shards:
            sh.close()


def test_host_threads_may_share_an_index():
    """The reference's callers reach the backend from several threads (the MMR
    thread pool of vector_db/mmr_diversifier.py:58, uvicorn workers): entry points
    that take an index serialise on it, so concurrent calls give the answers of
    sequential ones."""
    import threading
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(77)
    n, d = 60000, 384
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    jobs = []
    for t in range(6):
        nq = (1, 3, 9, 40, 1, 2)[t]
        jobs.append((rng.standard_normal((nq, d)).astype(np.float32), (10, 100, 10, 20, 200, 5)[t]))
    want = [idx.search(q, k) for q, k in jobs]
    segs = [(0, 1000), (1000, 30000), (30000, n)]
    want_seg = idx.search_segments(jobs[1][0], segs, 7)
    errors = []

    def work(t):
        try:
            q, k = jobs[t]
            for _ in range(25):
                D, I = idx.search(q, k)
                if not (np.array_equal(I, want[t][1]) and np.array_equal(D, want[t][0])):
                    errors.append(("search", t))
                if t == 1:
                    Ds, Is = idx.search_segments(q, segs, 7)
                    if not (np.array_equal(Is, want_seg[1]) and np.array_equal(Ds, want_seg[0])):
                        errors.append(("segments", t))
                if t == 4:
                    v = idx.reconstruct([5, 59999])
                    if not np.array_equal(v, X[[5, 59999]]):
                        errors.append(("reconstruct", t))
        except Exception as e:                                   # noqa: BLE001
            errors.append(("raised", t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    assert not any(th.is_alive() for th in threads)
    assert errors == []
    idx.close()


def test_device_search_can_be_captured_in_a_hip_graph():
    """rag_index_search_device enqueues kernels and memsets only once its workspaces have their size (a warm-up
    call on the stream that is captured): the call can be recorded into a HIP graph and replayed with new query
    contents, flat and IVF, with the answers of the direct call."""
    import torch
    from rag_amd import DeviceIndex
    dev = torch.device("cuda", 0)
    g0 = torch.Generator(device=dev)
    g0.manual_seed(5)
    n, d, k = 60_000, 256, 10
    X = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g0), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(8, d, device=dev, generator=g0), dim=1)
    flat_idx = DeviceIndex(d, metric="ip")
    flat_idx.add(X)
    ivf_idx = DeviceIndex(d, metric="ip", kind="ivf")
    ivf_idx.train_ivf(64, centroids=X[:64].cpu().numpy())
    ivf_idx.add(X)
    ivf_idx.search(Q[:1], k, nprobe=8)                     # packs the lists (synchronises: not part of a capture)
    for idx, kw in ((flat_idx, {}), (ivf_idx, {"nprobe": 8})):
        q = Q[0:1].clone()
        D = torch.empty((1, k), dtype=torch.float32, device=dev)
        I = torch.empty((1, k), dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            idx.search_into(q, k, D.data_ptr(), I.data_ptr(), **kw)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            idx.search_into(q, k, D.data_ptr(), I.data_ptr(), **kw)
        for j in range(1, 8):
            q.copy_(Q[j:j + 1])
            graph.replay()
            torch.cuda.synchronize()
            Dd, Id = idx.search(Q[j:j + 1], k, **kw)
            assert torch.equal(I, Id) and torch.equal(D, Dd), (idx.kind, j)
        del graph
        idx.close()
