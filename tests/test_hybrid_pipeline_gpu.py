"""search top-N -> hybrid semantic + BM25 fusion -> MMR as ONE library call
(rag_index_hybrid_mmr[_device]) against the oracle's restatement of
vector_db/search_manager.py:1023-1101 + :969-970 and vector_db/mmr_diversifier.py:151-289.

The semantic stage is checked against the exhaustive-search oracle with the usual tolerance; the
fusion and MMR stages are then checked EXACTLY (bit-identical fused scores, identical orderings)
given the scores that stage produced, since one ulp in an fp32 score legitimately moves a fused
fp64 score."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6


def _clustered(rng, n, d, ncl=40, noise=0.2):
    cent = rng.standard_normal((ncl, d))
    X = cent[rng.integers(0, ncl, n)] + noise * rng.standard_normal((n, d))
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    return X, cent


def _oracle_tail(X, D, rows, bm_rows, bm_scores, fuse_limit, ws, wb, lam, thr, metric, max_results):
    """fusion + MMR of the oracle given the semantic stage's (scores, row positions)."""
    from oracle import fusion as of
    from oracle import mmr as om
    n = X.shape[0]
    sem = [(int(r), float(s)) for s, r in zip(D, rows) if r >= 0]
    bm = [(int(r), float(s)) for r, s in zip(bm_rows, bm_scores) if 0 <= r < n]
    ranked = of.hybrid_rank(sem, bm, fuse_limit, ws, wb)
    status = 0
    if ranked is None:                     # zero maximum: the semantic list, sorted (it is) and cut
        status = 1
        ranked = [(r, s, s, 0.0, True, False) for r, s in sem][:fuse_limit]
    f_rows = [t[0] for t in ranked]
    f_scores = [t[1] for t in ranked]
    picked = om.mmr_select_fast(X[f_rows].astype(np.float64), f_scores, lam, max_results, thr, metric) if f_rows else []
    return f_rows, f_scores, [t[2] for t in ranked], [t[3] for t in ranked], picked, status


def _check(idx, X, q, topn, bm_rows, bm_scores, *, fuse_limit=None, max_results=10, lam=0.5, thr=0.8, metric="cosine",
           ws=0.65, wb=0.35, ids=None, id_base=0, row_mask=None, min_score=None, nprobe=1, Xsearch=None):
    from oracle import flat
    from rag_amd.index import hybrid_mmr
    n = X.shape[0]
    got = hybrid_mmr(idx, q, topn, bm_rows, bm_scores, fuse_limit=fuse_limit, max_results=max_results,
                     semantic_weight=ws, bm25_weight=wb, mmr_lambda=lam, mmr_threshold=thr, mmr_metric=metric,
                     min_score=min_score, nprobe=nprobe, row_mask=row_mask, want_fused=True)
    # the semantic stage, as a plain search: within tolerance of the oracle ...
    D, I = idx.search(q, topn, min_score=min_score, nprobe=nprobe, row_mask=row_mask)
    if idx.kind == "flat":
        Xs = X if Xsearch is None else Xsearch
        Dr, Ir = flat.flat_ip_search(Xs, q[None, :] if Xsearch is None else (q / np.linalg.norm(q))[None, :].astype(np.float32),
                                     topn, row_mask=row_mask, min_score=min_score, ids=ids)
        if ids is None and id_base:
            Ir = np.where(Ir >= 0, Ir + id_base, -1)
        assert flat.compare_topk(D, I, Dr, Ir, rtol=RTOL, atol=ATOL) == []
    # ... and the stages behind it exactly, given its scores
    if ids is not None:
        back = {int(v): r for r, v in enumerate(ids)}
        rows = np.array([back[int(i)] if i >= 0 else -1 for i in I[0]])
    else:
        rows = np.where(I[0] >= 0, I[0] - id_base, -1)
    fl = fuse_limit if fuse_limit is not None else min(1024, topn + len(bm_rows))
    f_rows, f_scores, f_sem, f_bm, picked, status = _oracle_tail(X, D[0], rows, bm_rows, bm_scores, fl, ws, wb, lam, thr,
                                                                   metric, max_results)
    assert got["status"] == status
    assert got["n_fused"] == len(f_rows)
    assert got["fused_rows"].tolist() == f_rows
    assert np.array_equal(got["fused_scores"], np.array(f_scores, dtype=np.float64))          # bit-exact
    if status == 0:
        assert np.array_equal(got["fused_sem"], np.array(f_sem)) and np.array_equal(got["fused_bm"], np.array(f_bm))
    assert got["picked"].tolist() == picked
    want_ids = [int(ids[f_rows[p]]) if ids is not None else f_rows[p] + id_base for p in picked]
    assert got["ids"].tolist() == want_ids
    assert np.array_equal(got["scores"], np.array([f_scores[p] for p in picked]))
    return got


def test_pipeline_matches_oracle_flat():
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(11)
    n, d = 40_000, 256
    X, cent = _clustered(rng, n, d)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    for trial in range(4):
        q = (cent[trial] / np.linalg.norm(cent[trial])).astype(np.float32)
        D, I = idx.search(q, 200)
        # keyword list: half of it among the semantic hits, half elsewhere; a repeated row (the later
        # score counts), rows outside the index (ignored)
        bm_rows = np.concatenate([rng.choice(I[0], 50, replace=False), rng.integers(0, n, 50), I[0][:2], [-5, n + 3]])
        bm_scores = rng.gamma(2.0, 2.0, size=bm_rows.size)
        for lam, thr, metric in ((0.5, 0.8, "cosine"), (0.3, 0.5, "cosine"), (0.6, 0.0, "dot_product"), (0.7, 0.4, "euclidean")):
            got = _check(idx, X, q, 200, bm_rows, bm_scores, lam=lam, thr=thr, metric=metric)
            assert len(got["ids"]) == 10
        _check(idx, X, q, 200, bm_rows, bm_scores, fuse_limit=60, max_results=25)      # cut before MMR
        _check(idx, X, q, 17, bm_rows[:5], bm_scores[:5], max_results=40)             # fewer candidates than picks
    idx.close()


def test_pipeline_edges():
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(12)
    n, d = 9_000, 128
    X, cent = _clustered(rng, n, d, ncl=12)
    q = (cent[1] / np.linalg.norm(cent[1])).astype(np.float32)
    ids = rng.permutation(10 * n)[:n].astype(np.int64)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X, ids=ids)
    none_r, none_s = np.zeros(0, np.int64), np.zeros(0)
    _check(idx, X, q, 100, none_r, none_s, ids=ids)                                   # no keyword side: divisor 1.0
    bm_rows = rng.integers(0, n, 30)
    _check(idx, X, q, 100, bm_rows, np.zeros(30), ids=ids)                            # zero maximum -> status 1
    _check(idx, X, q, 100, bm_rows, rng.uniform(size=30), ids=ids, min_score=0.6)     # few (or no) semantic hits
    _check(idx, X, q, 100, bm_rows, rng.uniform(size=30), ids=ids, min_score=2.0)     # no semantic hit at all
    mask = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    _check(idx, X, q, 64, bm_rows, rng.uniform(size=30), ids=ids, row_mask=mask)      # register-list scan (k <= 64)
    _check(idx, X, q, 1000, rng.integers(0, n, 1000), rng.uniform(size=1000), ids=ids, fuse_limit=1024, max_results=5)
    # equal fused scores everywhere: relevance 1.0 each, the first candidate leads
    same = np.tile(X[:1], (50, 1))
    idx2 = DeviceIndex(d, metric="ip")
    idx2.set_id_base(1000)
    idx2.add(same)
    _check(idx2, same, X[0], 20, none_r, none_s, id_base=1000, thr=0.0)
    idx.close()
    idx2.close()


def test_pipeline_on_ivf_and_cosine_indexes():
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(13)
    n, d, nlist = 30_000, 256, 32
    X, cent = _clustered(rng, n, d, ncl=25)
    q = (cent[2] / np.linalg.norm(cent[2])).astype(np.float32)
    C = X[rng.choice(n, nlist, replace=False)].copy()
    iv = DeviceIndex(d, metric="ip", kind="ivf")
    iv.train_ivf(nlist, centroids=C)
    iv.set_id_base(500)
    iv.add(X[:20_000])
    iv.add(X[20_000:])                          # the pipeline's own search re-packs before MMR reads rows
    bm_rows = rng.integers(0, n, 80)
    _check(iv, X, q, 150, bm_rows, rng.gamma(2.0, 2.0, 80), id_base=500, nprobe=8)
    _check(iv, X, q, 150, bm_rows, rng.gamma(2.0, 2.0, 80), id_base=500, nprobe=nlist, lam=0.4, thr=0.6)
    iv.close()
    # COSINE index: rows are stored normalised, so MMR reads the normalised rows
    Y = (X[:8000] * rng.uniform(0.5, 2.0, (8000, 1))).astype(np.float32)
    Yn = (Y / np.linalg.norm(Y, axis=1, keepdims=True)).astype(np.float32)
    cz = DeviceIndex(d, metric="cosine")
    cz.add(Y)
    stored = cz.reconstruct(np.arange(8000))
    _check(cz, stored, 3.0 * q, 120, rng.integers(0, 8000, 40), rng.gamma(2.0, 2.0, 40), Xsearch=Yn)
    cz.close()


def test_pipeline_device_variant_matches_host_variant_and_does_not_sync():
    import torch
    from rag_amd import DeviceIndex
    from rag_amd.index import hybrid_mmr
    rng = np.random.default_rng(14)
    n, d = 50_000, 384
    X, cent = _clustered(rng, n, d)
    idx = DeviceIndex(d, metric="ip")
    idx.adopt(torch.from_numpy(X).cuda())
    for trial in range(3):
        q = (cent[trial + 5] / np.linalg.norm(cent[trial + 5])).astype(np.float32)
        bm_rows = rng.integers(0, n, 100)
        bm_scores = rng.gamma(2.0, 2.0, 100)
        host = hybrid_mmr(idx, q, 200, bm_rows, bm_scores, mmr_lambda=0.5, want_fused=True)
        dev = hybrid_mmr(idx, torch.from_numpy(q).cuda(), 200, torch.from_numpy(bm_rows).cuda(),
                         torch.from_numpy(bm_scores).cuda(), mmr_lambda=0.5, want_fused=True)
        torch.cuda.synchronize()
        m, nf = int(dev["n_out"].item()), int(dev["n_fused"].item())
        assert m == len(host["ids"]) and nf == host["n_fused"] and int(dev["status"].item()) == host["status"]
        assert dev["ids"][:m].cpu().numpy().tolist() == host["ids"].tolist()
        assert np.array_equal(dev["scores"][:m].cpu().numpy(), host["scores"])
        assert dev["picked"][:m].cpu().numpy().tolist() == host["picked"].tolist()
        assert dev["fused_rows"][:nf].cpu().numpy().tolist() == host["fused_rows"].tolist()
        assert np.array_equal(dev["fused_scores"][:nf].cpu().numpy(), host["fused_scores"])
    idx.close()


def test_pipeline_argument_checks():
    from rag_amd import DeviceIndex, RagHipError
    from rag_amd.index import hybrid_mmr
    idx = DeviceIndex(32, metric="ip")
    idx.add(np.eye(32, dtype=np.float32))
    q = np.ones(32, np.float32)
    for kw in (dict(topn=0), dict(topn=2000), dict(topn=10, fuse_limit=5000), dict(topn=10, max_results=0),
               dict(topn=10, mmr_lambda=1.5)):
        args = dict(topn=10, fuse_limit=None, max_results=5, mmr_lambda=0.5)
        args.update(kw)
        with pytest.raises(RagHipError):
            hybrid_mmr(idx, q, args["topn"], np.zeros(0, np.int64), np.zeros(0), fuse_limit=args["fuse_limit"],
                       max_results=args["max_results"], mmr_lambda=args["mmr_lambda"])
    with pytest.raises(RagHipError):
        hybrid_mmr(idx, q, 10, np.zeros(1025, np.int64), np.zeros(1025))
    got = hybrid_mmr(idx, q, 10, np.zeros(0, np.int64), np.zeros(0), max_results=3)      # and a good call still works
    assert len(got["ids"]) == 3
    idx.close()


def _check_batch(idx, X, Q, topn, bm_rows, bm_scores, to_rows, to_id, *, max_results=10, lam=0.5, thr=0.8, metric="cosine", **kw):
    """hybrid_mmr_batch against the oracle's fusion + MMR per query, fed with the scores of the same search as a plain call.
    to_rows: ids of a search -> row positions (-1 kept); to_id: row position -> id."""
    import torch
    from rag_amd.index import hybrid_mmr_batch
    nq, nb = bm_rows.shape
    qt = torch.from_numpy(Q).cuda()
    tkw = {k: (torch.from_numpy(v).cuda() if k == "row_mask" else v) for k, v in kw.items()}
    got_ids, got_scores, got_n = hybrid_mmr_batch(idx, qt, topn, torch.from_numpy(bm_rows).cuda(), torch.from_numpy(bm_scores).cuda(),
                                                  max_results=max_results, mmr_lambda=lam, mmr_threshold=thr, mmr_metric=metric, **tkw)
    D, I = idx.search(qt, topn, **tkw)
    torch.cuda.synchronize()
    D, I = D.cpu().numpy(), I.cpu().numpy()
    rows = to_rows(I)
    gi, gs, gn = got_ids.cpu().numpy(), got_scores.cpu().numpy(), got_n.cpu().numpy()
    fl = min(1024, topn + nb)
    for i in range(nq):
        f_rows, f_scores, _, _, picked, status = _oracle_tail(X, D[i], rows[i], bm_rows[i], bm_scores[i], fl, 0.65, 0.35, lam, thr,
                                                               metric, max_results)
        want_ids = [to_id(f_rows[p]) for p in picked]
        assert gn[i] == len(picked), (idx.kind, nq, topn, i, int(gn[i]), len(picked))
        assert gi[i, :gn[i]].tolist() == want_ids, (idx.kind, nq, topn, i)
        assert np.array_equal(gs[i, :gn[i]], np.array([f_scores[p] for p in picked])), (idx.kind, nq, topn, i)
        assert np.all(gi[i, gn[i]:] == -1)
    if nq <= 12:           # host arrays in, host arrays out: the same answer
        from rag_amd.index import hybrid_mmr_batch as hb
        hi, hs, hn = hb(idx, Q, topn, bm_rows, bm_scores, max_results=max_results, mmr_lambda=lam, mmr_threshold=thr, mmr_metric=metric, **kw)
        assert np.array_equal(hi, gi) and np.array_equal(hn, gn) and all(np.array_equal(hs[i, :gn[i]], gs[i, :gn[i]]) for i in range(nq))


def test_batched_pipeline_equals_the_oracle_per_query():
    """rag_index_hybrid_mmr_batch_device: one search for every query's top-N, then fusion and MMR per query.  Each query's
    result is checked exactly against the oracle's fusion + MMR fed with the scores of that search (the same search as a plain
    call: at k = 200 both 12 and 3 queries take the segment walk), on a flat index with explicit ids and a mask, one
    with an id base, and an IVF index; keyword lists are padded with -1.  600 queries span two chunks of the tail launches
    (512 queries per launch)."""
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(21)
    n, d = 70_000, 128
    X, cent = _clustered(rng, n, d, ncl=30)
    ids = rng.permutation(5 * n)[:n].astype(np.int64)
    back = {int(v): r for r, v in enumerate(ids)}
    flat_ids = DeviceIndex(d, metric="ip")
    flat_ids.add(X, ids=ids)
    flat_base = DeviceIndex(d, metric="ip")
    flat_base.set_id_base(1000)
    flat_base.add(X)
    nlist = 32
    C0 = X[rng.choice(n, nlist, replace=False)].copy()
    iv = DeviceIndex(d, metric="ip", kind="ivf")
    iv.train_ivf(nlist, centroids=C0)
    iv.add(X)
    mask = (rng.uniform(size=n) < 0.6).astype(np.uint8)
    by_ids = (np.vectorize(lambda v: back[int(v)] if v >= 0 else -1), lambda r: int(ids[r]))
    by_base = (lambda I: np.where(I >= 0, I - 1000, -1), lambda r: r + 1000)
    plain = (lambda I: I, lambda r: r)
    for idx, nq, topn, nb, kw, (to_rows, to_id) in ((flat_ids, 12, 200, 60, {"row_mask": mask}, by_ids), (flat_base, 3, 200, 100, {}, by_base),
                                                    (flat_base, 9, 64, 0, {"min_score": 0.3}, by_base), (iv, 10, 150, 40, {"nprobe": 8}, plain),
                                                    (flat_base, 600, 70, 20, {}, by_base)):
        Q = np.stack([cent[i % cent.shape[0]] / np.linalg.norm(cent[i % cent.shape[0]]) for i in range(nq)]).astype(np.float32)
        Q += 0.02 * rng.standard_normal(Q.shape).astype(np.float32)
        bm_rows = rng.integers(0, n, (nq, nb)).astype(np.int64)
        if nb:
            bm_rows[:, -5:] = -1                                      # shorter lists, padded
            bm_rows[0, :] = -1                                        # a query without keyword hits
        bm_scores = rng.gamma(2.0, 2.0, size=(nq, nb))
        _check_batch(idx, X, Q, topn, bm_rows, bm_scores, to_rows, to_id, **kw)
    for idx in (flat_ids, flat_base, iv):
        idx.close()


def test_batched_pipeline_on_an_ivf_index_that_repacks_inside_the_call():
    """add -> search (the rows are packed by list) -> add more -> the batch call as the first search after the second add:
    that search re-packs the rows (a new row buffer, a new insertion -> storage map) while the call is being built, and the
    fusion / MMR stages must read the vectors through the buffers of AFTER the re-pack.  Both adds are shuffled so the
    storage order differs from the insertion order, and the candidates come from both."""
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(33)
    n, d, nlist = 40_000, 128, 24
    X, cent = _clustered(rng, n, d, ncl=30)
    X = X[rng.permutation(n)]
    C0 = X[rng.choice(n, nlist, replace=False)].copy()
    iv = DeviceIndex(d, metric="ip", kind="ivf")
    iv.train_ivf(nlist, centroids=C0)
    first = 25_000
    iv.add(X[:first])
    q0 = (cent[0] / np.linalg.norm(cent[0])).astype(np.float32)
    iv.search(q0[None, :], 10, nprobe=4)                             # packs the first 25 000 rows
    iv.add(X[first:])                                                # 15 000 rows wait at the tail
    nq, topn, nb = 7, 120, 30
    Q = np.stack([cent[i] / np.linalg.norm(cent[i]) for i in range(nq)]).astype(np.float32)
    Q += 0.02 * rng.standard_normal(Q.shape).astype(np.float32)
    bm_rows = rng.integers(0, n, (nq, nb)).astype(np.int64)
    bm_scores = rng.gamma(2.0, 2.0, size=(nq, nb))
    _check_batch(iv, X, Q, topn, bm_rows, bm_scores, lambda I: I, lambda r: r, nprobe=8)
    iv.add(X[:500] * np.float32(0.5))                                # and once more, with a larger pack behind it
    X2 = np.concatenate([X, X[:500] * np.float32(0.5)])
    bm_rows2 = rng.integers(0, n + 500, (nq, nb)).astype(np.int64)
    _check_batch(iv, X2, Q, topn, bm_rows2, bm_scores, lambda I: I, lambda r: r, nprobe=8)
    iv.close()
