"""The fused fan-out (segment search + weighted best-per-row fusion + ranking on the
device) against the oracle's restatement of vector_db/search_manager.py:877-932 fed with
the per-pair hit lists of the plain segment search.  Rows, scores and strategies must
match exactly: the fusion is fp64 products and comparisons only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _want(idx, Q, weights, segs, k, limit, min_score, mask):
    from oracle import fusion as ofusion
    D, I = idx.search_segments(Q, segs, k, min_score=min_score, row_mask=mask)
    per_doc = []
    for si in range(len(segs)):
        strategies = []
        for qi in range(Q.shape[0]):
            hits = [(str(int(r)), float(s)) for r, s in zip(I[qi, si], D[qi, si]) if r >= 0]
            strategies.append((qi, float(weights[qi]), hits))
        per_doc.append((si, strategies))
    return ofusion.multi_embedding_fuse(per_doc, limit)


def _check(idx, Q, weights, segs, k, limit, min_score=0.1, mask=None):
    want = _want(idx, Q, weights, segs, k, limit, min_score, mask)
    rows, scores, strat = idx.search_segments_fused(Q, weights, segs, k, limit, min_score=min_score, row_mask=mask)
    assert idx.last_path == "segments-fused"
    assert [int(r) for r in rows] == [int(w[0].split("_")[1]) for w in want]
    assert scores.tolist() == [w[1] for w in want]
    assert strat.tolist() == [w[2] for w in want]
    return len(rows)


def test_fused_fanout_matches_oracle():
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(77)
    for d, n, nseg, nq, k, limit in ((384, 6000, 12, 3, 20, 10), (128, 3000, 5, 1, 40, 20), (384, 20000, 150, 3, 20, 10),
                                     (64, 4000, 40, 3, 200, 100), (96, 2000, 3, 4, 256, 1024), (384, 9000, 700, 2, 8, 1000)):
        X = _unit(rng, n, d)
        idx = DeviceIndex(d, metric="cosine")
        idx.add(X)
        cuts = np.sort(rng.choice(np.arange(1, n), nseg - 1, replace=False))
        segs = [(int(a), int(b)) for a, b in zip(np.r_[0, cuts], np.r_[cuts, n])]
        Q = _unit(rng, nq, d)
        weights = [0.4, 0.6, 0.3, 0.5][:nq]
        got = _check(idx, Q, weights, segs, k, limit, min_score=0.02)
        assert got > 0
        mask = (rng.uniform(size=n) < 0.5).astype(np.uint8)
        _check(idx, Q, weights, segs, k, limit, min_score=0.0, mask=mask)
        idx.close()


def test_fused_fanout_ties_and_empty_results():
    """Equal weighted scores: the same query vector under two strategies (first wins),
    duplicated rows in different segments (segment order), a segment list with empty
    ranges, and a threshold nothing passes."""
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(5)
    d = 64
    base = _unit(rng, 50, d)
    X = np.concatenate([base, base, base[::-1], _unit(rng, 300, d)])      # repeated rows: equal scores across segments
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    segs = [(0, 50), (50, 50), (50, 100), (100, 150), (150, 450)]
    q = _unit(rng, 1, d)[0]
    Q = np.stack([q, q, 2 * q]).astype(np.float32)
    for weights in ([0.5, 0.5, 0.25], [0.6, 0.3, 0.3], [0.3, 0.6, 0.3]):
        _check(idx, Q, weights, segs, 30, 25, min_score=-10.0)
        _check(idx, Q, weights, segs, 60, 200, min_score=-10.0)
    rows, scores, strat = idx.search_segments_fused(Q, [0.4, 0.6, 0.3], segs, 10, 10, min_score=50.0)
    assert len(rows) == 0 and len(scores) == 0 and len(strat) == 0
    rows, _, _ = idx.search_segments_fused(Q, [0.4, 0.6, 0.3], [(7, 7)], 10, 10, min_score=0.0)
    assert len(rows) == 0
    with pytest.raises(Exception):
        idx.search_segments_fused(np.repeat(Q, 6, axis=0), [0.1] * 18, segs, 10, 10)       # more than 16 query vectors
    with pytest.raises(Exception):
        idx.search_segments_fused(Q, [0.4, 0.6, 0.3], segs, 512, 10)                       # nq * k above 1024
    idx.close()
