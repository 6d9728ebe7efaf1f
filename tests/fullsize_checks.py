"""Size-independent checks of an exact top-k answer, shared by the full-size GPU tests
(10M x 768 and the 12.5M x 768 per-GPU share of the 100M-row configuration): everything
runs on the device, so a 4096-query batch over tens of GB is checked in seconds."""
import numpy as np


def rescore_f64(torch, X, Q, I):
    """fp64 inner products of the returned rows, on the device: [nq, k]."""
    nq, k = I.shape
    out = torch.empty((nq, k), dtype=torch.float64, device=X.device)
    step = 512
    for a in range(0, nq, step):
        rows = X[I[a:a + step].reshape(-1)].double().reshape(-1, k, X.shape[1])
        out[a:a + step] = torch.einsum("qkd,qd->qk", rows, Q[a:a + step].double())
    return out


def check_batch_answer(torch, X, Q, D, I, planted=None):
    """Sorted lists, unique in-range ids, scores equal to an fp64 re-score of the returned
    rows within 1e-5 relative (+1e-6), planted copies of the queries first with score 1."""
    n = X.shape[0]
    nq, k = I.shape
    assert D.shape == (nq, k)
    assert bool((I >= 0).all()) and bool((I < n).all())
    assert bool((D[:, 1:] <= D[:, :-1]).all()), "scores not descending"
    srt = torch.sort(I, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all()), "duplicate ids in a list"
    ref = rescore_f64(torch, X, Q, I)
    err = (D.double() - ref).abs()
    tol = 1e-5 * ref.abs() + 1e-6
    assert bool((err <= tol).all()), float((err - tol).max())
    if planted is not None:
        p = torch.as_tensor(planted, device=I.device)
        assert torch.equal(I[:, 0], p), "a planted row is not first"
        assert bool((D[:, 0] - 1.0).abs().max() < 1e-5)


def block_edges(nq, block=256):
    """First and last query of every `block`-query block, plus a few in between."""
    qs = set()
    for b in range(0, nq, block):
        qs.add(b)
        qs.add(min(nq, b + block) - 1)
    rng = np.random.default_rng(3)
    qs.update(int(x) for x in rng.integers(0, nq, 8))
    return sorted(qs)


def check_against_single_scans(torch, idx, Q, D, I, queries, k):
    """The batch answer against single-query scans (another kernel) of the same index: the
    same ids, except where scores tie within the tolerance; scores within 1e-5 relative."""
    from oracle import flat
    Dn, In = D.cpu().numpy(), I.cpu().numpy()
    for qi in queries:
        Ds, Is = idx.search(Q[qi:qi + 1], k)
        assert idx.last_path == "scan"
        bad = flat.compare_topk(Dn[qi:qi + 1], In[qi:qi + 1], Ds.cpu().numpy(), Is.cpu().numpy(), rtol=1e-5, atol=1e-6)
        assert bad == [], (qi, bad[:3])


def check_no_sampled_row_beats_the_kth(torch, X, Q, D, I, queries, sample_rows=200_000, seed=9):
    """No row of a random sample scores above a query's k-th kept score unless it was returned."""
    g = torch.Generator(device=X.device)
    g.manual_seed(seed)
    sample = torch.randint(0, X.shape[0], (sample_rows,), generator=g, device=X.device)
    qsel = torch.as_tensor(queries, device=X.device)
    S = X[sample] @ Q[qsel].T
    for j, qi in enumerate(queries):
        kth = float(D[qi, -1])
        better = sample[S[:, j] > kth + 1e-6]
        assert set(better.cpu().tolist()) <= set(I[qi].cpu().tolist()), qi
