"""One GPU's share of the configuration 'exact flat 100M x 768 sharded across 8 GPUs': 12.5M x 768
rows with an id base (the rank's first global row), batch 1 and batch 4096, through the
size-independent properties of fullsize_checks.py -- the CPU oracle cannot run at this size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROWS_TOTAL, WORLD, RANK, D, K = 100_000_000, 8, 5, 768, 10
BIG = 4096


@pytest.fixture(scope="module")
def share():
    import torch
    from bench import make_corpus, make_queries
    from rag_amd import DeviceIndex
    from rag_amd.sharded import shard_range
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 48 << 30:
        pytest.skip("needs 48 GB of free HBM")
    lo, hi = shard_range(ROWS_TOTAL, RANK, WORLD)
    assert hi - lo == 12_500_000
    X = make_corpus(torch, lo, hi, D, dev)
    Q = make_queries(torch, BIG, D, dev)
    rng = np.random.default_rng(11)
    planted = rng.choice(hi - lo, BIG, replace=False)
    planted[0], planted[1] = 0, hi - lo - 1
    X[torch.from_numpy(planted).to(dev)] = Q
    idx = DeviceIndex(D, metric="ip")
    idx.set_id_base(lo)
    idx.adopt(X)
    yield torch, X, Q, idx, planted, lo
    idx.close()
    del X
    torch.cuda.empty_cache()


def test_share_batch_1(share):
    torch, X, Q, idx, planted, lo = share
    import fullsize_checks as fc
    for qi in (0, 1, 2, 700, 4095):
        Dq, Iq = idx.search(Q[qi:qi + 1], K)
        assert idx.last_path == "scan"
        assert int(Iq[0, 0]) == lo + planted[qi]
        fc.check_batch_answer(torch, X, Q[qi:qi + 1], Dq, Iq - lo, planted[qi:qi + 1])
        fc.check_no_sampled_row_beats_the_kth(torch, X, Q[qi:qi + 1], Dq, Iq - lo, [0])


def test_share_batch_4096(share):
    torch, X, Q, idx, planted, lo = share
    import fullsize_checks as fc
    Dg, Ig = idx.search(Q, K)
    assert idx.last_path == "gemm"
    fc.check_batch_answer(torch, X, Q, Dg, Ig - lo, planted)
    fc.check_against_single_scans(torch, idx, Q, Dg, Ig, fc.block_edges(BIG), K)
    fc.check_no_sampled_row_beats_the_kth(torch, X, Q, Dg, Ig - lo, [0, 255, 256, 3000, 4095])


def test_share_small_batches(share):
    """16 and 32 queries (the register-streaming kernel) and 200 queries on the same rows."""
    torch, X, Q, idx, planted, lo = share
    import fullsize_checks as fc
    for nq, path in ((16, "narrow"), (32, "narrow"), (200, "gemm")):
        Dn, In = idx.search(Q[:nq], K)
        assert idx.last_path == path
        fc.check_batch_answer(torch, X, Q[:nq], Dn, In - lo, planted[:nq])
        fc.check_against_single_scans(torch, idx, Q, Dn, In, [0, nq // 2, nq - 1], K)
