"""Parity of the batched MFMA search (GEMM fused with top-k) with the CPU oracle.
Same tolerance as the scan path: ids equal except score ties within 1e-5
relative (+1e-6 absolute); scores within that tolerance."""
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


@pytest.mark.parametrize("n,d,nq,k", [
    (40_000, 768, 256, 10), (33_333, 384, 100, 10), (20_000, 1024, 129, 10), (70_001, 128, 1000, 10),
    (5_000, 64, 16, 1), (9_000, 256, 300, 64), (100, 32, 40, 10), (127, 512, 64, 10), (128, 96, 17, 5),
    (3, 768, 20, 10), (50_000, 768, 2048, 10),
    # the narrow tile shapes (<= 32 and <= 64 queries per block, 256-row tiles)
    (40_000, 768, 9, 10), (40_000, 768, 32, 10), (40_001, 768, 33, 10), (25_000, 1024, 64, 64),
    (300, 64, 12, 10), (1_000, 256, 48, 3), (255, 384, 31, 10), (257, 384, 50, 64),
    # the 256 x 256 shape (8 waves) from 1024 queries
    (20_000, 768, 1024, 10), (513, 1024, 1500, 64), (70_000, 384, 4096, 10),
    # one, two and three K steps in the shapes whose K loop hands steps over early
    (30_000, 32, 130, 10), (30_000, 64, 1100, 10), (30_000, 96, 200, 20), (2_000, 32, 1024, 64),
    # 1M+ rows: the 256 x 256 shape also below 1024 queries where it pads the batch like 128 x 128 (200, 450), not at 300
    (1_050_000, 64, 200, 10), (1_048_576, 32, 450, 16), (1_050_001, 32, 300, 10),
    # 5..32 queries outside the register-streaming kernel (rows of 96 floats, k above 32) on more than 32768 rows: seeded 256 x 32 shape
    (70_000, 96, 9, 10), (100_000, 64, 32, 64), (40_000, 768, 20, 64), (33_000, 32, 5, 10),
    # query-block counts that no XCD grouping divides: 9 and 17 blocks of 256 (plain placement, no padding blocks), 12 blocks (groups of 4)
    (40_000, 64, 2304, 10), (36_000, 32, 4352, 10), (34_000, 64, 3000, 64)])
def test_gemm_matches_oracle(n, d, nq, k):
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(n * 7 + d + nq)
    X = _unit(rng, n, d)
    Q = _unit(rng, nq, d)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    D, I = idx.search(Q, k)
    from paths import flat_fp32_path
    assert idx.last_path == flat_fp32_path(n, d, nq, k)       # up to 32 queries (k <= 32, 128-float rows): the narrow kernel
    _check(D, I, X, Q, k)
    idx.close()


@pytest.mark.parametrize("nq", [20, 64, 130, 1030])
def test_gemm_adversarial_order_duplicates_and_mask(nq):
    """Rows sorted by increasing score for one query make every row a candidate
    (the buffers must compact, never overflow); duplicate rows tie exactly.
    One case per tile shape."""
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(23)
    n, d = 30_000, 128
    X = _unit(rng, n, d)
    Q = _unit(rng, nq, d)
    order = np.argsort(X @ Q[0])                 # ascending for query 0
    X = np.ascontiguousarray(X[order])
    X[1000:1040] = X[29_990]                      # 40 copies of a strong row, lower ids
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    D, I = idx.search(Q, 10)
    from paths import flat_fp32_path
    assert idx.last_path == flat_fp32_path(n, d, nq, 10)
    _check(D, I, X, Q, 10)
    D, I = idx.search(Q, 64)
    assert idx.last_path == flat_fp32_path(n, d, nq, 64)
    _check(D, I, X, Q, 64)
    mask = (rng.uniform(size=n) < 0.2).astype(np.uint8)
    D, I = idx.search(Q, 10, row_mask=mask)
    _check(D, I, X, Q, 10, row_mask=mask)
    idx.set_active(np.arange(29_000, 30_000), False)
    m2 = np.ones(n, np.uint8)
    m2[29_000:] = 0
    D, I = idx.search(Q, 10, min_score=0.05)
    _check(D, I, X, Q, 10, row_mask=m2, min_score=0.05)
    idx.close()


def test_batch_and_single_query_kernels_agree_bitwise_on_ids():
    """The two kernels order candidates by the same rule; on data without near
    ties they return identical id lists."""
    import os
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(29)
    n, d = 60_000, 768
    X = _unit(rng, n, d)
    Q = _unit(rng, 32, d)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    Dg, Ig = idx.search(Q, 10)
    assert idx.last_path == "narrow"
    rows = [idx.search(Q[i], 10) for i in range(32)]
    assert idx.last_path == "scan"
    Is = np.concatenate([r[1] for r in rows])
    Ds = np.concatenate([r[0] for r in rows])
    assert np.array_equal(Ig, Is)
    assert np.allclose(Dg, Ds, rtol=RTOL, atol=ATOL)
    idx.close()


@pytest.mark.parametrize("n,d,nq,k", [(1024, 64, 128, 10), (32_768, 128, 300, 64), (16_384, 1024, 1100, 64), (9_001, 96, 129, 1),
                                       (20_003, 384, 2048, 10), (5_000, 32, 4096, 1024), (30_001, 768, 200, 100)])
def test_scores_path_matches_oracle(n, d, nq, k):
    """A large batch over a short corpus: the matrix-core kernel writes every score and one workgroup per
    query selects (path "gemm-scores").  Row counts that are not multiples of 4 take the scalar stores; masks,
    thresholds, explicit ids and exact ties go through the selection kernel's rules."""
    from oracle import flat
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(n + nq + k)
    X = _unit(rng, n, d)
    X[n // 2: n // 2 + 5] = X[3]                           # exact ties: lower row first
    Q = _unit(rng, nq, d)
    Q[0] = X[3]
    ids = rng.permutation(4 * n)[:n].astype(np.int64)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X[: n // 3], ids=ids[: n // 3])
    idx.add(X[n // 3:], ids=ids[n // 3:])
    D, I = idx.search(Q, k)
    assert idx.last_path == "gemm-scores"
    _check(D, I, X, Q, k, ids=ids)
    if k >= 6:
        assert I[0, :6].tolist() == [int(ids[3])] + [int(v) for v in ids[n // 2: n // 2 + 5]]
    mask = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    D, I = idx.search(Q, k, row_mask=mask, min_score=0.05)
    assert idx.last_path == "gemm-scores"
    _check(D, I, X, Q, k, ids=ids, row_mask=mask, min_score=0.05)
    idx.set_active(np.arange(0, n, 2), False)
    alive = np.ones(n, np.uint8)
    alive[::2] = 0
    D, I = idx.search(Q, k)
    _check(D, I, X, Q, k, ids=ids, row_mask=alive)
    D2, I2 = idx.search(Q, k)                              # the same answer twice
    assert np.array_equal(D, D2) and np.array_equal(I, I2)
    idx.close()


@pytest.mark.parametrize("n,d,nq,k", [(100_000, 128, 40, 100), (65_537, 64, 300, 200), (40_000, 96, 9, 1024), (200_000, 32, 1100, 65),
                                       (32_769, 256, 8, 100), (1_500, 64, 20, 300), (98_304, 384, 600, 128),
                                       # the selecting kernel's 256 x 256 shape up to k = 200 (2048+ queries), the segment walk otherwise
                                       (120_000, 64, 2048, 200), (60_000, 128, 2500, 100), (50_000, 96, 2100, 201), (90_000, 32, 1024, 150),
                                       # groups of segments per launch: 32 + 2 segments under 3 queries (the last one of 17 rows),
                                       # 20 + 2 under 50 queries (the 128 x 64 shape), 5 + 5 + 3 under 200, 2 and 5 queries
                                       (33 * 32_768 + 17, 32, 3, 100), (700_000, 32, 50, 70), (400_000, 32, 200, 90),
                                       (70_001, 64, 2, 65), (40_000, 96, 5, 1000),
                                       # rows of a multiple of 128 floats under up to 32 queries: the register-streaming kernel writes the scores
                                       (32 * 32_768 + 17, 128, 6, 70), (150_001, 768, 8, 200), (90_000, 384, 20, 100), (70_000, 256, 32, 65),
                                       (1_030, 128, 2, 100), (140_000, 1024, 17, 80),
                                       # 2^20+ rows under 65+ queries, k <= 200: the selecting kernel (128 x 128 and 256 x 256 shapes) with its
                                       # thresholds seeded from the first 8 segments; 64 queries and k = 201 walk the segments
                                       (1_048_576 + 4_099, 32, 256, 200), (1_050_000, 32, 450, 65), (1_060_000, 32, 1030, 100),
                                       (1_048_576, 64, 300, 200), (1_048_576, 32, 2050, 200), (1_050_000, 32, 65, 100),
                                       (1_060_000, 32, 128, 200), (1_050_000, 64, 600, 150), (1_048_576, 32, 64, 100), (1_048_576, 32, 300, 201)])
def test_segments_path_matches_oracle(n, d, nq, k):
    """k above 64 with 2 or more queries: the corpus is walked in segments of 32768 rows (score matrix on the matrix
    cores, one selection workgroup per query and segment, several segments per launch under small batches) and the
    per-segment lists are merged (path "gemm-segments").  Last segments of one row and of a few thousand, an exact tie
    across a segment boundary, masks, a threshold, explicit ids, fewer hits than k."""
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(n + nq + k)
    X = _unit(rng, n, d)
    if n > 40_000:
        X[32_768 + 5] = X[7]                              # equal scores in two segments: the lower row first
    Q = _unit(rng, nq, d)
    Q[0] = X[7]
    ids = rng.permutation(4 * n)[:n].astype(np.int64)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X[: n // 3], ids=ids[: n // 3])
    idx.add(X[n // 3:], ids=ids[n // 3:])
    from paths import flat_fp32_path
    want_path = flat_fp32_path(n, d, nq, k)          # 2048+ queries up to k = 200 stay in the selecting kernel's 256 x 256 shape
    D, I = idx.search(Q, k)
    assert idx.last_path == want_path
    _check(D, I, X, Q, k, ids=ids)
    if n > 40_000:
        assert I[0, 0] == ids[7] and I[0, 1] == ids[32_768 + 5] and D[0, 0] == D[0, 1]
    mask = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    D, I = idx.search(Q, k, row_mask=mask, min_score=0.1)        # most queries end with fewer than k hits
    assert idx.last_path == want_path
    _check(D, I, X, Q, k, ids=ids, row_mask=mask, min_score=0.1)
    idx.set_active(np.arange(0, n, 3), False)
    alive = np.ones(n, np.uint8)
    alive[::3] = 0
    D, I = idx.search(Q, k)
    _check(D, I, X, Q, k, ids=ids, row_mask=alive)
    D2, I2 = idx.search(Q, k)
    assert np.array_equal(D, D2) and np.array_equal(I, I2)
    idx.close()


@pytest.mark.parametrize("nq,k", [(256, 200), (1024, 100), (300, 200), (70, 128)])
def test_seeded_thresholds_on_adversarial_order(nq, k):
    """k above 64 on 2^20+ rows: the selecting kernel starts from thresholds taken from the first segments (1 to 8 of 32768
    rows, by batch size; path "gemm-seeded").  Rows sorted by increasing score for one query make that starting value useless
    for it (every later row beats it: the buffers must compact, never overflow); blocks of duplicates lie across the segment
    ends where the seeding rows can stop; a mask that removes the best seeding rows must not raise a threshold above what the
    allowed rows reach."""
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(nq + k)
    n, d = (1 << 20) + 5_000, 32
    X = _unit(rng, n, d)
    Q = _unit(rng, nq, d)
    order = np.argsort(X @ Q[0], kind="stable")
    X = np.ascontiguousarray(X[order])
    edge = 8 * 32_768
    X[edge - 3: edge + 4] = X[11]
    for e in (32_768, 2 * 32_768, 4 * 32_768):
        X[e - 2: e + 2] = X[12]
    Q[1] = X[11]
    Q[3] = X[12]
    Q[2] = X[100]
    X[200_000:200_300] = X[100] * np.float32(0.999)       # 300 near-copies of a seeding row ...
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    D, I = idx.search(Q, k)
    assert idx.last_path == "gemm-seeded"
    _check(D, I, X, Q, k)
    assert I[1, :8].tolist() == [11] + list(range(edge - 3, edge + 4))
    assert I[3, :13].tolist() == [12] + [e + j for e in (32_768, 2 * 32_768, 4 * 32_768) for j in (-2, -1, 0, 1)]
    mask = np.ones(n, np.uint8)
    mask[200_000:200_300] = 0                              # ... masked out: they may not count towards query 2's threshold
    mask[100] = 0
    mask[rng.uniform(size=n) < 0.5] = 0
    D, I = idx.search(Q, k, row_mask=mask, min_score=0.3)
    assert idx.last_path == "gemm-seeded"
    _check(D, I, X, Q, k, row_mask=mask, min_score=0.3)
    idx.close()


@pytest.mark.parametrize("n,nq,k", [(60_000, 17, 10), (60_000, 32, 32), (300_000, 24, 10), (40_000, 16, 10)])
def test_narrow_kernel_on_rows_of_1024_floats(n, nq, k):
    """32 queries x 1024 floats fill 128 KiB of LDS: the kernel then keeps one transposer image per wave for both row
    groups instead of two (17..32 queries; up to 16 queries keep two).  Near ties, duplicates, a mask, a threshold."""
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(n + nq)
    d = 1024
    X = _unit(rng, n, d)
    Q = _unit(rng, nq, d)
    X[n // 2: n // 2 + 6] = X[9]
    Q[1] = X[9]
    X[100:130] = X[77] * np.float32(0.9995)
    Q[2] = X[77]
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    D, I = idx.search(Q, k)
    assert idx.last_path == "narrow"
    _check(D, I, X, Q, k)
    if k >= 7:
        assert I[1, :7].tolist() == [9] + list(range(n // 2, n // 2 + 6))
    mask = (rng.uniform(size=n) < 0.4).astype(np.uint8)
    D, I = idx.search(Q, k, row_mask=mask, min_score=0.05)
    _check(D, I, X, Q, k, row_mask=mask, min_score=0.05)
    rows = [idx.search(Q[i], k) for i in (0, nq - 1)]
    assert np.array_equal(np.concatenate([r[1] for r in rows]), idx.search(Q, k)[1][[0, nq - 1]])
    idx.close()


@pytest.mark.parametrize("nq,k", [(7, 10), (16, 32), (32, 10), (20, 1)])
def test_narrow_kernel_with_sample_launch(nq, k):
    """5..32 queries on 2^20+ rows: the register-streaming kernel starts from thresholds its first launch took from the
    leading 32768 rows.  Rows sorted by increasing score for query 0 (its starting value is useless), duplicates inside
    the leading rows, a mask that removes the best of them, fewer allowed rows than k for the last case."""
    from rag_amd import DeviceIndex
    rng = np.random.default_rng(100 * nq + k)
    n, d = (1 << 20) + 3_000, 128
    X = _unit(rng, n, d)
    Q = _unit(rng, nq, d)
    order = np.argsort(X @ Q[0], kind="stable")
    X = np.ascontiguousarray(X[order])
    X[8_190:8_195] = X[3]
    Q[1] = X[3]
    Q[2] = X[50]
    X[4_000:4_040] = X[50] * np.float32(0.999)
    idx = DeviceIndex(d, metric="ip")
    idx.add(X)
    D, I = idx.search(Q, k)
    assert idx.last_path == "narrow"
    _check(D, I, X, Q, k)
    if k >= 6:
        assert I[1, :6].tolist() == [3] + list(range(8_190, 8_195))
    mask = np.ones(n, np.uint8)
    mask[4_000:4_040] = 0
    mask[50] = 0
    mask[rng.uniform(size=n) < 0.5] = 0
    D, I = idx.search(Q, k, row_mask=mask, min_score=0.2)
    _check(D, I, X, Q, k, row_mask=mask, min_score=0.2)
    few = np.zeros(n, np.uint8)
    few[[5, 9_000, 500_000]] = 1
    D, I = idx.search(Q, k, row_mask=few)
    _check(D, I, X, Q, k, row_mask=few)
    idx.close()
