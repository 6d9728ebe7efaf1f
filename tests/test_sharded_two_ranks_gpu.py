"""The multi-rank device path with MORE THAN ONE rank: two (and three) processes share the one
GPU of the test box, each owns a row shard, runs the HIP search into its send slot, the slots are
exchanged (gloo stands in for RCCL, which wants one GPU per rank) and every rank merges on the
device with rag_merge_topk_device.  Every rank must end with the oracle's answer for the whole
corpus.  Also runs bench.py itself in that arrangement."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL, ATOL = 1e-5, 1e-6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import flat, ivf
    from rag_amd import DeviceIndex
    from rag_amd.sharded import ShardedIndex, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    rng = np.random.default_rng(123)
    n, d = 150_001, 256
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X[140_000] = X[5]                                # an exact tie across shards
    Q = rng.standard_normal((70, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    Q[0] = X[5]
    lo, hi = shard_range(n, rank, world)
    problems = []

    idx = DeviceIndex(d, metric="ip", device=0)
    idx.set_id_base(lo)
    idx.adopt(torch.from_numpy(X[lo:hi]).cuda())
    sh = ShardedIndex(idx)
    assert sh.world == world
    qt = torch.from_numpy(Q).cuda()
    for nq, k in ((1, 10), (3, 10), (8, 100), (70, 10), (33, 64)):
        for _ in range(2):
            D, I = sh.search(qt[:nq], k)
            torch.cuda.synchronize()
        Dr, Ir = flat.flat_ip_search(X, Q[:nq], k)
        bad = flat.compare_topk(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, rtol=RTOL, atol=ATOL)
        if bad:
            problems.append(("flat", nq, k, idx.last_path, bad[:3]))
    # top-200 -> hybrid fusion -> MMR over the shards = the same call on one index over all rows
    from rag_amd.index import hybrid_mmr
    whole = DeviceIndex(d, metric="ip", device=0)
    whole.add(X)
    sh.row_base = lo
    q = (X[5] + 0.3 * X[77]).astype(np.float32)
    bm_rows = np.concatenate([rng.integers(0, n, 60), [5, 140_000, 77, -3, n + 9]])
    bm_scores = rng.gamma(2.0, 2.0, size=bm_rows.size)
    for kw in ({"mmr_lambda": 0.5, "mmr_threshold": 0.3}, {"mmr_lambda": 0.7, "mmr_threshold": 0.0, "mmr_metric": "dot_product",
                                                            "fuse_limit": 80, "max_results": 25}):
        a = sh.hybrid_mmr(q, 200, bm_rows, bm_scores, want_fused=True, **kw)
        b = hybrid_mmr(whole, q, 200, bm_rows, bm_scores, want_fused=True, **kw)
        # ids and orderings equal; scores bit for bit where a shard is scored by the kernel the whole index uses (shards
        # above 32768 rows, like the whole index: the scan), else within the tolerance of the exhaustive search -- a
        # shard's row count selects its kernel, i.e. its summation order
        same_kernel = (hi - lo) > 32768
        for name in ("ids", "picked", "fused_rows", "scores", "fused_scores", "fused_sem", "fused_bm"):
            x, y = np.asarray(a[name]), np.asarray(b[name])
            exact = same_kernel or name in ("ids", "picked", "fused_rows", "fused_bm")
            if x.shape != y.shape or not (np.array_equal(x, y) if exact else np.allclose(x, y, rtol=1e-5, atol=1e-6)):
                problems.append(("hybrid_mmr", name, repr(kw)))
        if a["status"] != b["status"] or a["n_fused"] != b["n_fused"] or len(a["ids"]) == 0:
            problems.append(("hybrid_mmr", "status / counts", repr(kw)))
    whole.close()
    idx.close()

    # IVF shards with shared centroids = one IVF index over all rows
    nlist, nprobe = 64, 8
    C = X[rng.choice(n, nlist, replace=False)].copy()
    iv = DeviceIndex(d, metric="ip", kind="ivf", device=0)
    iv.train_ivf(nlist, centroids=C)
    iv.set_id_base(lo)
    iv.add(X[lo:hi])
    shi = ShardedIndex(iv)
    for nq, k in ((1, 10), (70, 10)):
        D, I = shi.search(qt[:nq], k, nprobe=nprobe)
        torch.cuda.synchronize()
        Dr, Ir = ivf.ivf_search(X, C, Q[:nq], k, nprobe)
        bad = flat.compare_topk(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, rtol=RTOL, atol=ATOL)
        if bad:
            problems.append(("ivf", nq, k, iv.last_path, bad[:3]))
    iv.close()
    with open(os.path.join(out_dir, f"rank_{rank}.json"), "w") as f:
        json.dump({"problems": repr(problems), "ok": not problems}, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_device_path_with_several_ranks_on_one_gpu(tmp_path, world):
    """world 8 = the rank count of the 8-GPU configurations: slot layout, shard ranges and the merge of eight lists."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        with open(tmp_path / f"rank_{r}.json") as f:
            res = json.load(f)
        assert res["ok"], res["problems"]


def test_bench_with_two_ranks_on_one_gpu():
    """bench.py started the way the driver starts it (`python bench.py --gpus 2`, no launcher), on
    a small corpus, ranks sharing the GPU: the line reports two ranks and the sharded path."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--rows", "400000", "--big-batch", "1024", "--big-steps", "2", "--no-cpu", "--no-side",
                        "--dist-backend", "gloo", "--share-gpu"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2
    assert out["config"]["rows_per_gpu"] == 200000 and out["config"]["dist_backend"] == "gloo"
    assert out["scaling"] == "strong" and out["value"] > 0
    assert out["roofline"]["launches"] == 5
    assert "batch_1024" in out
    assert out["verified"] and all(out["verified"].values()), out["verified"]


def test_bench_with_eight_ranks_on_one_gpu():
    """The launch the driver uses for the 8-GPU scaling point (`python bench.py --gpus 8`), rehearsed with the eight
    ranks sharing the one GPU over gloo: eight row shards of 100 000 rows, one all-gather per step, the device merge
    of eight lists on every rank, and every rank's answers verified."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2",
                        "--rows", "800000", "--big-batch", "512", "--big-steps", "2", "--no-cpu", "--no-side", "--no-prefilter",
                        "--dist-backend", "gloo", "--share-gpu"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["world_size"] == 8
    assert out["config"]["rows_per_gpu"] == 100000 and out["config"]["dist_backend"] == "gloo"
    assert out["verified"] and all(out["verified"].values()), out["verified"]
