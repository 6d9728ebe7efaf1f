"""The way `bench.py --gpus N` starts its ranks, rehearsed without a GPU: N fresh processes over
gloo with a torch CPU scan standing in for the HIP search (`--stub-cpu`).  Checks the self-spawn,
the rank count the line reports, the loud failure on a world-size mismatch, and the slot layout
the device exchange hands to rag_merge_topk_device."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    env.update(kw)
    return env


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "1", "--stub-cpu"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["world_size"] == 2
    assert out["steps"] == 4 and out["warmup"] == 1
    assert out["data"] == "stub-cpu" and out["exchange_ok"] is True
    assert out["config"]["rows_per_gpu"] * 2 >= out["config"]["rows"]
    assert out["value"] > 0 and abs(out["ms_per_step"] * out["value"] - 1e3) < 1e-6 * 1e3


def test_gpus_8_without_a_launcher_spawns_eight_ranks():
    """The rank count of the 8-GPU configurations: eight shards, one all-gather per step, every rank ends with the
    whole corpus's answer."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "3", "--warmup", "1", "--stub-cpu"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 8 and out["world_size"] == 8 and out["exchange_ok"] is True
    assert out["config"]["rows_per_gpu"] * 8 >= out["config"]["rows"]


def test_under_a_launcher_the_same_line_comes_out():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3",
                        "--warmup", "0", "--stub-cpu"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["exchange_ok"] is True


def test_world_size_mismatch_is_an_error_not_a_warning():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-cpu"], env=_env(WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2
    assert "WORLD_SIZE=1" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_single_rank_needs_no_process_group():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "0", "--stub-cpu"],
                       env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["world_size"] == 1


def _slot_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import flat
    from rag_amd.sharded import shard_range, slot_layout
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    n, d = 3001, 32
    X = rng.standard_normal((n, d)).astype(np.float32)
    X[2500] = X[3]                                   # a tie across the shards
    ok = True
    for nq, k in ((1, 10), (3, 5), (5, 7), (4, 64)):           # odd nq*k: the score half ends on a half word
        Q = X[[3, 100, 2000, 2999, 17][:nq]] * 1.01
        lo, hi = shard_range(n, rank, world)
        D, I = flat.flat_ip_search(X[lo:hi], Q, k)
        I = np.where(I >= 0, I + lo, -1)
        words, d_off, d_stride, i_stride = slot_layout(nq, k)
        mine = torch.zeros(words, dtype=torch.int64)
        raw = mine.numpy().view(np.uint8)
        raw[:nq * k * 8] = I.astype(np.int64).reshape(-1).view(np.uint8)          # what search_into writes at i_ptr
        raw[d_off:d_off + nq * k * 4] = D.reshape(-1).view(np.uint8)              # ... and at i_ptr + d_off
        buf = torch.zeros(world * words, dtype=torch.int64)
        dist.all_gather_into_tensor(buf, mine)
        # read the gathered buffer exactly as rag_merge_topk_device is told to: list g at
        # D + g*d_stride floats / I + g*i_stride int64s, D starting d_off bytes into the buffer
        flat_bytes = buf.numpy().reshape(-1).view(np.uint8)
        Df = flat_bytes[d_off:].view(np.float32)
        Iw = buf.numpy().reshape(-1)
        D_all = np.stack([Df[g * d_stride: g * d_stride + nq * k].reshape(nq, k) for g in range(world)])
        I_all = np.stack([Iw[g * i_stride: g * i_stride + nq * k].reshape(nq, k) for g in range(world)])
        Dc = D_all.transpose(1, 0, 2).reshape(nq, world * k)
        Ic = I_all.transpose(1, 0, 2).reshape(nq, world * k)
        Ic_sort = np.where(Ic < 0, np.iinfo(np.int64).max, Ic)
        order = np.lexsort((Ic_sort, -Dc.astype(np.float64)), axis=1)[:, :k]
        Dm, Im = np.take_along_axis(Dc, order, 1), np.take_along_axis(Ic, order, 1)
        Dr, Ir = flat.flat_ip_search(X, Q, k)
        ok = ok and np.array_equal(Im, Ir) and np.array_equal(Dm, Dr)
    np.save(os.path.join(out_dir, f"slot_ok_{rank}.npy"), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_slot_layout_two_ranks_gloo(tmp_path):
    """The arguments the device exchange passes to the merge entry point (score offset, list
    strides) address every rank's ids and scores in a gathered buffer, checked by reading the
    buffer with that arithmetic on the CPU."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_slot_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(tmp_path / f"slot_ok_{r}.npy").all()
