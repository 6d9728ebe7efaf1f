"""The multi-GPU exchange logic (row shards -> all-gather of (score, id) -> merge)
with world_size 2 on CPU over gloo.  The local search and the merge are the CPU
oracle here (injected): what is under test is the sharding, the id bases, the
packing and the collective."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import flat
    from rag_amd.sharded import ShardedIndex, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(99)
    n, d, k = 5003, 64, 10
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X[4000] = X[10]                      # an exact tie across the two shards
    Q = X[[10, 77, 3000]] + 0.01
    lo, hi = shard_range(n, rank, world)

    def local_search(q, kk, **kw):
        D, I = flat.flat_ip_search(X[lo:hi], q, kk)
        return D, np.where(I >= 0, I + lo, -1)

    def merge(D_all, I_all):
        W, nq, kk = D_all.shape
        Dc = D_all.permute(1, 0, 2).reshape(nq, W * kk).numpy()
        Ic = I_all.permute(1, 0, 2).reshape(nq, W * kk).numpy()
        order = np.lexsort((Ic, -Dc), axis=1)[:, :kk]      # score descending, lower id first
        return (torch.from_numpy(np.take_along_axis(Dc, order, 1)), torch.from_numpy(np.take_along_axis(Ic, order, 1)))

    sh = ShardedIndex(None, local_search=local_search, merge=merge)
    D, I = sh.search(Q, k)
    Dr, Ir = flat.flat_ip_search(X, Q, k)
    ok = np.array_equal(I, Ir) and np.array_equal(D, Dr)

    # the candidate fetch of the sharded fusion + MMR: every rank ends with the vectors of rows it does not hold
    class LocalRows:
        dim, device = d, 0

        def __len__(self):
            return hi - lo

        def reconstruct(self, rows):
            assert ((rows >= 0) & (rows < hi - lo)).all()          # only rows of this shard are asked for
            return X[lo:hi][rows]

    sh2 = ShardedIndex(LocalRows(), row_base=lo, local_search=local_search, merge=merge)
    want_rows = np.array([4000, 10, 0, n - 1, 2501, 2502, 10])      # the same request on every rank
    ok = ok and np.array_equal(sh2.gather_rows(want_rows), X[want_rows]) and sh2.gather_rows([]).shape == (0, d)
    np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([ok, sh.world == world]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(tmp_path / f"ok_{r}.npy").all()
