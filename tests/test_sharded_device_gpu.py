"""The device path of the sharded search (send slot -> all-gather -> strided merge) for a
world of two and three shards, all of them on the one GPU of the test box: the collective is
replaced by a stand-in that places each shard's send slot where RCCL's all-gather would put
it, everything else (slot layout, id bases, the device merge) is the code the ranks run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Collective:
    """`all_gather_into_tensor` of a process group whose other ranks' send buffers are known."""

    def __init__(self, rank, world, others):
        self.rank, self.world, self.others = rank, world, others
        self.sent = None

    def is_initialized(self):
        return True

    def get_world_size(self, group=None):
        return self.world

    def get_rank(self, group=None):
        return self.rank

    def all_gather_into_tensor(self, out, mine, group=None):
        self.sent = mine.clone()
        slots = out.view(self.world, -1)          # the gather buffer is the concatenation of the slots
        for r in range(self.world):
            slots[r].copy_(mine if r == self.rank else self.others.get(r, mine * 0))

    def get_backend(self, group=None):
        return "nccl"


@pytest.mark.parametrize("world,nq,k", [(2, 1, 10), (2, 37, 10), (3, 4, 64), (2, 1100, 5)])
def test_device_exchange_equals_one_index_over_all_rows(world, nq, k):
    import torch
    from oracle import flat
    from rag_amd import DeviceIndex
    from rag_amd.sharded import ShardedIndex, shard_range
    rng = np.random.default_rng(world * 100 + nq)
    n, d = 30_011, 128
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X[n - 1] = X[3]                               # a tie between the first and the last shard
    Q = (X[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    Q[0] = X[3]
    qt = torch.from_numpy(Q).cuda()
    shards = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        idx = DeviceIndex(d, metric="ip")
        idx.set_id_base(lo)
        idx.add(X[lo:hi])
        shards.append(idx)
    # first pass: every rank's send slot (what it would contribute to the all-gather)
    slots = {}
    for r in range(world):
        sh = ShardedIndex(shards[r])
        sh.dist, sh.world, sh.rank = _Collective(r, world, {}), world, r
        sh.search(qt, k)
        slots[r] = sh.dist.sent
    # second pass: each rank with the others' slots in place ends with the full answer
    Dr, Ir = flat.flat_ip_search(X, Q, k)
    for r in range(world):
        sh = ShardedIndex(shards[r])
        sh.dist, sh.world, sh.rank = _Collective(r, world, {o: s for o, s in slots.items() if o != r}), world, r
        D, I = sh.search(qt, k)
        torch.cuda.synchronize()
        bad = flat.compare_topk(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, rtol=1e-5, atol=1e-6)
        assert bad == [], (r, bad[:3])
        assert I[0, 0].item() == 3 and I[0, 1].item() == n - 1      # the tie goes to the lower global id
    for idx in shards:
        idx.close()


@pytest.mark.parametrize("world,nq,k,nprobe", [(2, 1, 10, 8), (3, 33, 10, 16), (2, 600, 5, 8)])
def test_device_exchange_ivf_shards_equal_one_ivf_index(world, nq, k, nprobe):
    """IVF over row shards: every rank holds the same centroids and its own rows of every list,
    probes the same lists, and the exchange merges the slices -- the answer of one IVF index over
    all rows (and of the CPU oracle with those centroids)."""
    import torch
    from oracle import flat, ivf
    from rag_amd import DeviceIndex
    from rag_amd.sharded import ShardedIndex, shard_range
    rng = np.random.default_rng(world * 1000 + nq)
    n, d, nlist = 40_003, 128, 64
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    C = X[rng.choice(n, nlist, replace=False)].copy()
    Q = (X[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    qt = torch.from_numpy(Q).cuda()
    shards = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        idx = DeviceIndex(d, metric="ip", kind="ivf")
        idx.train_ivf(nlist, centroids=C)
        idx.set_id_base(lo)
        idx.add(X[lo:hi])
        shards.append(idx)
    slots = {}
    for r in range(world):
        sh = ShardedIndex(shards[r])
        sh.dist, sh.world, sh.rank = _Collective(r, world, {}), world, r
        sh.search(qt, k, nprobe=nprobe)
        slots[r] = sh.dist.sent
    Dr, Ir = ivf.ivf_search(X, C, Q, k, nprobe)
    for r in range(world):
        sh = ShardedIndex(shards[r])
        sh.dist, sh.world, sh.rank = _Collective(r, world, {o: s for o, s in slots.items() if o != r}), world, r
        D, I = sh.search(qt, k, nprobe=nprobe)
        torch.cuda.synchronize()
        bad = flat.compare_topk(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, rtol=1e-5, atol=1e-6)
        assert bad == [], (r, bad[:3])
    for idx in shards:
        idx.close()
