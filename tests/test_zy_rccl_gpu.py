"""The multi-GPU exchange step on a real RCCL process group (one rank).  Collected last: it
initialises torch.distributed inside the test process."""
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


def test_sharded_exchange_path_on_rccl_with_one_rank():
    """The multi-GPU step (local search into the send slot, RCCL all-gather, strided device
    merge) run for real on a one-rank process group: same answer as the plain search."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from rag_amd import DeviceIndex
    from rag_amd.sharded import ShardedIndex
    rng = np.random.default_rng(41)
    n, d, k = 200_000, 768, 10
    X = _unit(rng, n, d)
    Q = _unit(rng, 40, d)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        idx = DeviceIndex(d, metric="ip")
        idx.set_id_base(5_000_000)
        idx.adopt(torch.from_numpy(X).cuda())
        sh = ShardedIndex(idx)
        sh.exchange_when_alone = True
        qt = torch.from_numpy(Q).cuda()
        for nq in (1, 3, 40):
            for _ in range(2):                      # second round reuses the gather buffers
                D, I = sh.search(qt[:nq], k)
                torch.cuda.synchronize()
                _check(D.cpu().numpy(), I.cpu().numpy() - 5_000_000, X, Q[:nq], k)
        idx.close()
    finally:
        dist.destroy_process_group()
