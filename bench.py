#!/usr/bin/env python3
"""Headline benchmark: queries/sec of exact inner-product top-10 over a
10M x 768 fp32 corpus of synthetic unit vectors, batch 1 (HBM-bound scan) and
batch 4096 (fp32-MFMA GEMM fused with top-k), on N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one batch of queries answered end to end with the corpus and the
queries already resident in HBM: query preparation, corpus scan, selection,
merge and, for N > 1, the all-gather of per-shard candidates plus the device
merge.  N > 1 shards the SAME corpus row-wise over the ranks (strong scaling:
total work per step is fixed).  Rank 0 prints one JSON line; `value` is the
batch-1 fp32 rate.  Beside it: `batch_4096` and `batch_32` (fp32 MFMA path), and --
clearly labelled, never `value` -- `batch_4096_bf16_first_pass` / `batch_1_bf16_first_pass`,
the opt-in mode that searches a bf16 copy first and re-scores the kept rows in fp32
(same answers).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, vendor peak
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--big-batch", type=int, default=4096)
    ap.add_argument("--big-steps", type=int, default=0, help="batch-4096 steps (0: min(steps, 5))")
    ap.add_argument("--no-big", action="store_true", help="skip the batch-4096 leg")
    ap.add_argument("--no-prefilter", action="store_true", help="skip the opt-in bf16-first-pass leg")
    ap.add_argument("--mid-batch", type=int, default=32, help="a serving-sized batch reported beside the two headline ones (0: skip)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-side", action="store_true", help="skip the legs on the other configurations (small corpus, pipeline, IVF)")
    ap.add_argument("--cpu-rows", type=int, default=0,
                    help="rows of the corpus the CPU baseline scans (0: the whole corpus when the host has the memory for it, "
                         "else the first 1M rows with the rate scaled)")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="rehearsal of the launch / exchange path without a GPU: gloo, a torch CPU scan stands in for the "
                         "HIP search (never a measurement; the JSON line says data = stub-cpu)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) is the product path; gloo only rehearses the N-rank path where RCCL cannot run")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal: ranks beyond the visible GPUs share them (with --dist-backend gloo)")
    return ap.parse_args(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes through
    torch.distributed.run (one per GPU) and hand back their exit status.  Runs before this
    process has imported torch or touched the GPU; the parent only waits."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1" if args.stub_cpu else env.get("OMP_NUM_THREADS", "16"))
    return subprocess.run(cmd, env=env).returncode


def measured_traffic(kernel, rows, dim, world):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled as gfx950
    needs).  Only valid for the shape it was taken on; otherwise None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        if world == 1 and t.get("rows") == rows and t.get("dim") == dim and kernel in t:
            return t[kernel]["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def make_corpus(torch, lo, hi, dim, device, chunk=500_000):
    """Rows [lo, hi) of the global corpus: N(0,1), L2-normalised, fp32.  Chunk c of
    the GLOBAL row space is drawn from seed 1234 + c, so the corpus is the same
    whatever the number of ranks."""
    x = torch.empty((hi - lo, dim), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    c0, c1 = lo // chunk, (hi - 1) // chunk if hi > lo else -1
    for c in range(c0, c1 + 1):
        g.manual_seed(1234 + c)
        blk = torch.randn((chunk, dim), generator=g, device=device, dtype=torch.float32)
        blk /= blk.norm(dim=1, keepdim=True)
        a, b = max(lo, c * chunk), min(hi, (c + 1) * chunk)
        x[a - lo:b - lo] = blk[a - c * chunk:b - c * chunk]
        del blk
    return x


def make_queries(torch, nq, dim, device):
    g = torch.Generator(device=device)
    g.manual_seed(4321)
    q = torch.randn((nq, dim), generator=g, device=device, dtype=torch.float32)
    q /= q.norm(dim=1, keepdim=True)
    return q


def same_topk(torch, Da, Ia, Db, Ib, rtol=1e-5, atol=1e-6):
    """Two answers for the same queries from different kernels: score lists agree position by position
    within rtol*|s| + atol, and an id held by one side only scores within that bound of the other side's
    last kept score (a tie at the cut).  The summation orders differ, so exact ties may swap places."""
    Da, Db = Da.double(), Db.double()
    tol = rtol * Db.abs() + atol
    if not bool(((Da - Db).abs() <= tol).all()):
        return False
    for q in range(Ia.shape[0]):
        a, b = Ia[q].tolist(), Ib[q].tolist()
        if len(set(a)) != len(a):
            return False
        sa, sb = dict(zip(a, Da[q].tolist())), dict(zip(b, Db[q].tolist()))
        cut = float(tol[q, -1])
        for i in set(a) - set(b):
            if abs(sa[i] - float(Db[q, -1])) > cut:
                return False
        for i in set(b) - set(a):
            if abs(sb[i] - float(Da[q, -1])) > cut:
                return False
    return True


def rescored_ok(torch, X, Q, D, I, lo, hi, rtol=1e-5, atol=1e-6):
    """Scores against an fp64 re-score of the returned rows this rank holds (ids in [lo, hi)); lists
    descending, ids in the corpus."""
    if not bool((D[:, 1:] <= D[:, :-1]).all()):
        return False
    own = (I >= lo) & (I < hi)
    if not bool(((I >= 0) | (D < -1e38)).all()):
        return False
    ok = True
    step = 512
    for a in range(0, I.shape[0], step):
        Ic = (I[a:a + step] - lo).clamp(0, hi - lo - 1)
        rows = X[Ic.reshape(-1)].double().reshape(Ic.shape + (X.shape[1],))
        ref = torch.einsum("qkd,qd->qk", rows, Q[a:a + step].double())
        err = (D[a:a + step].double() - ref).abs()
        ok = ok and bool(((err <= rtol * ref.abs() + atol) | ~own[a:a + step]).all())
    return ok


def sample_queries(B, block=256, extra=8):
    """First and last query of every 256-query block plus a few in between (at most ~40)."""
    qs = set()
    for b in range(0, B, block):
        qs.add(b)
        qs.add(min(B, b + block) - 1)
    step = max(1, B // extra)
    qs.update(range(step // 2, B, step))
    return sorted(qs)


def timed_steps(torch, dist, world, step, nwarm, nsteps, device):
    """nwarm untimed steps, then exactly nsteps timed ones between barrier + device
    synchronisation on both sides; the time returned is the maximum over the ranks."""
    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    for i in range(nwarm):
        step(i)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(nsteps):
        step(nwarm + i)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        on_host = dist.get_backend() == "gloo"
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if on_host else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def rendezvous(args):
    """(world, rank, local_rank) from the launcher's environment.  The world the launcher
    made must be the one --gpus asked for: a mismatch ends the run with status 2."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"error: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    return world, rank, local


def check_world(dist, world):
    """The process group must have the size the launcher announced (RCCL / gloo saw every rank)."""
    seen = dist.get_world_size() if world > 1 else 1
    if seen != world:
        print(f"error: process group has {seen} ranks, expected {world}", file=sys.stderr)
        sys.exit(2)
    return seen


def stub_main(args):
    """The N-rank launch, sharding, exchange and timing path with a torch CPU scan in place of
    the HIP search and gloo in place of RCCL.  For tests of the launch path on a machine without
    a GPU; the line it prints is marked data = stub-cpu and is never a measurement."""
    import torch
    import torch.distributed as dist
    from rag_amd.sharded import ShardedIndex, shard_range

    world, rank, local = rendezvous(args)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    seen = check_world(dist, world)
    device = torch.device("cpu")
    rows, dim, k = min(args.rows, 20_000), min(args.dim, 64), args.k
    lo, hi = shard_range(rows, rank, world)
    g = torch.Generator().manual_seed(1234)
    Xall = torch.randn((rows, dim), generator=g)
    Xall /= Xall.norm(dim=1, keepdim=True)
    X = Xall[lo:hi]
    g.manual_seed(4321)
    Q = torch.randn((64, dim), generator=g)
    Q /= Q.norm(dim=1, keepdim=True)

    def order_rule_topk(S, ids, kk):
        # score descending, ties to the lower id: the order rule of the library
        o = torch.argsort(ids, dim=1, stable=True)
        S, ids = torch.gather(S, 1, o), torch.gather(ids, 1, o)
        o = torch.argsort(S, dim=1, descending=True, stable=True)[:, :kk]
        return torch.gather(S, 1, o), torch.gather(ids, 1, o)

    def local_search(q, kk):
        S = q @ X.T
        ids = torch.arange(lo, hi, dtype=torch.int64)[None, :].expand(q.shape[0], -1)
        D, I = order_rule_topk(S, ids, min(kk, hi - lo))
        if D.shape[1] < kk:
            pad = kk - D.shape[1]
            D = torch.cat([D, torch.full((q.shape[0], pad), -3.4028234663852886e38)], 1)
            I = torch.cat([I, torch.full((q.shape[0], pad), -1, dtype=torch.int64)], 1)
        return D.contiguous(), I.contiguous()

    def merge(D_all, I_all):
        w, nq, kk = D_all.shape
        S = D_all.permute(1, 0, 2).reshape(nq, w * kk)
        ids = I_all.permute(1, 0, 2).reshape(nq, w * kk)
        ids_sort = torch.where(ids < 0, torch.full_like(ids, 2 ** 62), ids)
        D, I = order_rule_topk(S, ids_sort, kk)
        return D, torch.where(I == 2 ** 62, torch.full_like(I, -1), I)

    sharded = ShardedIndex(local_search=local_search, merge=merge)
    search = sharded.search if world > 1 else local_search

    def step(i):
        j = i % Q.shape[0]
        return search(Q[j:j + 1], k)

    dt = timed_steps(torch, dist, world, step, args.warmup, args.steps, device)
    # every rank must hold the answer of the whole corpus
    D, I = search(Q[:8], k)
    Dr, Ir = order_rule_topk(Q[:8] @ Xall.T, torch.arange(rows)[None, :].expand(8, -1), k)
    ok = bool(torch.equal(I, Ir) and torch.allclose(D, Dr, rtol=1e-5, atol=1e-6))
    flag = torch.tensor([1 if ok else 0])
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({
            "metric": f"queries/sec, top-{k} exact IP, {rows}x{dim} fp32, batch 1", "value": args.steps / dt,
            "unit": "queries/sec", "n_gpus": world, "world_size": seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "stub-cpu", "exchange_ok": bool(flag.item()),
            "config": {"workload": "launch-path rehearsal: torch CPU scan + gloo, not a measurement",
                       "rows": rows, "dim": dim, "k": k, "rows_per_gpu": hi - lo}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not flag.item():
        sys.exit(4)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(torch, X, Q, k, rows_total, big_batch):
    """The oracle's fp32 sgemm/sgemv path (a CPU FAISS-style flat scan) over `X` -- the whole corpus when the host
    has the memory for it, else its first rows with the rate scaled -- on the host cores of this box: about ten
    batch-1 queries and one 256-query batch."""
    import numpy as np
    from oracle import flat
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    t0 = time.perf_counter()
    Xh = np.empty(tuple(X.shape), dtype=np.float32)
    step = 1_000_000
    for r0 in range(0, X.shape[0], step):                   # chunked: no second full-size staging buffer on the host
        Xh[r0:r0 + step] = X[r0:r0 + step].cpu().numpy()
    t_copy = time.perf_counter() - t0
    Qh = Q.cpu().numpy()
    sample = Xh.shape[0]
    scale = sample / float(rows_total)
    whole = sample == rows_total
    flat.flat_ip_search(Xh, Qh[:1], k, precise=False)        # warm
    nq1 = 0
    budget, cap = (8.0, 10) if whole else (10.0, 1000)
    t0 = time.perf_counter()
    while nq1 < cap and time.perf_counter() - t0 < budget:
        flat.flat_ip_search(Xh, Qh[nq1:nq1 + 1], k, precise=False)
        nq1 += 1
    dt1 = time.perf_counter() - t0
    qps1 = nq1 / dt1 * scale
    nb = min(big_batch, 256 if whole else 2048)
    t0 = time.perf_counter()
    flat.flat_ip_search(Xh, Qh[:nb], k, precise=False)
    dtb = time.perf_counter() - t0
    qpsb = nb / dtb * scale
    faiss_leg = None
    try:                                  # FAISS itself, only where the box already has it
        import faiss
        fidx = faiss.IndexFlatIP(Xh.shape[1])
        fidx.add(Xh)
        fidx.search(Qh[:1], k)
        nf, t0 = 0, time.perf_counter()
        while nf < 1000 and time.perf_counter() - t0 < 5.0:
            fidx.search(Qh[nf:nf + 1], k)
            nf += 1
        faiss_leg = {"value": nf / (time.perf_counter() - t0) * scale, "unit": "queries/sec",
                     "sample": f"faiss.IndexFlatIP, {nf} batch-1 queries over the same {sample} rows, scaled by {scale:.3g}"}
    except Exception:
        pass
    over = f"the whole {sample} x {Xh.shape[1]} corpus" if whole else \
        f"the first {sample} rows x {Xh.shape[1]} of the same corpus, rate scaled by {scale:.3g} to {rows_total} rows"
    return {
        "value": qps1, "unit": "queries/sec", "cores": threads, "cpu_model": cpu_model(), "kind": "port", "faiss_cpu": faiss_leg,
        "rows_scanned": sample, "host_copy_s": t_copy,
        "sample": (f"oracle.flat.flat_ip_search(precise=False) = numpy/OpenBLAS fp32 sgemv + partition top-{k}, {threads} threads; "
                   f"{nq1} batch-1 queries over {over} in {dt1:.1f}s"),
        "batch_big": {"value": qpsb, "batch": nb,
                      "sample": f"one sgemm batch of {nb} queries over {over} in {dtb:.1f}s"},
    }


def leg_pipeline(torch, index, Q, args, device, topn=200, final=10, nq=16):
    """Configuration 'hybrid semantic + BM25 fusion + MMR (lambda 0.5) re-rank of top-200 to top-10' on
    the corpus of the headline legs: one library call per query (rag_index_hybrid_mmr_device), every
    buffer in HBM, nothing synchronised inside the timed region.  The keyword side is synthetic:
    100 rows, half of them among the query's dense top-200, scores U(0,1)."""
    from rag_amd.index import hybrid_mmr
    g = torch.Generator(device=device)
    g.manual_seed(99)
    n = len(index)
    bm = []
    for j in range(nq):
        _, I = index.search(Q[j:j + 1], topn)
        inside = I[0][torch.randperm(topn, generator=g, device=device)[:50]]
        outside = torch.randint(0, n, (50,), generator=g, device=device)
        bm.append((torch.cat([inside, outside]).contiguous(),
                   torch.rand(100, generator=g, device=device, dtype=torch.float64)))
    keep = []

    def step(i):
        j = i % nq
        keep.append(hybrid_mmr(index, Q[j], topn, bm[j][0], bm[j][1], max_results=final, mmr_lambda=0.5,
                               mmr_threshold=0.8, mmr_metric="cosine"))
        if len(keep) > 4:
            keep.pop(0)

    nsteps = max(1, min(args.steps, 32))
    index.profile(True)
    for i in range(3):
        step(i)
    torch.cuda.synchronize(device)
    index.profile_read(reset=True)
    t0 = time.perf_counter()
    for i in range(nsteps):
        step(i)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    kms, launches = index.profile_read(reset=True)
    index.profile(False)
    scan_ms = kms / max(launches, 1)
    gbs = n * args.dim * 4 / (scan_ms * 1e-3) / 1e9 if launches else 0.0
    total_ms = dt / nsteps * 1e3
    batch = None
    if not args.no_big:
        # the same configuration for a batch of queries: one search for every query's top-200 (segment walk on the matrix
        # cores), then fusion and MMR per query behind it on the stream (rag_index_hybrid_mmr_batch_device)
        from rag_amd.index import hybrid_mmr_batch
        B = min(1024, Q.shape[0])
        _, Ib = index.search(Q[:B], topn)
        pick = torch.argsort(torch.rand(B, topn, generator=g, device=device), dim=1)[:, :50]
        bm_rows = torch.cat([torch.gather(Ib, 1, pick), torch.randint(0, n, (B, 50), generator=g, device=device)], dim=1).contiguous()
        bm_scores = torch.rand(B, 100, generator=g, device=device, dtype=torch.float64)
        del Ib
        out = hybrid_mmr_batch(index, Q[:B], topn, bm_rows, bm_scores, max_results=final, mmr_lambda=0.5, mmr_threshold=0.8)
        torch.cuda.synchronize(device)
        bsteps = 2
        t0 = time.perf_counter()
        for _ in range(bsteps):
            out = hybrid_mmr_batch(index, Q[:B], topn, bm_rows, bm_scores, max_results=final, mmr_lambda=0.5, mmr_threshold=0.8)
        torch.cuda.synchronize(device)
        dtb = (time.perf_counter() - t0) / bsteps
        batch = {"value": B / dtb, "unit": "queries/sec", "batch": B, "steps": bsteps, "ms_per_step": dtb * 1e3,
                 "answered": int((out[2] > 0).sum().item()),
                 "note": "one search of the whole batch (k = 200 in the selecting kernel, thresholds seeded from the leading rows), "
                         "then fusion and MMR query by query on the same stream"}
    res = _pipeline_result(nsteps, dt, total_ms, scan_ms, topn, final, n, args, gbs, launches)
    if batch is not None:
        res["batch"] = batch
    return res


def _pipeline_result(nsteps, dt, total_ms, scan_ms, topn, final, n, args, gbs, launches):
    return {"value": nsteps / dt, "unit": "queries/sec", "steps": nsteps, "ms_per_step": total_ms,
            "scan_kernel_ms": scan_ms, "after_scan_ms": total_ms - scan_ms,
            "workload": f"top-{topn} exact scan -> 0.65/0.35 max-normalised fusion with 100 keyword hits -> MMR(0.5, 0.8, cosine) "
                        f"-> {final}, {n}x{args.dim}, one enqueue per query",
            "roofline": {"bound": "hbm", "kernel": "flat_scan_kernel (k = 200)", "achieved": gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "kernel_ms": scan_ms,
                         "launches": launches, "algorithmic_bytes_per_launch": n * args.dim * 4}}


def leg_keyword(torch, args, device, docs=20_000, candidates=100, limit=30):
    """The keyword side of the hybrid configuration: BM25 + min-max + ranking of 100 candidate chunks
    (the reference's candidate_limit, vector_db/search_manager.py:1129) on the device against the
    reference's per-query Python loop (restated in rag_amd.bm25.CorpusStatsManager, bit-identical).
    Synthetic corpus: 20k chunks of 40-120 words over a 5k-word vocabulary."""
    import random
    from rag_amd.bm25 import CorpusStatsManager, DeviceKeywordIndex
    rng = random.Random(17)
    vocab = ["w%04d" % i for i in range(5000)]
    texts = [" ".join(rng.choice(vocab) for _ in range(rng.randint(40, 120))) for _ in range(docs)]
    host = CorpusStatsManager()
    kw = DeviceKeywordIndex(device=device.index)
    t0 = time.perf_counter()
    kw.add_texts(texts)
    t_add = time.perf_counter() - t0
    stats = kw.corpus_stats()
    queries = [" ".join(rng.choice(vocab) for _ in range(6)) for _ in range(16)]
    cands = [sorted(rng.sample(range(docs), candidates)) for _ in range(16)]
    cands_dev = [torch.tensor(c, dtype=torch.int64, device=device) for c in cands]
    for i in range(4):
        out = kw.rank_device(queries[i], cands_dev[i], limit, None, stats)
    torch.cuda.synchronize(device)
    nsteps = 200
    t0 = time.perf_counter()
    for i in range(nsteps):
        out = kw.rank_device(queries[i % 16], cands_dev[i % 16], limit, None, stats)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    hsteps = 16
    t0 = time.perf_counter()
    for i in range(hsteps):
        terms = host.preprocess_query(queries[i])
        raw = [host.calculate_bm25_score(terms, texts[c], stats) for c in cands[i]]
        host.normalize_scores(raw, "minmax")
    dth = time.perf_counter() - t0
    same = kw.rank(queries[0], cands[0], limit, None, stats) == [
        (cands[0][i], nv, rv) for i, nv, rv in host.keyword_rank(queries[0], [texts[c] for c in cands[0]], texts, limit, None)]
    del out
    kw.close()
    return {"value": nsteps / dt, "unit": "queries/sec", "us_per_query": dt / nsteps * 1e6,
            "workload": f"BM25 + min-max + rank of {candidates} candidate chunks, corpus of {docs} chunks as token ids in HBM, "
                        f"top-{limit}; query terms and idf uploaded per call",
            "host_python_us_per_query": dth / hsteps * 1e6, "bit_identical_to_host": bool(same),
            "tokenise_and_upload_s": t_add}


def leg_small_corpus(torch, args, device, rows=100_000, dim=384, k=10):
    """Configuration 'IndexFlatIP 100k x 384 top-10' (the reference's CPU-runnable case) on the GPU:
    batch 1 latency and a 4096-query batch.  The corpus (154 MB) stays in the Infinity Cache, so the
    HBM peak is a conservative denominator."""
    from rag_amd import DeviceIndex
    X = make_corpus(torch, 0, rows, dim, device)
    Q = make_queries(torch, 4096, dim, device)
    idx = DeviceIndex(dim, metric="ip", device=device.index)
    idx.adopt(X)
    # results go to buffers the caller owns, as the C ABI has it.  Two loops: the timed one without the event pairs around the
    # scan kernel (two timed event records weigh several microseconds on a 40 us search), then the same searches with them
    # for the kernel time
    Dout = torch.empty((1, k), dtype=torch.float32, device=device)
    Iout = torch.empty((1, k), dtype=torch.int64, device=device)
    for i in range(10):
        idx.search_into(Q[i:i + 1], k, Dout.data_ptr(), Iout.data_ptr())
    torch.cuda.synchronize(device)
    nsteps = 200
    t0 = time.perf_counter()
    for i in range(nsteps):
        idx.search_into(Q[i:i + 1], k, Dout.data_ptr(), Iout.data_ptr())
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    idx.profile(True)
    idx.profile_read(reset=True)
    for i in range(nsteps):
        idx.search_into(Q[i:i + 1], k, Dout.data_ptr(), Iout.data_ptr())
    torch.cuda.synchronize(device)
    kms, launches = idx.profile_read(reset=True)
    scan_us = kms / max(launches, 1) * 1e3
    gbs = rows * dim * 4 / (scan_us * 1e-6) / 1e9 if launches else 0.0
    idx.search(Q, k)
    torch.cuda.synchronize(device)
    idx.profile_read(reset=True)
    t0 = time.perf_counter()
    for i in range(5):
        idx.search(Q, k)
    torch.cuda.synchronize(device)
    dtb = time.perf_counter() - t0
    kb, lb = idx.profile_read(reset=True)
    tf = 2.0 * 4096 * rows * dim / (kb / max(lb, 1) * 1e-3) / 1e12 if lb else 0.0
    idx.profile(False)
    Db, Ib = idx.search(Q, k)
    path_b = idx.last_path
    okb = rescored_ok(torch, X, Q, Db, Ib, 0, rows)
    ok1 = True
    for qi in sample_queries(4096):
        D1, I1 = idx.search(Q[qi:qi + 1], k)
        ok1 = ok1 and rescored_ok(torch, X, Q[qi:qi + 1], D1, I1, 0, rows)
        okb = okb and same_topk(torch, Db[qi:qi + 1], Ib[qi:qi + 1], D1, I1)
    path_1 = idx.last_path
    idx.close()
    return {"value": nsteps / dt, "unit": "queries/sec", "batch": 1, "us_per_query": dt / nsteps * 1e6,
            "verified": {"batch_1": bool(ok1), "batch_4096": bool(okb)}, "paths": {"batch_1": path_1, "batch_4096": path_b},
            "scan_kernel_us": scan_us, "workload": f"exact flat top-{k}, {rows}x{dim} fp32, batch 1",
            "roofline": {"bound": "hbm", "kernel": "flat_scan_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": None, "kernel_ms": scan_us * 1e-3, "launches": launches,
                         "algorithmic_bytes_per_launch": rows * dim * 4,
                         "note": "corpus resident in the 256 MiB Infinity Cache; HBM peak used as the denominator"},
            "batch_4096": {"value": 4096 * 5 / dtb, "unit": "queries/sec", "ms_per_step": dtb / 5 * 1e3,
                           "roofline": {"bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": None, "kernel_ms": kb / max(lb, 1),
                                        "whole_step_frac": 2.0 * 4096 * rows * dim / (dtb / 5) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                        "note": "frac = the selecting kernel alone; whole_step_frac = the same flops over the whole "
                                                "step (with the launches that seed its thresholds, and the merge)"}}}


def leg_ivf_share(torch, args, device, rows=12_500_000, dim=1024, nlist=16384, nprobe=64, k=10):
    """One GPU's share of configuration 'IVFFlat nlist 16384, nprobe 64, 100M x 1024 on 8 GPUs':
    12.5M rows.  Rows are nlist unit centres + noise (so that probing 64 lists finds the neighbours);
    centroids = the centres.  Batch 1 (per-query list scan, HBM-bound) and batch 4096 (list-major
    MFMA scan).  Algorithmic bytes of a list scan = nprobe * (rows / nlist) * dim * 4 per query
    (lists of average length)."""
    from rag_amd import DeviceIndex
    free, _ = torch.cuda.mem_get_info(device)
    if free < 1.2 * rows * dim * 4:          # the rows once (they are packed into their lists in place) + workspaces
        return {"skipped": f"needs {1.2 * rows * dim * 4 / 2**30:.0f} GiB of free HBM, {free / 2**30:.0f} free"}
    g = torch.Generator(device=device)
    g.manual_seed(99)
    C = torch.nn.functional.normalize(torch.randn(nlist, dim, device=device, generator=g), dim=1)
    ivf = DeviceIndex(dim, metric="ip", kind="ivf", device=device.index)
    ivf.train_ivf(nlist, centroids=C.cpu().numpy())
    ivf.reserve(rows)
    t_add = 0.0
    chunk = 500_000
    pick_rows = []
    for r0 in range(0, rows, chunk):
        c = min(chunk, rows - r0)
        a = torch.randint(0, nlist, (c,), device=device, generator=g)
        blk = torch.nn.functional.normalize(C[a] + (0.6 / dim ** 0.5) * torch.randn(c, dim, device=device, generator=g), dim=1)
        pick_rows.append(blk[:4096 * c // rows + 1].clone())
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        ivf.add(blk)
        t_add += time.perf_counter() - t0
        del blk, a
    base = torch.cat(pick_rows)[:4096]
    Q = torch.nn.functional.normalize(base + (0.3 / dim ** 0.5) * torch.randn(base.shape[0], dim, device=device, generator=g), dim=1)
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    ivf.search(Q[:1], k, nprobe=nprobe)              # the first search packs the lists
    torch.cuda.synchronize(device)
    t_pack = time.perf_counter() - t0
    out = {"workload": f"IVF-flat, {rows}x{dim} fp32 (one GPU's share of 100M), nlist {nlist}, nprobe {nprobe}, top-{k}",
           "add_assign_s": t_add, "first_search_pack_s": t_pack}
    per_query = nprobe * (rows / nlist) * dim * 4
    for nq, nsteps in ((1, 100), (4096, 3)):
        ivf.profile(False)
        ivf.search(Q[:nq], k, nprobe=nprobe)
        torch.cuda.synchronize(device)
        Dout = torch.empty((nq, k), dtype=torch.float32, device=device)      # caller-owned result buffers, as the C ABI has it
        Iout = torch.empty((nq, k), dtype=torch.int64, device=device)

        def run():
            for i in range(nsteps):
                j = i % 64 if nq == 1 else 0
                ivf.search_into(Q[j:j + nq], k, Dout.data_ptr(), Iout.data_ptr(), nprobe=nprobe)
            torch.cuda.synchronize(device)

        t0 = time.perf_counter()
        run()                                           # timed without the event pairs around the list scan ...
        dt = time.perf_counter() - t0
        ivf.profile(True)
        ivf.profile_read(reset=True)
        run()                                           # ... which this second pass records for the kernel time
        kms, launches = ivf.profile_read(reset=True)
        scan_ms = kms / max(launches, 1)
        leg = {"value": nq * nsteps / dt, "unit": "queries/sec", "batch": nq, "ms_per_step": dt / nsteps * 1e3,
               "path": ivf.last_path, "list_scan_kernel_ms": scan_ms}
        if nq == 1:
            gbs = per_query / (scan_ms * 1e-3) / 1e9 if launches else 0.0
            leg["roofline"] = {"bound": "hbm", "kernel": "flat_scan_kernel (list ranges)", "achieved": gbs, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "kernel_ms": scan_ms,
                               "launches": launches, "algorithmic_bytes_per_launch": per_query}
            leg["bytes_bound_us"] = (per_query + nlist * dim * 4) / (HBM_PEAK_GBS * 1e9) * 1e6
        else:
            # a list is read once per 32 queries that probe it.  With nq * nprobe / nlist = 16 probes per list on
            # average every list is read about once, so the bytes (the whole corpus) bound this kernel, not the flops
            # (2 * pairs * rows-per-list * dim); both are reported, the HBM view is the roofline
            import math
            tf = 2.0 * nq * nprobe * (rows / nlist) * dim / (scan_ms * 1e-3) / 1e12 if launches else 0.0
            per_list = nq * nprobe / nlist
            read_once = rows * dim * 4 * (1.0 - math.exp(-per_list))        # lists probed at all (Poisson), each read once
            gbs = read_once / (scan_ms * 1e-3) / 1e9 if launches else 0.0
            leg["roofline"] = {"bound": "hbm", "kernel": "ivf_lists_kernel", "achieved": gbs,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                               "kernel_ms": scan_ms, "launches": launches, "algorithmic_bytes_per_launch": read_once,
                               "note": "every probed list read once (lists probed by more than 32 queries are read again)",
                               "mfma_TFLOPs": tf, "mfma_frac_of_fp32_peak": tf / FP32_MFMA_PEAK_TFLOPS}
        out["batch_%d" % nq] = leg
    # answers of the two timed paths: the batch's lists against per-query searches of sampled queries (another scan
    # kernel over the same probed lists), every sampled score re-computed in fp64 from the stored rows, and the
    # query's own base row (a row of the corpus 0.3 sigma away) found
    ivf.profile(False)
    Db, Ib = ivf.search(Q, k, nprobe=nprobe)
    ok_b, ok_1 = bool((Db[:, 1:] <= Db[:, :-1]).all()), True
    import numpy as np
    for qi in sample_queries(Q.shape[0]):
        D1, I1 = ivf.search(Q[qi:qi + 1], k, nprobe=nprobe)
        ok_b = ok_b and same_topk(torch, Db[qi:qi + 1], Ib[qi:qi + 1], D1, I1)
        ids = I1[0].cpu().numpy()
        rows_h = ivf.reconstruct(ids[ids >= 0]).astype(np.float64)
        ref = rows_h @ Q[qi].cpu().numpy().astype(np.float64)
        got = D1[0].cpu().numpy().astype(np.float64)[ids >= 0]
        ok_1 = ok_1 and bool(np.all(np.abs(got - ref) <= 1e-5 * np.abs(ref) + 1e-6)) and bool(np.all(np.diff(got) <= 0))
    out["verified"] = {"batch_1": ok_1, "batch_4096": ok_b}
    ivf.close()
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus < 1:
        print("error: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become the launcher (fresh rank processes; nothing here has touched the GPU)
        sys.exit(spawn_ranks(args, argv))
    if args.stub_cpu:
        return stub_main(args)
    world, rank, local = rendezvous(args)
    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    if args.share_gpu and ndev > 0:
        if args.dist_backend != "gloo":
            print("error: --share-gpu needs --dist-backend gloo (RCCL wants one GPU per rank)", file=sys.stderr)
            sys.exit(2)
        local = local % ndev
    if local >= ndev:
        print(f"error: rank {rank} needs GPU {local} but {ndev} are visible", file=sys.stderr)
        sys.exit(3)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo")
    seen = check_world(dist, world)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    from rag_amd import DeviceIndex
    from rag_amd.sharded import ShardedIndex, shard_range

    lo, hi = shard_range(args.rows, rank, world)
    X = make_corpus(torch, lo, hi, args.dim, device)
    Q = make_queries(torch, args.big_batch, args.dim, device)
    index = DeviceIndex(args.dim, metric="ip", device=local)
    index.set_id_base(lo)
    index.adopt(X)
    sharded = ShardedIndex(index) if world > 1 else None
    search = (sharded.search if sharded else index.search)
    k = args.k
    bytes_per_row = args.dim * 4

    # ---------------- batch 1 ----------------
    index.profile(True)
    index.profile_read(reset=True)

    def step1(i):
        j = i % args.big_batch
        search(Q[j:j + 1], k)

    for i in range(args.warmup):
        step1(i)
    torch.cuda.synchronize(device)
    index.profile_read(reset=True)
    dt1 = timed_steps(torch, dist, world, step1, 0, args.steps, device)
    kms, launches = index.profile_read(reset=True)
    path1 = index.last_path
    qps1 = args.steps / dt1
    k_avg_ms = kms / max(launches, 1)
    achieved_gbs = (hi - lo) * bytes_per_row / (k_avg_ms * 1e-3) / 1e9 if launches else 0.0
    roof1 = {"bound": "hbm", "kernel": "flat_scan_kernel", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
             "traffic": measured_traffic("flat_scan_kernel", args.rows, args.dim, world),
             "kernel_ms": k_avg_ms, "launches": launches,
             "algorithmic_bytes_per_launch": (hi - lo) * bytes_per_row}

    # ---------------- what the timed legs answered, checked outside every timed region ----------------
    # single-query fp32 scans of a sample of the queries are the reference the batch legs are held to; the batch-1
    # leg itself is held to a 64-query batch (another kernel); every score is re-computed in fp64 from the rows
    verified = {}
    index.profile(False)
    ref_q = sample_queries(args.big_batch)
    ref_scan = {qi: search(Q[qi:qi + 1], k) for qi in ref_q}
    ref_path = index.last_path
    nb64 = min(64, args.big_batch)
    D64, I64 = search(Q[:nb64], k)
    other_path = index.last_path
    ok1 = True
    for qi in [q for q in ref_q if q < nb64]:
        Dq, Iq = ref_scan[qi]
        ok1 = ok1 and same_topk(torch, Dq, Iq, D64[qi:qi + 1], I64[qi:qi + 1])
    for qi in ref_q[:8]:
        Dq, Iq = ref_scan[qi]
        ok1 = ok1 and rescored_ok(torch, X, Q[qi:qi + 1], Dq, Iq, lo, hi)
    verified["batch_1"] = bool(ok1)
    verify_paths = {"single_query_reference": ref_path, "batch_1_checked_against_batch_%d" % nb64: other_path}
    del D64, I64
    index.profile(True)

    def verify_batch(B):
        """The batch a leg timed, answered once more (profiling off) and checked: every list re-scored in fp64,
        the sampled queries against their single-query scans."""
        index.profile(False)
        Db, Ib = search(Q[:B], k)
        ok = rescored_ok(torch, X, Q[:B], Db, Ib, lo, hi)
        for qi in [q for q in ref_q if q < B]:
            Dq, Iq = ref_scan[qi]
            ok = ok and same_topk(torch, Db[qi:qi + 1], Ib[qi:qi + 1], Dq, Iq)
        index.profile(True)
        index.profile_read(reset=True)
        return bool(ok)

    # ---------------- batch 4096 (and a serving-sized batch) ----------------
    def batch_leg(B, nsteps, peak_tflops=FP32_MFMA_PEAK_TFLOPS):
        def stepb(i):
            search(Q[:B], k)

        stepb(0)
        torch.cuda.synchronize(device)
        index.profile_read(reset=True)
        dtb = timed_steps(torch, dist, world, stepb, 0, nsteps, device)
        kmsb, launchesb = index.profile_read(reset=True)
        pathb = index.last_path
        flops = 2.0 * B * (hi - lo) * args.dim           # per step on this rank
        kb_ms = kmsb / max(nsteps, 1)                     # dominant-kernel time per step
        ach_tf = flops / (kb_ms * 1e-3) / 1e12 if kmsb > 0 else 0.0
        ach_gbs = (hi - lo) * bytes_per_row / (kb_ms * 1e-3) / 1e9 if kmsb > 0 else 0.0
        if pathb == "narrow":
            # up to 32 queries: one pass over the corpus, HBM-bound (the events bracket the sample launch,
            # its merge and the main launch, so every row is inside)
            return {"value": B * nsteps / dtb, "unit": "queries/sec", "batch": B, "steps": nsteps,
                    "ms_per_step": dtb / nsteps * 1e3, "path": pathb,
                    "roofline": {"bound": "hbm", "kernel": "narrow_topk_kernel", "achieved": ach_gbs, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS, "traffic": None,
                                 "kernel_ms_per_step": kb_ms, "launches": launchesb,
                                 "algorithmic_bytes_per_launch": (hi - lo) * bytes_per_row,
                                 "mfma_TFLOPs": ach_tf, "mfma_frac_of_fp32_peak": ach_tf / peak_tflops}}
        return {"value": B * nsteps / dtb, "unit": "queries/sec", "batch": B, "steps": nsteps,
                "ms_per_step": dtb / nsteps * 1e3, "path": pathb,
                "roofline": {"bound": "mfma", "achieved": ach_tf,
                             "peak": peak_tflops, "unit": "TFLOP/s",
                             "frac": ach_tf / peak_tflops,
                             "traffic": measured_traffic("gemm_topk_kernel", args.rows, args.dim, world)
                             if B == 4096 else None,
                             "kernel_ms_per_step": kb_ms, "launches": launchesb,
                             "algorithmic_flops_per_step": flops,
                             "corpus_GBps": ach_gbs, "corpus_frac_of_hbm_peak": ach_gbs / HBM_PEAK_GBS}}

    big = mid = None
    if not args.no_big:
        big = batch_leg(args.big_batch, args.big_steps or min(args.steps, 5))
        verified["batch_%d" % args.big_batch] = verify_batch(args.big_batch)
    mid16 = mid24 = None
    if args.mid_batch > 0:
        mid = batch_leg(min(args.mid_batch, args.big_batch), min(args.steps, 20))
        verified["batch_%d" % mid["batch"]] = verify_batch(mid["batch"])
        if args.mid_batch > 16:
            mid16 = batch_leg(16, min(args.steps, 20))
            verified["batch_16"] = verify_batch(16)
        if args.mid_batch > 24:
            # 17..32 queries take two 16-query column blocks: 24 queries cost what 32 cost
            mid24 = batch_leg(24, min(args.steps, 20))
            verified["batch_24"] = verify_batch(24)
    # opt-in mode, reported beside the fp32 numbers and never as `value`: bf16 first pass on the
    # matrix cores + fp32 re-scoring of the rows within the error margin (same answers)
    pre = pre1 = pre_mid = None
    if not args.no_big and not args.no_prefilter and args.big_batch >= 1024:
        index.set_prefilter(True)
        pre = batch_leg(args.big_batch, args.big_steps or min(args.steps, 5), BF16_MFMA_PEAK_TFLOPS)
        pre["roofline"]["traffic"] = None
        pre["dtype"] = "bf16 first pass (fp32 accumulate) + f32 re-score of the kept rows"
        verified["batch_%d_bf16_first_pass" % args.big_batch] = verify_batch(args.big_batch)
        if mid is not None:      # the serving-sized batch: one pass over the bf16 copy in the narrow shape
            pre_mid = batch_leg(mid["batch"], min(args.steps, 20), BF16_MFMA_PEAK_TFLOPS)
            pre_mid["dtype"] = pre["dtype"]
            verified["batch_%d_bf16_first_pass" % mid["batch"]] = verify_batch(mid["batch"])
            r = pre_mid["roofline"]                 # one pass over the bf16 rows (half the bytes): HBM-bound
            gb = r["corpus_GBps"] / 2
            pre_mid["roofline"] = {"bound": "hbm", "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": gb / HBM_PEAK_GBS, "traffic": None,
                                   "kernel_ms_per_step": r["kernel_ms_per_step"], "launches": r["launches"],
                                   "algorithmic_bytes_per_launch": (hi - lo) * args.dim * 2}
        # batch 1: scan of the bf16 copy (half the bytes), fp32 re-score of the 64 kept rows, acceptance test
        index.profile_read(reset=True)
        for i in range(3):
            step1(i)
        torch.cuda.synchronize(device)
        index.profile_read(reset=True)
        n1 = min(args.steps, 50)
        dtp = timed_steps(torch, dist, world, step1, 0, n1, device)
        kp, lp = index.profile_read(reset=True)
        gbs = (hi - lo) * args.dim * 2 / (kp / max(lp, 1) * 1e-3) / 1e9 if lp else 0.0
        pre1 = {"value": n1 / dtp, "unit": "queries/sec", "batch": 1, "steps": n1, "ms_per_step": dtp / n1 * 1e3,
                "path": index.last_path, "dtype": "bf16 rows x f32 query first pass + f32 re-score of the kept rows",
                "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                             "traffic": None, "kernel_ms": kp / max(lp, 1), "launches": lp,
                             "algorithmic_bytes_per_launch": (hi - lo) * args.dim * 2}}
        okp = True
        for qi in ref_q[:8]:
            Dp, Ip = search(Q[qi:qi + 1], k)
            okp = okp and same_topk(torch, Dp, Ip, *ref_scan[qi]) and rescored_ok(torch, X, Q[qi:qi + 1], Dp, Ip, lo, hi)
        verified["batch_1_bf16_first_pass"] = bool(okp)
        index.set_prefilter(False)
    index.profile(False)

    # ---------------- the other configurations (one GPU only; beside `value`, never instead of it) ----------------
    side = {}
    if world == 1 and not args.no_side:
        side["pipeline_hybrid_mmr"] = leg_pipeline(torch, index, Q, args, device)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        nrows = min(args.cpu_rows, hi - lo) if args.cpu_rows > 0 else hi - lo
        if args.cpu_rows <= 0:
            # the whole corpus on the host when it fits beside everything else with room to spare, else its first 1M rows
            try:
                import psutil
                avail = psutil.virtual_memory().available
            except Exception:
                avail = 0
            if avail < 2.5 * (hi - lo) * bytes_per_row or avail < (64 << 30):
                nrows = min(1_000_000, hi - lo)
        cpu = cpu_baseline(torch, X[:nrows], Q, k, args.rows, args.big_batch)

    if world > 1:
        # a rank re-scores the rows it holds: the run passes only if every rank's checks did
        on_host = dist.get_backend() == "gloo"
        flag = torch.tensor([1 if all(verified.values()) else 0], dtype=torch.int32, device="cpu" if on_host else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not int(flag.item()):
            verified["every_rank"] = False
    if rank == 0:
        out = {
            "metric": f"queries/sec, top-{k} exact IP, {args.rows}x{args.dim} fp32, batch 1",
            "value": qps1, "unit": "queries/sec", "n_gpus": world, "world_size": seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt1 / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"exact flat top-{k}, {args.rows}x{args.dim} fp32 unit vectors "
                                   f"(N(0,1) normalised, seeds 1234/4321), row-sharded over {world} GPU(s), "
                                   f"batch 1 (value) and batch {args.big_batch}",
                       "rows": args.rows, "dim": args.dim, "k": k, "batch": 1,
                       "rows_per_gpu": hi - lo, "path": path1,
                       "exchange": "none" if world == 1 else "1 all-gather of (score,id) + device merge",
                       "dist_backend": None if world == 1 else args.dist_backend},
            "roofline": roof1,
            "verified": verified, "verified_how": dict(verify_paths, rule=(
                "outside the timed regions: each leg's batch answered once more, every list re-scored in fp64 from the rows "
                "(1e-5 relative + 1e-6), ids of sampled queries (first and last of every 256-query block and 8 more) equal to "
                "their single-query fp32 scans except at ties within that bound")),
        }
        if big is not None:
            out["batch_%d" % args.big_batch] = big
        if mid is not None:
            out["batch_%d" % mid["batch"]] = mid
        if mid16 is not None:
            out["batch_16"] = mid16
        if mid24 is not None:
            out["batch_24"] = mid24
        if pre is not None:
            out["batch_%d_bf16_first_pass" % args.big_batch] = pre
        if pre_mid is not None:
            out["batch_%d_bf16_first_pass" % pre_mid["batch"]] = pre_mid
        if pre1 is not None:
            out["batch_1_bf16_first_pass"] = pre1
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if world == 1 and not args.no_side:
            # the corpus of the headline legs is released first: the IVF share is 51 GB of rows
            index.close()
            del X, index
            torch.cuda.empty_cache()
            side["keyword_bm25_rank"] = leg_keyword(torch, args, device)
            side["config1_100k_384"] = leg_small_corpus(torch, args, device)
            side["ivf_per_gpu_share"] = leg_ivf_share(torch, args, device)
        out.update(side)
        for name, leg in side.items():
            if isinstance(leg, dict) and "verified" in leg:
                verified[name] = leg["verified"] if isinstance(leg["verified"], bool) else all(leg["verified"].values())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not all(verified.values()):
        print(f"error: a timed leg returned answers that fail verification: {verified}", file=sys.stderr)
        sys.exit(5)


if __name__ == "__main__":
    main()
