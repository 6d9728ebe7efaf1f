"""Row-sharded exhaustive search over the GPUs of one node.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI).
Rank r owns the contiguous global rows ``[base_r, base_r + n_r)``; queries are
replicated; every rank runs the local scan, the ranks exchange their k best
``(score, id)`` per query with ONE all-gather (12 bytes per hit: latency-bound,
never bandwidth-bound on xGMI), and every rank merges the ``world*k`` candidates
per query on its own GPU, so each rank ends with the full answer.  The order
rule (score descending, lower id first) is the same at every level, so the
result equals a single-GPU search of the whole corpus.

The reference has no multi-GPU path; its closest loop is the per-document fan
out of vector_db/search_manager.py:879-925 (sequential RPCs, max-merge).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split, ceil-sized shards first (base_g = g * ceil(N / G))."""
    per = -(-n_total // world)
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def slot_layout(nq: int, k: int):
    """Layout of one rank's slot in the all-gather buffer, in the terms rag_merge_topk_device
    takes them.  A slot is ``words`` int64 words: nq*k ids, then nq*k fp32 scores packed two to a
    word.  Returns (words, score_offset_bytes, d_list_stride in floats, i_list_stride in int64s);
    slot g of the gathered buffer starts g*words words in."""
    nk = nq * k
    words = nk + (nk + 1) // 2
    return words, nk * 8, words * 2, words


class ShardedIndex:
    """``local_search(q, k, **kw) -> (D, I)`` with global ids, and
    ``merge(D_all, I_all) -> (D, I)`` for ``[world, nq, k]`` stacks.  By default
    both are the HIP paths of a ``DeviceIndex``; the CPU (gloo) tests of the
    exchange logic inject their own."""

    def __init__(self, local_index=None, *, group=None, row_base: int = 0,
                 local_search: Optional[Callable] = None, merge: Optional[Callable] = None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # an 'l2' index answers with distances, nearest first: they travel negated, so that the merge's
        # "larger first" (and its -FLT_MAX for an empty slot) holds, and are negated back after it
        self.nearest_first = getattr(local_index, "metric", None) == "l2"
        self.index = local_index
        self.row_base = int(row_base)        # global position of the local index's first row (hybrid_mmr)
        if local_search is None:
            if local_index is None:
                raise ValueError("need a local DeviceIndex or a local_search callable")
            local_search = local_index.search
        if merge is None:
            from .index import merge_topk
            merge = merge_topk
        self._search = local_search
        self._merge = merge
        self._gather_buf = None
        self.exchange_when_alone = False     # run the all-gather + merge even with one rank (tests)

    def _search_device(self, q, k: int, **kw):
        """CUDA-tensor queries on a DeviceIndex: the local search writes straight
        into the send buffer of the all-gather, and the merge reads the gathered
        slots through list strides -- no packing or unpacking kernels.
        Slot layout (int64 words): [nq*k ids | ceil(nq*k/2) words holding nq*k fp32 scores]."""
        import torch
        from .index import merge_topk_strided
        if q.dim() == 1:
            q = q[None, :]
        nq = q.shape[0]
        slot, d_off, d_stride, i_stride = slot_layout(nq, k)
        shape = (self.world * slot,)       # flat: every backend takes the concatenation form
        if self._gather_buf is None or self._gather_buf.shape != shape or self._gather_buf.device != q.device:
            self._gather_buf = torch.empty(shape, dtype=torch.int64, device=q.device)
            self._send_buf = torch.empty((slot,), dtype=torch.int64, device=q.device)
        buf = self._gather_buf
        mine = self._send_buf
        i_ptr = mine.data_ptr()
        self.index.search_into(q, k, i_ptr + d_off, i_ptr, **kw)
        if self.nearest_first:
            mine[nq * k:].view(torch.float32)[: nq * k].neg_()
        self._all_gather(buf, mine)
        D, I = merge_topk_strided(self.world, nq, k, buf.data_ptr() + d_off, d_stride, buf.data_ptr(), i_stride, q.device)
        return (D.neg_(), I) if self.nearest_first else (D, I)

    def _all_gather(self, out, mine):
        """The path's one exchange.  RCCL gathers device buffers in place; a gloo group (CPU
        rehearsals of the multi-rank path, several ranks sharing one GPU) is fed host copies."""
        if self.dist.is_initialized() and self.dist.get_backend(self.group) == "gloo" and mine.is_cuda:
            import torch
            host_out = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(host_out, mine.cpu(), group=self.group)
            out.copy_(host_out)
            return
        self.dist.all_gather_into_tensor(out, mine, group=self.group)

    def search(self, q, k: int, **kw):
        import torch
        if ((self.world > 1 or self.exchange_when_alone) and self.index is not None and self._search == self.index.search
                and type(q).__module__.startswith("torch") and q.is_cuda):
            return self._search_device(q, k, **kw)
        D, I = self._search(q, k, **kw)
        if self.world == 1:
            return D, I
        if self.nearest_first:
            D = -D
        was_numpy = isinstance(D, np.ndarray)
        if was_numpy:
            D, I = torch.from_numpy(D), torch.from_numpy(I)
        nq = D.shape[0]
        # one exchange: scores travel as the low 32 bits of an int64 next to the id
        packed = torch.empty((2, nq, k), dtype=torch.int64, device=D.device)
        packed[0] = I
        packed[1] = D.contiguous().view(torch.int32).to(torch.int64)
        shape = (self.world * 2, nq, k)          # rank r's block lands at [2r, 2r + 2)
        if self._gather_buf is None or self._gather_buf.shape != shape or self._gather_buf.device != D.device:
            self._gather_buf = torch.empty(shape, dtype=torch.int64, device=D.device)
        self.dist.all_gather_into_tensor(self._gather_buf, packed, group=self.group)
        gathered = self._gather_buf.view(self.world, 2, nq, k)
        I_all = gathered[:, 0].contiguous()
        D_all = gathered[:, 1].to(torch.int32).view(torch.float32).contiguous()
        Dm, Im = self._merge(D_all, I_all)
        if self.nearest_first:
            Dm = -Dm
        if was_numpy:
            return Dm.cpu().numpy(), Im.cpu().numpy()
        return Dm, Im

    def gather_rows(self, rows) -> np.ndarray:
        """Vectors of the global rows ``rows`` on every rank: each rank fills in the rows its index
        holds, the rest arrive through one all-reduce (a sum in which exactly one rank contributes to
        an element).  The candidate fetch of the multi-GPU fusion + MMR: <= 1024 rows."""
        import torch
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        n_local = len(self.index)
        mine = (rows >= self.row_base) & (rows < self.row_base + n_local)
        out = np.zeros((rows.size, self.index.dim), dtype=np.float32)
        if mine.any():
            out[mine] = self.index.reconstruct(rows[mine] - self.row_base)
        if self.world > 1:
            t = torch.from_numpy(out)
            on_device = self.dist.get_backend(self.group) != "gloo"
            if on_device:
                t = t.cuda(self.index.device)
            self.dist.all_reduce(t, group=self.group)
            out = t.cpu().numpy()
        return out

    def hybrid_mmr(self, q, topn: int, bm25_rows, bm25_scores, **kw):
        """Config "top-N -> hybrid fusion -> MMR" over the sharded corpus: the sharded search, the
        candidates' vectors gathered from the ranks that own them, fusion and MMR on every rank's own
        GPU (same inputs, same arithmetic: every rank returns the same answer).  ``row_base`` must be
        the local index's id base; keyword arguments and result as ``rag_amd.index.hybrid_mmr``."""
        import torch
        from .index import hybrid_mmr_over
        if self.nearest_first:
            raise ValueError("the fusion divides similarities by their maximum: not available on an 'l2' index")
        n_total = torch.tensor([len(self.index)], dtype=torch.int64)
        if self.world > 1:
            if self.dist.get_backend(self.group) != "gloo":
                n_total = n_total.cuda(self.index.device)
            self.dist.all_reduce(n_total, group=self.group)

        def search(qq, k, min_score=None, nprobe=1, row_mask=None):
            kw2 = {"min_score": min_score, "nprobe": nprobe}
            if row_mask is not None:
                kw2["row_mask"] = row_mask
            qt = torch.from_numpy(qq).cuda(self.index.device)
            D, I = self.search(qt, k, **kw2)
            return D.cpu().numpy(), I.cpu().numpy()

        return hybrid_mmr_over(search, self.gather_rows, int(n_total.item()), q, topn, bm25_rows, bm25_scores,
                               device=self.index.device, **kw)
