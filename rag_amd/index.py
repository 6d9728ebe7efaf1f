"""Device-resident vector index: a thin object over the rag_index_t handle.

One ``DeviceIndex`` stands where the reference keeps one Qdrant collection plus
its ``QdrantSearchOptimizer`` (vector_db/search_optimizer.py:68-86).  Host
arrays (numpy) go through the host-pointer entry points; ``torch`` tensors that
live on the index's GPU go through the ``*_device`` entry points on torch's
current stream without leaving HBM.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import check, lib

FLT_MAX = float(np.finfo(np.float32).max)
MAX_K = 1024


def _is_torch_cuda(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "is_cuda") and x.is_cuda


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceFilter:
    """A payload filter resolved to one byte per row and kept in device memory (``rag_filter_*``): what the
    reference sends with every search of a user (vector_db/search_manager.py:893) is uploaded once and passed
    as ``row_mask`` to any search of an index with that many rows on that device."""

    def __init__(self, mask, device: int = 0):
        m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
        self.device = int(device)
        self.n = int(m.size)
        self._f = C.c_void_p()
        check(lib.rag_filter_create(self.device, self.n, _ptr(m), C.byref(self._f)))

    def __len__(self) -> int:
        return self.n

    @property
    def device_ptr(self) -> int:
        return int(lib.rag_filter_device_bytes(self._f) or 0)

    def close(self) -> None:
        if getattr(self, "_f", None) is not None and self._f.value:
            lib.rag_filter_free(self._f)
            self._f = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _device_mask_ptr(row_mask, index):
    """Device address of a filter given as a DeviceFilter or a CUDA uint8 tensor (None: no filter)."""
    if row_mask is None:
        return None, None
    if isinstance(row_mask, DeviceFilter):
        if row_mask.n != len(index) or row_mask.device != index.device:
            raise ValueError("the filter was made for another row count or device")
        return C.c_void_p(row_mask.device_ptr), row_mask
    import torch
    if not _is_torch_cuda(row_mask) or row_mask.dtype != torch.uint8 or row_mask.numel() != len(index):
        raise ValueError("row_mask must be a DeviceFilter or a CUDA uint8 tensor with one byte per row")
    row_mask = row_mask.contiguous()
    return C.c_void_p(row_mask.data_ptr()), row_mask


class DeviceIndex:
    """metric: 'ip' | 'cosine' | 'l2'; kind: 'flat' | 'ivf'.

    An 'l2' index (flat only) answers with SQUARED distances, smallest first, empty slots as
    ``(-1, FLT_MAX)``; its ``min_score`` argument is the largest squared distance to return (None or a
    negative value: no limit).  Segment search and the fusion pipeline rank by similarity and refuse it."""

    def __init__(self, dim: int, metric: str = "cosine", kind: str = "flat", device: int = 0):
        metrics = {"ip": _lib.METRIC_IP, "dot": _lib.METRIC_IP, "cosine": _lib.METRIC_COSINE, "l2": _lib.METRIC_L2}
        kinds = {"flat": _lib.KIND_FLAT, "ivf": _lib.KIND_IVF}
        if metric not in metrics:
            raise ValueError(f"unknown metric {metric!r}")
        if kind not in kinds:
            raise ValueError(f"unknown index kind {kind!r}")
        self.dim = int(dim)
        self.metric = metric
        self.kind = kind
        self.device = int(device)
        self._h = C.c_void_p()
        self._keepalive = None
        check(lib.rag_index_create(self.dim, metrics[metric], kinds[kind], self.device, C.byref(self._h)))
        # searches run on torch's current stream, and torch keeps its streams for the life of the process
        check(lib.rag_index_set_stream_lifetime(self._h, 1))

    # ------------------------------------------------------------------ lifetime
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.rag_index_free(self._h)
            self._h = C.c_void_p()
            self._keepalive = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(lib.rag_index_ntotal(self._h))

    @property
    def ntotal(self) -> int:
        return len(self)

    @property
    def last_path(self) -> str:
        return (lib.rag_index_last_path(self._h) or b"").decode()

    # --------------------------------------------------------------- persistence
    def save(self, path: str) -> None:
        check(lib.rag_index_save(self._h, os.fsencode(path)))

    @classmethod
    def load(cls, path: str, device: int = 0) -> "DeviceIndex":
        h = C.c_void_p()
        check(lib.rag_index_load(os.fsencode(path), int(device), C.byref(h)))
        self = cls.__new__(cls)
        self._h = h
        self._keepalive = None
        check(lib.rag_index_set_stream_lifetime(h, 1))
        self.dim = int(lib.rag_index_dim(h))
        self.device = int(device)
        self.kind = "ivf" if int(lib.rag_index_kind(h)) == _lib.KIND_IVF else "flat"
        self.metric = {_lib.METRIC_COSINE: "cosine", _lib.METRIC_L2: "l2"}.get(int(lib.rag_index_metric(h)), "ip")
        return self

    # --------------------------------------------------------------- measurement
    def profile(self, enable: bool = True) -> None:
        check(lib.rag_index_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self, reset: bool = True):
        """(summed dominant-kernel milliseconds, launches) since the last reset."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        check(lib.rag_index_profile_read(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    # -------------------------------------------------------------------- ingest
    def reserve(self, nrows: int) -> None:
        check(lib.rag_index_reserve(self._h, int(nrows)))

    def set_id_base(self, base: int) -> None:
        check(lib.rag_index_set_id_base(self._h, int(base)))

    def set_prefilter(self, enabled: bool) -> None:
        """bf16 first pass for single queries (k <= 256) and batches (k <= 32) (results are re-scored in fp32 and equal
        those of the fp32 path); keeps a bf16 copy of the rows on the device."""
        check(lib.rag_index_set_prefilter(self._h, 1 if enabled else 0))

    def add(self, x, ids=None) -> None:
        ids_np = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
        if _is_torch_cuda(x):
            import torch
            if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != self.dim:
                raise ValueError(f"expected float32 [n, {self.dim}] tensor")
            if x.device.index != self.device:
                raise ValueError("tensor is on another device than the index")
            x = x.contiguous()
            torch.cuda.current_stream(x.device).synchronize()
            n = x.shape[0]
            if ids_np is not None and ids_np.shape != (n,):
                raise ValueError("ids must have one entry per row")
            check(lib.rag_index_add_device(self._h, n, C.c_void_p(x.data_ptr()), _ptr(ids_np)))
            return
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected float32 [n, {self.dim}] array, got {x.shape}")
        if ids_np is not None and ids_np.shape != (x.shape[0],):
            raise ValueError("ids must have one entry per row")
        check(lib.rag_index_add(self._h, x.shape[0], _ptr(x), _ptr(ids_np)))

    def adopt(self, x) -> None:
        """Search in place in a torch CUDA tensor [n, dim] fp32 (no copy; the index
        keeps a reference).  COSINE indexes normalise the rows in place."""
        import torch
        if not _is_torch_cuda(x) or x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected a CUDA float32 [n, {self.dim}] tensor")
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        if x.device.index != self.device:
            raise ValueError("tensor is on another device than the index")
        torch.cuda.current_stream(x.device).synchronize()
        check(lib.rag_index_adopt_device(self._h, x.shape[0], C.c_void_p(x.data_ptr())))
        self._keepalive = x

    def set_active(self, rows, active: bool) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        check(lib.rag_index_set_active(self._h, rows.size, _ptr(rows), 1 if active else 0))

    def reconstruct(self, rows) -> np.ndarray:
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        out = np.empty((rows.size, self.dim), dtype=np.float32)
        check(lib.rag_index_reconstruct(self._h, rows.size, _ptr(rows), _ptr(out)))
        return out

    def train_ivf(self, nlist: int, centroids=None, sample=None) -> None:
        c = None if centroids is None else np.ascontiguousarray(centroids, dtype=np.float32)
        s = None if sample is None else np.ascontiguousarray(sample, dtype=np.float32)
        if c is not None and c.shape != (nlist, self.dim):
            raise ValueError("centroids must be [nlist, dim]")
        ns = 0 if s is None else s.shape[0]
        check(lib.rag_index_train_ivf(self._h, int(nlist), _ptr(c), ns, _ptr(s)))

    def make_filter(self, mask) -> DeviceFilter:
        """``mask`` (one byte per row, 1 = may be returned) uploaded once; pass the result as ``row_mask``."""
        f = DeviceFilter(mask, self.device)
        if f.n != len(self):
            f.close()
            raise ValueError("row_mask must have one byte per row")
        return f

    # -------------------------------------------------------------------- search
    def search_into(self, q, k: int, d_ptr: int, i_ptr: int, min_score: Optional[float] = None,
                    nprobe: int = 1, row_mask=None) -> None:
        """Device search of CUDA-tensor queries with the results written to caller
        memory: nq*k fp32 scores at d_ptr and nq*k int64 ids at i_ptr (device
        addresses), enqueued on torch's current stream."""
        import torch
        if q.dim() == 1:
            q = q[None, :]
        if not _is_torch_cuda(q) or q.dtype != torch.float32 or q.shape[1] != self.dim:
            raise ValueError(f"expected a CUDA float32 [nq, {self.dim}] tensor")
        if not 1 <= int(k) <= MAX_K:
            raise ValueError(f"k must be in [1, {MAX_K}]")
        q = q.contiguous()
        m_ptr, _keep = _device_mask_ptr(row_mask, self)
        ms = -FLT_MAX if min_score is None else float(min_score)
        stream = torch.cuda.current_stream(q.device).cuda_stream
        check(lib.rag_index_search_device(self._h, q.shape[0], C.c_void_p(q.data_ptr()), int(k), ms, int(nprobe),
                                          m_ptr, C.c_void_p(d_ptr), C.c_void_p(i_ptr), C.c_void_p(stream)))

    def search_segments(self, q, segments, k: int, min_score: Optional[float] = None, row_mask=None):
        """The k best of every (query, row range) pair in one device pass.
        ``segments``: [(begin, end), ...] row ranges.  Returns numpy (D, I) of shape
        [nq, nseg, k]."""
        k = int(k)
        if not 1 <= k <= MAX_K:
            raise ValueError(f"k must be in [1, {MAX_K}]")
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected float32 [nq, {self.dim}] array, got {q.shape}")
        seg = np.ascontiguousarray(segments, dtype=np.int64).reshape(-1, 2)
        sb = np.ascontiguousarray(seg[:, 0])
        se = np.ascontiguousarray(seg[:, 1])
        nq, ns = q.shape[0], seg.shape[0]
        D = np.full((nq, ns, k), -FLT_MAX, dtype=np.float32)
        I = np.full((nq, ns, k), -1, dtype=np.int64)
        if isinstance(row_mask, DeviceFilter):
            ms = -FLT_MAX if min_score is None else float(min_score)
            check(lib.rag_index_search_segments_filtered(self._h, nq, _ptr(q), ns, _ptr(sb), _ptr(se), k, ms, row_mask._f,
                                                         _ptr(D), _ptr(I)))
            return D, I
        m = None
        if row_mask is not None:
            m = np.ascontiguousarray(row_mask, dtype=np.uint8).reshape(-1)
            if m.size != len(self):
                raise ValueError("row_mask must have one byte per row")
        ms = -FLT_MAX if min_score is None else float(min_score)
        check(lib.rag_index_search_segments(self._h, nq, _ptr(q), ns, _ptr(sb), _ptr(se), k, ms, _ptr(m),
                                            _ptr(D), _ptr(I)))
        return D, I

    def search_segments_fused(self, q, weights, segments, k: int, limit: int, min_score: Optional[float] = None,
                              row_mask=None):
        """``search_segments`` followed on the device by the weighted best-per-row fusion
        and the ranking of vector_db/search_manager.py:898-932.  ``weights``: one fp64
        per query vector.  Returns numpy (rows int64, scores float64, strategy int32),
        at most ``limit`` long, best first."""
        k, limit = int(k), int(limit)
        if not 1 <= k <= MAX_K or not 1 <= limit <= MAX_K:
            raise ValueError(f"k and limit must be in [1, {MAX_K}]")
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected float32 [nq, {self.dim}] array, got {q.shape}")
        w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
        if w.size != q.shape[0]:
            raise ValueError("one weight per query vector")
        seg = np.ascontiguousarray(segments, dtype=np.int64).reshape(-1, 2)
        sb = np.ascontiguousarray(seg[:, 0])
        se = np.ascontiguousarray(seg[:, 1])
        rows = np.full(limit, -1, dtype=np.int64)
        scores = np.zeros(limit, dtype=np.float64)
        strategy = np.full(limit, -1, dtype=np.int32)
        n = C.c_int32(0)
        ms = -FLT_MAX if min_score is None else float(min_score)
        if isinstance(row_mask, DeviceFilter):
            check(lib.rag_index_search_segments_fused_filtered(self._h, q.shape[0], _ptr(q), _ptr(w), seg.shape[0], _ptr(sb), _ptr(se),
                                                               k, ms, row_mask._f, limit, _ptr(rows), _ptr(scores), _ptr(strategy),
                                                               C.byref(n)))
            return rows[:n.value], scores[:n.value], strategy[:n.value]
        m = None
        if row_mask is not None:
            m = np.ascontiguousarray(row_mask, dtype=np.uint8).reshape(-1)
            if m.size != len(self):
                raise ValueError("row_mask must have one byte per row")
        check(lib.rag_index_search_segments_fused(self._h, q.shape[0], _ptr(q), _ptr(w), seg.shape[0], _ptr(sb), _ptr(se),
                                                  k, ms, _ptr(m), limit, _ptr(rows), _ptr(scores), _ptr(strategy),
                                                  C.byref(n)))
        return rows[:n.value], scores[:n.value], strategy[:n.value]

    def search(self, q, k: int, min_score: Optional[float] = None, nprobe: int = 1,
               row_mask=None) -> Tuple[object, object]:
        """Returns (D [nq,k] float32, I [nq,k] int64), numpy for numpy queries and
        CUDA tensors for CUDA-tensor queries."""
        k = int(k)
        if not 1 <= k <= MAX_K:
            raise ValueError(f"k must be in [1, {MAX_K}]")
        ms = -FLT_MAX if min_score is None else float(min_score)
        if _is_torch_cuda(q):
            import torch
            if q.dim() == 1:
                q = q[None, :]
            if q.dtype != torch.float32 or q.shape[1] != self.dim:
                raise ValueError(f"expected float32 [nq, {self.dim}] tensor")
            if q.device.index != self.device:
                raise ValueError("tensor is on another device than the index")
            q = q.contiguous()
            nq = q.shape[0]
            D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            m_ptr, row_mask = _device_mask_ptr(row_mask, self)
            stream = torch.cuda.current_stream(q.device).cuda_stream
            check(lib.rag_index_search_device(self._h, nq, C.c_void_p(q.data_ptr()), k, ms, int(nprobe), m_ptr,
                                              C.c_void_p(D.data_ptr()), C.c_void_p(I.data_ptr()),
                                              C.c_void_p(stream)))
            return D, I
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected float32 [nq, {self.dim}] array, got {q.shape}")
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        if isinstance(row_mask, DeviceFilter):
            check(lib.rag_index_search_filtered(self._h, nq, _ptr(q), k, ms, int(nprobe), row_mask._f, _ptr(D), _ptr(I)))
            return D, I
        m = None
        if row_mask is not None:
            m = np.ascontiguousarray(row_mask, dtype=np.uint8).reshape(-1)
            if m.size != len(self):
                raise ValueError("row_mask must have one byte per row")
        check(lib.rag_index_search(self._h, nq, _ptr(q), k, ms, int(nprobe), _ptr(m), _ptr(D), _ptr(I)))
        return D, I


_MMR_METRICS = {"cosine": _lib.MMR_COSINE, "dot_product": _lib.MMR_DOT, "euclidean": _lib.MMR_EUCLIDEAN}


class MultiDeviceFilter:
    """A caller filter of a ``MultiDeviceIndex`` kept on its devices (``rag_multi_filter_*``)."""

    def __init__(self, index: "MultiDeviceIndex", mask: np.ndarray):
        self.n = int(mask.size)
        self._f = C.c_void_p()
        check(lib.rag_multi_filter_create(index._h, _ptr(mask), C.byref(self._f)))

    def __len__(self) -> int:
        return self.n

    def close(self) -> None:
        if getattr(self, "_f", None) is not None and self._f.value:
            lib.rag_multi_filter_free(self._f)
            self._f = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiDeviceIndex:
    """One index over several GPUs of this process (``rag_multi_*``): rows of every ``add`` are
    dealt over the devices, a search runs on all of them at once and the answers are merged.
    For callers without a process group -- the reference's server answers a search with a
    blocking call inside one worker process (vector_db/search_optimizer.py:213-222).
    ``devices`` may name a GPU more than once.  numpy in, numpy out, like ``DeviceIndex.search``."""

    def __init__(self, dim: int, devices: Sequence[int], metric: str = "ip", kind: str = "flat"):
        metrics = {"ip": _lib.METRIC_IP, "dot": _lib.METRIC_IP, "cosine": _lib.METRIC_COSINE, "l2": _lib.METRIC_L2}
        kinds = {"flat": _lib.KIND_FLAT, "ivf": _lib.KIND_IVF}
        if metric not in metrics or kind not in kinds:
            raise ValueError(f"unknown metric {metric!r} or index kind {kind!r}")
        self.dim, self.metric, self.kind = int(dim), metric, kind
        self.devices = [int(d) for d in devices]
        self._explicit_ids = False
        ids = (C.c_int * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        check(lib.rag_multi_create(self.dim, metrics[metric], kinds[kind], len(self.devices), ids, C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.rag_multi_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(lib.rag_multi_ntotal(self._h))

    @property
    def last_merge(self) -> str:
        """Where the last search merged the devices' lists: 'device' (peer copies into the first device and
        one merge kernel there) or 'host' (a device cannot reach the first one peer to peer)."""
        return (lib.rag_multi_last_merge(self._h) or b"").decode()

    def shard_rows(self) -> List[int]:
        return [int(lib.rag_multi_shard_rows(self._h, s)) for s in range(len(self.devices))]

    def set_prefilter(self, enabled: bool) -> None:
        check(lib.rag_multi_set_prefilter(self._h, 1 if enabled else 0))

    def train_ivf(self, nlist: int, centroids=None, sample=None) -> None:
        c = None if centroids is None else np.ascontiguousarray(centroids, dtype=np.float32)
        s = None if sample is None else np.ascontiguousarray(sample, dtype=np.float32)
        if c is not None and c.shape != (nlist, self.dim):
            raise ValueError("centroids must be [nlist, dim]")
        check(lib.rag_multi_train_ivf(self._h, int(nlist), _ptr(c), 0 if s is None else s.shape[0], _ptr(s)))

    def add(self, x, ids=None) -> None:
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected float32 [n, {self.dim}] array, got {x.shape}")
        ids_np = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
        if ids_np is not None and ids_np.shape != (x.shape[0],):
            raise ValueError("ids must have one entry per row")
        check(lib.rag_multi_add(self._h, x.shape[0], _ptr(x), _ptr(ids_np)))
        self._explicit_ids = self._explicit_ids or ids_np is not None

    def set_active(self, rows, active: bool) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        check(lib.rag_multi_set_active(self._h, rows.size, _ptr(rows), 1 if active else 0))

    def reconstruct(self, rows) -> np.ndarray:
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        out = np.empty((rows.size, self.dim), dtype=np.float32)
        check(lib.rag_multi_reconstruct(self._h, rows.size, _ptr(rows), _ptr(out)))
        return out

    def search(self, q, k: int, min_score: Optional[float] = None, nprobe: int = 1, row_mask=None):
        k = int(k)
        if not 1 <= k <= MAX_K:
            raise ValueError(f"k must be in [1, {MAX_K}]")
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected float32 [nq, {self.dim}] array, got {q.shape}")
        D = np.full((q.shape[0], k), -FLT_MAX, dtype=np.float32)
        I = np.full((q.shape[0], k), -1, dtype=np.int64)
        ms = -FLT_MAX if min_score is None else float(min_score)
        if isinstance(row_mask, MultiDeviceFilter):
            check(lib.rag_multi_search_filtered(self._h, q.shape[0], _ptr(q), k, ms, int(nprobe), row_mask._f, _ptr(D), _ptr(I)))
            return D, I
        m = None
        if row_mask is not None:
            m = np.ascontiguousarray(row_mask, dtype=np.uint8).reshape(-1)
            if m.size != len(self):
                raise ValueError("row_mask must have one byte per row")
        check(lib.rag_multi_search(self._h, q.shape[0], _ptr(q), k, ms, int(nprobe), _ptr(m), _ptr(D), _ptr(I)))
        return D, I

    def make_filter(self, mask) -> "MultiDeviceFilter":
        """``mask`` (one byte per row in insertion order) cut into the shards' row orders and uploaded to
        their devices once; pass the result as ``row_mask``."""
        m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
        if m.size != len(self):
            raise ValueError("row_mask must have one byte per row")
        return MultiDeviceFilter(self, m)

    def hybrid_mmr(self, q, topn: int, bm25_rows, bm25_scores, **kw):
        """Search over every device -> hybrid fusion with the keyword hits (global row positions) -> MMR
        on the first device, the candidates' vectors fetched from the devices that hold them.  Keyword
        arguments and result as ``hybrid_mmr``; rows must have been added without explicit ids."""
        if self._explicit_ids:
            raise ValueError("hybrid_mmr needs row positions as ids: add rows without explicit ids")
        if self.metric == "l2":
            raise ValueError("the fusion divides similarities by their maximum: not available on an 'l2' index")
        return hybrid_mmr_over(self.search, self.reconstruct, len(self), q, topn, bm25_rows, bm25_scores,
                               device=self.devices[0], **kw)


def hybrid_mmr(index: "DeviceIndex", q, topn: int, bm25_rows, bm25_scores, *, fuse_limit: Optional[int] = None,
               max_results: int = 10, semantic_weight: float = 0.65, bm25_weight: float = 0.35,
               mmr_lambda: float = 0.6, mmr_threshold: float = 0.8, mmr_metric: str = "cosine",
               min_score: Optional[float] = None, nprobe: int = 1, row_mask=None, want_fused: bool = False):
    """Search top-``topn`` -> hybrid fusion with the keyword hits ``(bm25_rows, bm25_scores)`` (row
    positions, BM25 scores) -> MMR -> ``max_results`` hits, as ONE call into the library
    (vector_db/search_manager.py:938-970 and :1620-1681).

    numpy inputs: returns a dict with ``ids``, ``scores`` (fused), ``picked`` (positions in the
    fused list), ``status`` (1 = a zero maximum, the reference's fall-back to the semantic list)
    and, with ``want_fused``, ``fused_rows / fused_scores / fused_sem / fused_bm``.
    CUDA-tensor query (bm25 lists as CUDA tensors too): nothing is synchronised; the dict holds
    CUDA tensors plus ``n_out`` / ``n_fused`` / ``status`` as one-element tensors."""
    topn = int(topn)
    if fuse_limit is None:
        fuse_limit = min(1024, topn + len(bm25_rows))
    fuse_limit = int(fuse_limit)
    ms = -FLT_MAX if min_score is None else float(min_score)
    metric = _MMR_METRICS.get(mmr_metric, _lib.MMR_COSINE)
    out = _lib.HybridOut()
    if _is_torch_cuda(q):
        import torch
        q = q.reshape(-1).contiguous()
        if q.dtype != torch.float32 or q.numel() != index.dim:
            raise ValueError(f"expected a float32 query of {index.dim} elements")
        dev = q.device
        bmr = bm25_rows.to(device=dev, dtype=torch.int64).contiguous()
        bms = bm25_scores.to(device=dev, dtype=torch.float64).contiguous()
        nb = bmr.numel()
        res = {"ids": torch.empty(max_results, dtype=torch.int64, device=dev),
               "scores": torch.empty(max_results, dtype=torch.float64, device=dev),
               "picked": torch.empty(max_results, dtype=torch.int32, device=dev),
               "n_out": torch.zeros(1, dtype=torch.int32, device=dev),
               "n_fused": torch.zeros(1, dtype=torch.int32, device=dev),
               "status": torch.zeros(1, dtype=torch.int32, device=dev)}
        if want_fused:
            res["fused_rows"] = torch.empty(fuse_limit, dtype=torch.int64, device=dev)
            for name in ("fused_scores", "fused_sem", "fused_bm"):
                res[name] = torch.empty(fuse_limit, dtype=torch.float64, device=dev)
        for name, t in res.items():
            setattr(out, name, t.data_ptr())
        m_ptr, row_mask = _device_mask_ptr(row_mask, index)
        stream = torch.cuda.current_stream(dev).cuda_stream
        check(lib.rag_index_hybrid_mmr_device(index._h, C.c_void_p(q.data_ptr()), topn, ms, int(nprobe), m_ptr, nb,
                                              C.c_void_p(bmr.data_ptr()) if nb else None,
                                              C.c_void_p(bms.data_ptr()) if nb else None,
                                              float(semantic_weight), float(bm25_weight), fuse_limit, float(mmr_lambda),
                                              float(mmr_threshold), metric, int(max_results), C.byref(out),
                                              C.c_void_p(stream)))
        res["_keepalive"] = (q, bmr, bms, row_mask)
        return res
    q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1)
    if q.size != index.dim:
        raise ValueError(f"expected a query of {index.dim} elements")
    bmr = np.ascontiguousarray(bm25_rows, dtype=np.int64).reshape(-1)
    bms = np.ascontiguousarray(bm25_scores, dtype=np.float64).reshape(-1)
    if bmr.size != bms.size:
        raise ValueError("one BM25 score per row")
    m = None
    if row_mask is not None:
        m = np.ascontiguousarray(row_mask, dtype=np.uint8).reshape(-1)
        if m.size != len(index):
            raise ValueError("row_mask must have one byte per row")
    ids = np.empty(max_results, dtype=np.int64)
    scores = np.empty(max_results, dtype=np.float64)
    picked = np.empty(max_results, dtype=np.int32)
    counts = np.zeros(3, dtype=np.int32)
    out.ids, out.scores, out.picked = ids.ctypes.data, scores.ctypes.data, picked.ctypes.data
    out.n_out, out.n_fused, out.status = counts.ctypes.data, counts.ctypes.data + 4, counts.ctypes.data + 8
    fused = {}
    if want_fused:
        fused["fused_rows"] = np.empty(fuse_limit, dtype=np.int64)
        for name in ("fused_scores", "fused_sem", "fused_bm"):
            fused[name] = np.empty(fuse_limit, dtype=np.float64)
        for name, a in fused.items():
            setattr(out, name, a.ctypes.data)
    check(lib.rag_index_hybrid_mmr(index._h, _ptr(q), topn, ms, int(nprobe), _ptr(m), bmr.size,
                                   _ptr(bmr) if bmr.size else None, _ptr(bms) if bms.size else None,
                                   float(semantic_weight), float(bm25_weight), fuse_limit, float(mmr_lambda),
                                   float(mmr_threshold), metric, int(max_results), C.byref(out)))
    n_out, n_fused, status = (int(c) for c in counts)
    res = {"ids": ids[:n_out], "scores": scores[:n_out], "picked": picked[:n_out], "status": status, "n_fused": n_fused}
    for name, a in fused.items():
        res[name] = a[:n_fused]
    return res


def hybrid_mmr_batch(index: "DeviceIndex", q, topn: int, bm25_rows, bm25_scores, *, fuse_limit: Optional[int] = None,
                     max_results: int = 10, semantic_weight: float = 0.65, bm25_weight: float = 0.35, mmr_lambda: float = 0.6,
                     mmr_threshold: float = 0.8, mmr_metric: str = "cosine", min_score: Optional[float] = None, nprobe: int = 1,
                     row_mask=None):
    """``hybrid_mmr`` for a batch of queries on the device: ONE search of all queries for their ``topn`` semantic hits, then
    fusion and MMR per query behind it on torch's current stream (rag_index_hybrid_mmr_batch_device).

    ``q``: CUDA float32 ``[nq, dim]``; ``bm25_rows`` / ``bm25_scores``: CUDA ``[nq, nb]`` (int64 row positions, -1 pads a
    shorter list; float64 scores); numpy arrays of the same shapes are staged to the device and answered with numpy arrays.
    Returns CUDA tensors ``(ids [nq, M] int64, -1 past a query's count; scores [nq, M]
    float64, the fused scores; n_out [nq] int32)`` with ``M = min(max_results, fuse_limit, topn + nb)``; nothing is synchronised."""
    import torch
    if isinstance(q, np.ndarray):
        # host arrays: staged to the index's device, answered there, copied back (one synchronisation at the end)
        dev = torch.device("cuda", index.device)
        mask_t = None if row_mask is None else torch.from_numpy(np.ascontiguousarray(row_mask, dtype=np.uint8)).to(dev)
        out = hybrid_mmr_batch(index, torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(dev), topn,
                               torch.from_numpy(np.ascontiguousarray(bm25_rows, dtype=np.int64)).to(dev),
                               torch.from_numpy(np.ascontiguousarray(bm25_scores, dtype=np.float64)).to(dev), fuse_limit=fuse_limit,
                               max_results=max_results, semantic_weight=semantic_weight, bm25_weight=bm25_weight, mmr_lambda=mmr_lambda,
                               mmr_threshold=mmr_threshold, mmr_metric=mmr_metric, min_score=min_score, nprobe=nprobe, row_mask=mask_t)
        return tuple(t.cpu().numpy() for t in out)
    if not _is_torch_cuda(q) or q.dtype != torch.float32 or q.dim() != 2 or q.shape[1] != index.dim:
        raise ValueError(f"expected a CUDA float32 [nq, {index.dim}] tensor")
    q = q.contiguous()
    nq, dev = q.shape[0], q.device
    bmr = bm25_rows.to(device=dev, dtype=torch.int64).reshape(nq, -1).contiguous()
    bms = bm25_scores.to(device=dev, dtype=torch.float64).reshape(nq, -1).contiguous()
    if bmr.shape != bms.shape:
        raise ValueError("one BM25 score per row")
    nb = bmr.shape[1]
    topn = int(topn)
    if fuse_limit is None:
        fuse_limit = min(1024, topn + nb)
    fuse_limit = int(fuse_limit)
    m_out = min(int(max_results), fuse_limit, topn + nb)
    ids = torch.empty((nq, m_out), dtype=torch.int64, device=dev)
    scores = torch.zeros((nq, m_out), dtype=torch.float64, device=dev)
    n_out = torch.empty(nq, dtype=torch.int32, device=dev)
    m_ptr, row_mask = _device_mask_ptr(row_mask, index)
    ms = -FLT_MAX if min_score is None else float(min_score)
    stream = torch.cuda.current_stream(dev).cuda_stream
    check(lib.rag_index_hybrid_mmr_batch_device(index._h, nq, C.c_void_p(q.data_ptr()), topn, ms, int(nprobe), m_ptr, nb,
                                                C.c_void_p(bmr.data_ptr()) if nb else None,
                                                C.c_void_p(bms.data_ptr()) if nb else None, float(semantic_weight),
                                                float(bm25_weight), fuse_limit, float(mmr_lambda), float(mmr_threshold),
                                                _MMR_METRICS.get(mmr_metric, _lib.MMR_COSINE), int(max_results),
                                                C.c_void_p(ids.data_ptr()), C.c_void_p(scores.data_ptr()),
                                                C.c_void_p(n_out.data_ptr()), C.c_void_p(stream)))
    ids._rag_keepalive = (q, bmr, bms, row_mask)
    return ids, scores, n_out


def hybrid_mmr_over(search, reconstruct, ntotal: int, q, topn: int, bm25_rows, bm25_scores, *,
                    fuse_limit: Optional[int] = None, max_results: int = 10, semantic_weight: float = 0.65,
                    bm25_weight: float = 0.35, mmr_lambda: float = 0.6, mmr_threshold: float = 0.8,
                    mmr_metric: str = "cosine", min_score: Optional[float] = None, nprobe: int = 1, row_mask=None,
                    want_fused: bool = False, device: int = 0):
    """``hybrid_mmr`` for a corpus that is spread over several devices: the search is whatever
    ``search(q, topn, min_score=, nprobe=, row_mask=)`` answers with global row positions as ids,
    the candidates' vectors come from ``reconstruct(rows)`` (the <= 1024 fused candidates, fetched
    from the devices that own them), and fusion (``rag_fuse_maxnorm``) and MMR (``rag_mmr_select``)
    run on ``device``.  Same arithmetic and order rules as ``rag_index_hybrid_mmr``
    (vector_db/search_manager.py:938-970, :1023-1101, :1620-1681), same result dictionary."""
    from . import fusion
    from .mmr_diversifier import mmr_select_device, normalise_relevance
    topn = int(topn)
    bmr = np.ascontiguousarray(bm25_rows, dtype=np.int64).reshape(-1)
    bms = np.ascontiguousarray(bm25_scores, dtype=np.float64).reshape(-1)
    if bmr.size != bms.size:
        raise ValueError("one BM25 score per row")
    if fuse_limit is None:
        fuse_limit = min(1024, topn + bmr.size)
    fuse_limit = int(fuse_limit)
    if not 1 <= fuse_limit <= 1024 or not 1 <= max_results <= 1024:
        raise ValueError("fuse_limit and max_results must be in [1, 1024]")
    q = np.ascontiguousarray(q, dtype=np.float32).reshape(1, -1)
    D, I = search(q, topn, min_score=min_score, nprobe=nprobe, row_mask=row_mask)
    D, I = np.asarray(D)[0], np.asarray(I)[0]
    sem = [(int(r), float(s)) for s, r in zip(D, I) if r >= 0]
    bm = [(int(r), float(s)) for r, s in zip(bmr, bms) if 0 <= r < ntotal]
    ranked = fusion.hybrid_fuse(sem, bm, fuse_limit, semantic_weight, bm25_weight)
    status = 0
    if ranked is None:                       # a zero maximum: the semantic list as it is (:1098-1101), cut
        status = 1
        ranked = [(r, s, s, 0.0, True, False) for r, s in sem][:fuse_limit]
    f_rows = np.array([t[0] for t in ranked], dtype=np.int64)
    f_scores = np.array([t[1] for t in ranked], dtype=np.float64)
    picked: List[int] = []
    if f_rows.size:
        emb = np.ascontiguousarray(reconstruct(f_rows), dtype=np.float32)
        picked = mmr_select_device(emb, normalise_relevance(f_scores.tolist()), mmr_lambda, mmr_threshold, mmr_metric,
                                   min(int(max_results), f_rows.size), device=device)
    picked_np = np.array(picked, dtype=np.int32)
    res = {"ids": f_rows[picked_np] if picked else np.zeros(0, np.int64),
           "scores": f_scores[picked_np] if picked else np.zeros(0, np.float64),
           "picked": picked_np, "status": status, "n_fused": int(f_rows.size)}
    if want_fused:
        res["fused_rows"] = f_rows
        res["fused_scores"] = f_scores
        res["fused_sem"] = np.array([t[2] for t in ranked], dtype=np.float64)
        res["fused_bm"] = np.array([t[3] for t in ranked], dtype=np.float64)
    return res


def merge_topk_strided(nlists: int, nq: int, k: int, d_ptr: int, d_stride: int, i_ptr: int, i_stride: int, device):
    """Merge `nlists` result sets that sit at d_ptr + g*d_stride floats / i_ptr +
    g*i_stride int64s in device memory (e.g. the slots of an all-gather buffer)."""
    import torch
    Do = torch.empty((nq, k), dtype=torch.float32, device=device)
    Io = torch.empty((nq, k), dtype=torch.int64, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream
    check(lib.rag_merge_topk_device(nlists, nq, k, C.c_void_p(d_ptr), d_stride, C.c_void_p(i_ptr), i_stride,
                                    C.c_void_p(Do.data_ptr()), C.c_void_p(Io.data_ptr()), device.index,
                                    C.c_void_p(stream)))
    return Do, Io


def merge_topk(D, I, device: Optional[int] = None):
    """Merge result sets stacked as [nlists, nq, k] CUDA tensors into [nq, k]
    (the step after the all-gather of per-shard results)."""
    import torch
    if not (_is_torch_cuda(D) and _is_torch_cuda(I)) or D.dim() != 3 or D.shape != I.shape:
        raise ValueError("expected CUDA tensors D, I of shape [nlists, nq, k]")
    if D.dtype != torch.float32 or I.dtype != torch.int64:
        raise ValueError("expected float32 scores and int64 ids")
    D = D.contiguous()
    I = I.contiguous()
    nl, nq, k = D.shape
    dev = D.device.index if device is None else device
    Do = torch.empty((nq, k), dtype=torch.float32, device=D.device)
    Io = torch.empty((nq, k), dtype=torch.int64, device=D.device)
    stream = torch.cuda.current_stream(D.device).cuda_stream
    check(lib.rag_merge_topk_device(nl, nq, k, C.c_void_p(D.data_ptr()), 0, C.c_void_p(I.data_ptr()), 0,
                                    C.c_void_p(Do.data_ptr()), C.c_void_p(Io.data_ptr()), dev,
                                    C.c_void_p(stream)))
    return Do, Io
