"""ctypes binding of librag_hip.so (the C ABI declared in include/rag_hip.h).

There is no CPU fallback: if the shared library is missing or does not load,
importing this module raises, and every product entry point fails with it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAG_HIP_LIB points measurements at an alternative build of the same library (A/B runs)
LIB_PATH = os.environ.get("RAG_HIP_LIB") or os.path.join(_HERE, "librag_hip.so")


class RagHipError(RuntimeError):
    """A call into librag_hip.so returned a non-zero status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"librag_hip status {code}: {message}")
        self.code = code
        self.message = message


RAG_OK, RAG_ERR_INVALID, RAG_ERR_HIP, RAG_ERR_OOM, RAG_ERR_UNSUPPORTED = 0, 1, 2, 3, 4
METRIC_IP, METRIC_COSINE, METRIC_L2 = 0, 1, 2
KIND_FLAT, KIND_IVF = 0, 1
MMR_COSINE, MMR_DOT, MMR_EUCLIDEAN = 0, 1, 2

_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes); every symbol include/rag_hip.h declares
SIGNATURES = {
    "rag_version": (_i32, []),
    "rag_last_error": (C.c_char_p, []),
    "rag_device_count": (_i32, [C.POINTER(_i32)]),
    "rag_thread_release": (_i32, []),
    "rag_index_create": (_i32, [_i32, _i32, _i32, _i32, C.POINTER(_vp)]),
    "rag_index_free": (_i32, [_vp]),
    "rag_index_reserve": (_i32, [_vp, _i64]),
    "rag_index_add": (_i32, [_vp, _i64, _vp, _vp]),
    "rag_index_add_device": (_i32, [_vp, _i64, _vp, _vp]),
    "rag_index_adopt_device": (_i32, [_vp, _i64, _vp]),
    "rag_index_set_id_base": (_i32, [_vp, _i64]),
    "rag_index_set_prefilter": (_i32, [_vp, _i32]),
    "rag_index_set_stream_lifetime": (_i32, [_vp, _i32]),
    "rag_index_ntotal": (_i64, [_vp]),
    "rag_index_dim": (_i32, [_vp]),
    "rag_index_metric": (_i32, [_vp]),
    "rag_index_kind": (_i32, [_vp]),
    "rag_index_set_active": (_i32, [_vp, _i64, _vp, _i32]),
    "rag_index_reconstruct": (_i32, [_vp, _i64, _vp, _vp]),
    "rag_index_save": (_i32, [_vp, C.c_char_p]),
    "rag_index_load": (_i32, [C.c_char_p, _i32, C.POINTER(_vp)]),
    "rag_index_search": (_i32, [_vp, _i64, _vp, _i32, _f32, _i32, _vp, _vp, _vp]),
    "rag_index_search_device": (_i32, [_vp, _i64, _vp, _i32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "rag_index_search_segments": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _f32, _vp, _vp, _vp]),
    "rag_index_search_segments_fused": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i32, _f32, _vp, _i32, _vp, _vp,
                                               _vp, _vp]),
    "rag_filter_create": (_i32, [_i32, _i64, _vp, C.POINTER(_vp)]),
    "rag_filter_free": (_i32, [_vp]),
    "rag_filter_rows": (_i64, [_vp]),
    "rag_filter_device_bytes": (_vp, [_vp]),
    "rag_index_search_filtered": (_i32, [_vp, _i64, _vp, _i32, _f32, _i32, _vp, _vp, _vp]),
    "rag_index_search_segments_filtered": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _f32, _vp, _vp, _vp]),
    "rag_index_search_segments_fused_filtered": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i32, _f32, _vp, _i32, _vp, _vp,
                                                        _vp, _vp]),
    "rag_index_last_path": (C.c_char_p, [_vp]),
    "rag_merge_topk_device": (_i32, [_i32, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp]),
    "rag_merge_topk_device_static": (_i32, [_i32, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp]),
    "rag_index_profile_enable": (_i32, [_vp, _i32]),
    "rag_index_profile_read": (_i32, [_vp, C.POINTER(_f64), C.POINTER(_i64), _i32]),
    "rag_index_train_ivf": (_i32, [_vp, _i32, _vp, _i64, _vp]),
    "rag_fuse_weighted": (_i32, [_i64, _vp, _vp, _f64, _f64, _i32, _f64, _i32, _vp, _vp, _vp]),
    "rag_fuse_maxnorm": (_i32, [_i64, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "rag_mmr_select": (_i32, [_i64, _i32, _vp, _i32, _vp, _f64, _f64, _i32, _i32, _i32, _vp, _vp]),
    "rag_similarity_matrix": (_i32, [_i64, _i32, _vp, _i32, _i32, _i32, _vp]),
    "rag_mmr_select_rows": (_i32, [_vp, _i64, _vp, _vp, _f64, _f64, _i32, _i32, _vp, _vp]),
    "rag_index_hybrid_mmr": (_i32, [_vp, _vp, _i32, _f32, _i32, _vp, _i64, _vp, _vp, _f64, _f64, _i32, _f64, _f64, _i32,
                                    _i32, _vp]),
    "rag_index_hybrid_mmr_batch_device": (_i32, [_vp, _i64, _vp, _i32, _f32, _i32, _vp, _i64, _vp, _vp, _f64, _f64, _i32, _f64, _f64,
                                                  _i32, _i32, _vp, _vp, _vp, _vp]),
    "rag_index_hybrid_mmr_device": (_i32, [_vp, _vp, _i32, _f32, _i32, _vp, _i64, _vp, _vp, _f64, _f64, _i32, _f64, _f64,
                                           _i32, _i32, _vp, _vp]),
    "rag_multi_create": (_i32, [_i32, _i32, _i32, _i32, _vp, C.POINTER(_vp)]),
    "rag_multi_free": (_i32, [_vp]),
    "rag_multi_ntotal": (_i64, [_vp]),
    "rag_multi_ndevices": (_i32, [_vp]),
    "rag_multi_shard_rows": (_i64, [_vp, _i32]),
    "rag_multi_train_ivf": (_i32, [_vp, _i32, _vp, _i64, _vp]),
    "rag_multi_add": (_i32, [_vp, _i64, _vp, _vp]),
    "rag_multi_set_prefilter": (_i32, [_vp, _i32]),
    "rag_multi_set_active": (_i32, [_vp, _i64, _vp, _i32]),
    "rag_multi_reconstruct": (_i32, [_vp, _i64, _vp, _vp]),
    "rag_multi_search": (_i32, [_vp, _i64, _vp, _i32, _f32, _i32, _vp, _vp, _vp]),
    "rag_multi_filter_create": (_i32, [_vp, _vp, C.POINTER(_vp)]),
    "rag_multi_filter_free": (_i32, [_vp]),
    "rag_multi_search_filtered": (_i32, [_vp, _i64, _vp, _i32, _f32, _i32, _vp, _vp, _vp]),
    "rag_multi_last_merge": (C.c_char_p, [_vp]),
    "rag_bm25_create": (_i32, [_i32, C.POINTER(_vp)]),
    "rag_bm25_destroy": (None, [_vp]),
    "rag_bm25_ndocs": (_i64, [_vp]),
    "rag_bm25_add": (_i32, [_vp, _i64, _vp, _vp]),
    "rag_bm25_clear": (_i32, [_vp]),
    "rag_bm25_score": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _f64, _f64, _f64, _vp]),
    "rag_bm25_score_device": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _f64, _f64, _f64, _vp, _vp]),
    "rag_bm25_rank": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _f64, _f64, _f64, _i32, _f64, _i32, _vp, _vp, _vp, _vp]),
    "rag_bm25_rank_device": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _f64, _f64, _f64, _i32, _f64, _i32, _vp, _vp, _vp, _vp,
                                    _vp, _vp]),
}


class HybridOut(C.Structure):
    """rag_hybrid_out_t: where the search -> fusion -> MMR pipeline leaves its results."""
    _fields_ = [("ids", _vp), ("scores", _vp), ("picked", _vp), ("n_out", _vp), ("fused_rows", _vp),
                ("fused_scores", _vp), ("fused_sem", _vp), ("fused_bm", _vp), ("n_fused", _vp), ("status", _vp)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C rag_amd/csrc` (hipcc --offload-arch=gfx950). rag_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error() -> str:
    return (lib.rag_last_error() or b"").decode("utf-8", "replace")


def check(status: int) -> None:
    if status != RAG_OK:
        raise RagHipError(status, last_error())


def device_count() -> int:
    n = _i32(0)
    status = lib.rag_device_count(C.byref(n))
    if status != RAG_OK:
        return 0
    return n.value
