"""rag_amd: MI355X-native dense-retrieval hot path (exact / IVF top-k, score
fusion, MMR) behind the storage-manager / search-optimizer / MMR-diversifier
interfaces of llano1025/rAG.  All arithmetic runs in librag_hip.so (hand-written
HIP for gfx950); importing the package fails if that library is not built."""
from ._lib import RagHipError, device_count, lib  # noqa: F401
from .index import DeviceFilter, DeviceIndex, MultiDeviceIndex, merge_topk  # noqa: F401

__all__ = ["DeviceFilter", "DeviceIndex", "MultiDeviceIndex", "merge_topk", "RagHipError", "device_count"]
