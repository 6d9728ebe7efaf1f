"""Data carriers that cross the drop-in boundary, with the field names of the
reference (api/schemas/search_schemas.py:36-126,182-266; vector_db/chunking.py:28-41;
vector_db/search_optimizer.py:33-66)."""
from __future__ import annotations

from dataclasses import dataclass, field
from datetime import datetime
from enum import Enum
from typing import Any, Dict, List, Optional, Union


class MetricType(Enum):
    COSINE = "cosine"
    EUCLIDEAN = "euclidean"
    DOT_PRODUCT = "dot"


class IndexType(Enum):
    HNSW = "hnsw"
    FLAT = "flat"
    IVF = "ivf"


@dataclass
class SearchConfig:
    """vector_db/search_optimizer.py:43-62.  ``hnsw_config`` is accepted and ignored: every
    search here is exact (or IVF when asked).  A ``quantization_config`` (the reference hands
    it to the collection, :130, where the server searches a compressed copy and re-scores with
    the originals) switches on the device-side equivalent: a bf16 copy of the rows searched
    first, the kept rows re-scored in fp32 -- answers unchanged."""
    dimension: int
    metric: MetricType = MetricType.COSINE
    collection_name: str = "default"
    hnsw_config: Optional[Dict[str, Any]] = None
    quantization_config: Optional[Dict[str, Any]] = None
    device: int = 0
    index_type: IndexType = IndexType.FLAT
    nlist: int = 0
    nprobe: int = 1


class SearchError(Exception):
    """Raised when a search operation fails (search_optimizer.py:64-66)."""


@dataclass
class Chunk:
    """vector_db/chunking.py:28-41."""
    text: str
    start_idx: int
    end_idx: int
    metadata: Dict
    context_text: Optional[str] = None
    embedding: Optional[Any] = None
    context_embedding: Optional[Any] = None
    neighbor_chunks: Optional[List[str]] = None
    document_id: Optional[int] = None
    chunk_id: Optional[str] = None


@dataclass
class SearchResult:
    """api/schemas/search_schemas.py:182-266 (a plain dataclass: same attribute
    names, ``to_dict`` / ``from_dict`` round trip)."""
    document_id: Union[str, int]
    chunk_id: str
    text: str
    score: float
    metadata: Dict = field(default_factory=dict)
    document_metadata: Dict = field(default_factory=dict)
    snippet: Optional[str] = None
    filename: Optional[str] = None
    highlight: Optional[str] = None
    timestamp: Optional[datetime] = None

    def to_dict(self) -> Dict:
        out = {"chunk_id": self.chunk_id, "document_id": self.document_id, "text": self.text,
               "score": self.score, "metadata": self.metadata, "document_metadata": self.document_metadata}
        if self.snippet is not None:
            out["snippet"] = self.snippet
        if self.highlight is not None:
            out["highlight"] = self.highlight
        if self.timestamp is not None:
            out["timestamp"] = self.timestamp.isoformat() if isinstance(self.timestamp, datetime) else self.timestamp
        return out

    @classmethod
    def from_dict(cls, data: Dict) -> "SearchResult":
        ts = data.get("timestamp")
        if isinstance(ts, str):
            try:
                ts = datetime.fromisoformat(ts.replace("Z", "+00:00"))
            except ValueError:
                ts = None
        return cls(document_id=data["document_id"], chunk_id=data["chunk_id"], text=data["text"],
                   snippet=data.get("snippet"), score=data["score"], metadata=data.get("metadata", {}),
                   document_metadata=data.get("document_metadata", {}), highlight=data.get("highlight"),
                   timestamp=ts)


@dataclass
class SearchFilters:
    """The fields of api/schemas/search_schemas.py:36-98 that reach the hot path."""
    user_id: Optional[int] = None
    document_ids: Optional[List[int]] = None
    min_score: Optional[float] = None
    metadata_filters: Optional[dict] = None
    enable_mmr: bool = False
    mmr_lambda: float = 0.6
    mmr_similarity_threshold: float = 0.8
    mmr_max_results: Optional[int] = None
    mmr_similarity_metric: str = "cosine"

    def __post_init__(self):
        if not 0.0 <= self.mmr_lambda <= 1.0:
            raise ValueError("mmr_lambda must be in [0, 1]")
        if not 0.0 <= self.mmr_similarity_threshold <= 1.0:
            raise ValueError("mmr_similarity_threshold must be in [0, 1]")
        if self.mmr_max_results is not None and not 1 <= self.mmr_max_results <= 100:
            raise ValueError("mmr_max_results must be in [1, 100]")
        if self.mmr_similarity_metric not in ("cosine", "euclidean", "dot_product"):
            raise ValueError("mmr_similarity_metric must be cosine, euclidean or dot_product")


MAX_DEVICE_LIMIT = 1024


def device_limit(limit: int) -> int:
    """The device keeps at most 1024 hits per query (one workgroup merges them); a larger
    ``limit`` is an error, not a silently shorter answer."""
    limit = int(limit)
    if limit > MAX_DEVICE_LIMIT:
        raise ValueError(f"limit {limit} exceeds the device top-k bound of {MAX_DEVICE_LIMIT}")
    return limit
