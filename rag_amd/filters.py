"""Payload filters of a search, resolved on the host into one byte per row.

The reference turns ``filters`` into a Qdrant ``Filter(must=[...])``
(vector_db/search_optimizer.py:173-201); here the same conditions are evaluated
against the payload table and handed to the kernels as a row mask.
"""
from __future__ import annotations

from typing import Any, Dict


def _match(value, wanted) -> bool:
    """Qdrant MatchValue: equality, or membership when the payload field is a list."""
    if isinstance(value, (list, tuple)):
        return wanted in value
    return value == wanted


def payload_passes(payload: Dict[str, Any], filters: Dict[str, Any]) -> bool:
    """vector_db/search_optimizer.py:173-201: every condition must hold
    (Filter(must=...)); a list value adds one equality condition per element;
    {'range': {...}} is a numeric range; anything else is equality."""
    for key, value in filters.items():
        pv = payload.get(key)
        if isinstance(value, list):
            if pv is None or not all(_match(pv, v) for v in value):
                return False
        elif isinstance(value, dict) and "range" in value:
            arg = value["range"]
            if not isinstance(pv, (int, float)) or isinstance(pv, bool):
                return False
            if arg.get("gte") is not None and not pv >= arg["gte"]:
                return False
            if arg.get("lte") is not None and not pv <= arg["lte"]:
                return False
            if arg.get("gt") is not None and not pv > arg["gt"]:
                return False
            if arg.get("lt") is not None and not pv < arg["lt"]:
                return False
        elif pv is None or not _match(pv, value):
            return False
    return True
