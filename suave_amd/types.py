"""Schema description of a mixed-type table.

Same semantics as the reference's ``suave/types.py:9-33,134-171,267-287``
(``ColumnSpec`` / ``Schema``: five column types, ``n_classes`` mandatory and
> 1 for ``cat`` / ``ordinal``, optional positive ``y_dim`` override, default
``y_dim`` = 1 for numeric types and ``n_classes`` for discrete ones, insertion
order = feature order), written for this package.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterator, Mapping, Optional, Tuple

SUPPORTED_TYPES = ("real", "pos", "count", "cat", "ordinal")
NUMERIC_TYPES = ("real", "pos", "count")
DISCRETE_TYPES = ("cat", "ordinal")


@dataclass(frozen=True)
class ColumnSpec:
    type: str
    n_classes: Optional[int] = None
    y_dim: Optional[int] = None


def _parse(name: str, raw: Mapping[str, object] | ColumnSpec) -> ColumnSpec:
    if isinstance(raw, ColumnSpec):
        raw = {"type": raw.type, "n_classes": raw.n_classes, "y_dim": raw.y_dim}
    spec = dict(raw)
    ctype = spec.get("type")
    if ctype not in SUPPORTED_TYPES:
        raise ValueError(
            f"Unsupported column type '{ctype}' for '{name}'. Supported types: real, pos, count, cat, ordinal."
        )
    n_classes = spec.get("n_classes")
    if ctype in DISCRETE_TYPES:
        if n_classes is None:
            raise ValueError(f"Column '{name}' of type '{ctype}' requires 'n_classes'.")
        n_classes = int(n_classes)
        if n_classes <= 1:
            raise ValueError(f"Column '{name}' of type '{ctype}' must declare 'n_classes' greater than 1.")
    elif n_classes is not None:
        raise ValueError(f"Column '{name}' of type '{ctype}' should not provide 'n_classes'.")
    y_dim = spec.get("y_dim")
    if y_dim is not None:
        y_dim = int(y_dim)
        if y_dim <= 0:
            raise ValueError(f"Column '{name}' must declare a positive 'y_dim' share")
    return ColumnSpec(type=ctype, n_classes=n_classes, y_dim=y_dim)


class Schema:
    """Ordered mapping ``column name -> ColumnSpec``."""

    def __init__(self, columns: Mapping[str, Mapping[str, object] | ColumnSpec]):
        self._columns: Dict[str, ColumnSpec] = {name: _parse(name, raw) for name, raw in columns.items()}

    # -- views ---------------------------------------------------------------
    @property
    def feature_names(self) -> Tuple[str, ...]:
        return tuple(self._columns)

    def _of_type(self, ctype: str) -> Tuple[str, ...]:
        return tuple(n for n, s in self._columns.items() if s.type == ctype)

    @property
    def real_features(self) -> Tuple[str, ...]:
        return self._of_type("real")

    @property
    def positive_features(self) -> Tuple[str, ...]:
        return self._of_type("pos")

    @property
    def count_features(self) -> Tuple[str, ...]:
        return self._of_type("count")

    @property
    def categorical_features(self) -> Tuple[str, ...]:
        return self._of_type("cat")

    @property
    def ordinal_features(self) -> Tuple[str, ...]:
        return self._of_type("ordinal")

    # -- mapping protocol ------------------------------------------------------
    def __contains__(self, item: str) -> bool:
        return item in self._columns

    def __getitem__(self, item: str) -> ColumnSpec:
        return self._columns[item]

    def __len__(self) -> int:
        return len(self._columns)

    def __iter__(self) -> Iterator[str]:
        return iter(self._columns)

    def to_dict(self) -> Dict[str, Dict[str, object]]:
        out: Dict[str, Dict[str, object]] = {}
        for name, spec in self._columns.items():
            entry: Dict[str, object] = {"type": spec.type}
            if spec.n_classes:
                entry["n_classes"] = spec.n_classes
            if spec.y_dim:
                entry["y_dim"] = spec.y_dim
            out[name] = entry
        return out

    def update(self, other: Mapping[str, Mapping[str, object]]) -> None:
        for name, raw in other.items():
            self._columns[name] = _parse(name, raw)

    def require_columns(self, columns) -> None:
        missing = [c for c in columns if c not in self._columns]
        if missing:
            raise KeyError(f"Columns {missing} missing from schema")

    def y_dimensions(self) -> Dict[str, int]:
        return {n: (s.y_dim if s.y_dim is not None else self._default_y_dim(s)) for n, s in self._columns.items()}

    @staticmethod
    def _default_y_dim(spec: ColumnSpec) -> int:
        if spec.type in NUMERIC_TYPES:
            return 1
        if spec.n_classes is None:
            raise ValueError("Categorical and ordinal columns require 'n_classes' to infer y_dim")
        return int(spec.n_classes)
