"""Host data layer: normalisation statistics and the packed-table writer.

Semantics of the reference's ``suave/data.py:133-266`` (``standardize``) and
``suave/model.py:2784-2922`` (``_prepare_training_tensors``) /
``:3116-3199`` (``_apply_training_normalization``), but emitting the packed
table (float32 numerics, int16 codes, uint8 masks) directly instead of dense
one-hot / thermometer tensors -- no per-row Python loop (the reference builds
the ordinal thermometer row by row, model.py:2890-2891).
"""

from __future__ import annotations

import warnings
from typing import Dict, Mapping, Tuple

import numpy as np
import pandas as pd

from .types import Schema

_EPS = 1e-6


def fit_norm_stats(X: pd.DataFrame, schema: Schema) -> Dict[str, Dict[str, object]]:
    """Per-column statistics, as ``standardize`` computes them (data.py:182-264)."""

    schema.require_columns([c for c in X.columns if c in schema])
    stats: Dict[str, Dict[str, object]] = {}
    for column in schema.feature_names:
        spec = schema[column]
        if column not in X:
            raise KeyError(f"column '{column}' of the schema is missing from the data")
        if spec.type == "real":
            v = pd.to_numeric(X[column], errors="coerce")
            mean = float(v.mean()) if len(v) else 0.0
            std = float(v.std(ddof=0)) if len(v) else 1.0
            if not np.isfinite(std) or std < _EPS:
                std = 1.0
            if not np.isfinite(mean):
                mean = 0.0
            stats[column] = {"type": "real", "mean": mean, "std": std}
        elif spec.type == "pos":
            v = pd.to_numeric(X[column], errors="coerce")
            fin = v[v.notna()]
            if not fin.empty and (fin < -1.0 + _EPS).any():
                raise ValueError(f"Column '{column}' of type 'pos' must be >= -1 to apply log1p")
            lv = np.log1p(v.to_numpy(dtype=np.float64))
            mean = float(np.nanmean(lv)) if np.isfinite(lv).any() else 0.0
            std = float(np.nanstd(lv)) if np.isfinite(lv).any() else 1.0
            if not np.isfinite(std) or std < _EPS:
                std = 1.0
            stats[column] = {"type": "pos", "mean_log": mean, "std_log": std}
        elif spec.type == "count":
            v = pd.to_numeric(X[column], errors="coerce")
            fin = v[v.notna()]
            if not fin.empty and (fin < 0).any():
                raise ValueError(f"Column '{column}' of type 'count' must be non-negative")
            offset = 1.0 if (not fin.empty and (fin <= 0).any()) else 0.0
            stats[column] = {"type": "count", "offset": offset}
        elif spec.type == "cat":
            cats = X[column].astype("category").cat.categories.tolist()
            stats[column] = {"type": "cat", "categories": cats}
        else:
            n_classes = int(spec.n_classes or 0)
            if isinstance(X[column].dtype, pd.CategoricalDtype):
                cats = X[column].cat.categories.tolist()
            else:
                cats = list(range(n_classes))
            stats[column] = {"type": "ordinal", "n_classes": n_classes, "categories": cats}
    return stats


def pack_frame(X: pd.DataFrame, schema: Schema, stats: Mapping[str, Mapping[str, object]]
               ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Normalise ``X`` with stored statistics and write the packed table.

    Returns ``(xnum float32 [N, n_num], codes int16 [N, n_disc], mask uint8
    [N, n_cols])``.  Masked cells are zero-filled / coded ``-1``.
    """

    names = list(schema.feature_names)
    n = len(X)
    n_num = sum(schema[c].type in ("real", "pos", "count") for c in names)
    xnum = np.zeros((n, n_num), np.float32)
    codes = np.full((n, len(names) - n_num), -1, np.int16)
    mask = np.zeros((n, len(names)), np.uint8)
    i_num = i_disc = 0
    for j, column in enumerate(names):
        spec = schema[column]
        st = stats.get(column, {})
        if spec.type in ("real", "pos", "count"):
            v = pd.to_numeric(X[column], errors="coerce").to_numpy(dtype=np.float64)
            if spec.type == "real":
                std = float(st.get("std", 1.0))
                if std <= 0:
                    std = 1.0
                v = (v - float(st.get("mean", 0.0))) / max(std, _EPS)
            elif spec.type == "pos":
                fin = v[~np.isnan(v)]
                if fin.size and (fin < -1.0 + _EPS).any():
                    raise ValueError(f"Column '{column}' of type 'pos' must be >= -1 to apply log1p")
                v = (np.log1p(v) - float(st.get("mean_log", 0.0))) / max(float(st.get("std_log", 1.0)), _EPS)
            else:
                fin = v[~np.isnan(v)]
                if fin.size and (fin < 0).any():
                    raise ValueError(f"Column '{column}' of type 'count' must be non-negative")
                with np.errstate(divide="ignore", invalid="ignore"):
                    v = np.log(v + float(st.get("offset", 0.0)))
            obs = np.isfinite(v)
            xnum[:, i_num] = np.where(obs, v, 0.0).astype(np.float32)
            mask[:, j] = obs
            i_num += 1
        elif spec.type == "cat":
            cats = st.get("categories")
            series = pd.Categorical(X[column], categories=cats) if cats is not None else X[column].astype("category")
            c = np.asarray(series.codes if isinstance(series, pd.Categorical) else series.cat.codes, dtype=np.int64)
            n_classes = int(spec.n_classes)
            c = np.where((c >= 0) & (c < n_classes), c, -1)
            codes[:, i_disc] = c.astype(np.int16)
            mask[:, j] = c >= 0
            i_disc += 1
        else:
            n_classes = int(st.get("n_classes", spec.n_classes or 0))
            orig = X[column]
            v = pd.to_numeric(orig, errors="coerce").to_numpy(dtype=np.float64)
            bad = (~np.isnan(v)) & ((v < 0) | (v > n_classes - 1))
            invalid = orig.notna().to_numpy() & np.isnan(v)
            if bad.any() or invalid.any():
                warnings.warn(
                    f"Column '{column}' contains ordinal values outside [0, {n_classes - 1}] or non-numeric entries; "
                    "they will be treated as missing.", UserWarning, stacklevel=2)
                v = np.where(bad, np.nan, v)
            obs = ~np.isnan(v)
            lvl = np.where(obs, np.clip(np.nan_to_num(v, nan=0.0).astype(np.int64), 0, n_classes - 1), -1)
            codes[:, i_disc] = lvl.astype(np.int16)
            mask[:, j] = obs
            i_disc += 1
    return xnum, codes, mask


def frame_from_packed(out_num: np.ndarray, out_codes: np.ndarray, schema: Schema,
                      stats: Mapping[str, Mapping[str, object]]) -> pd.DataFrame:
    """``sampling.decoder_outputs_to_frame`` (sampling.py:138-179) on columnar outputs."""

    data: Dict[str, object] = {}
    i_num = i_disc = 0
    for column in schema.feature_names:
        spec = schema[column]
        if spec.type in ("real", "pos"):
            data[column] = out_num[:, i_num].astype(np.float32, copy=False)
            i_num += 1
        elif spec.type == "count":
            data[column] = np.maximum(np.rint(out_num[:, i_num]), 0).astype(np.int64)
            i_num += 1
        else:
            c = out_codes[:, i_disc].astype(np.int64)
            i_disc += 1
            cats = stats.get(column, {}).get("categories")
            n_classes = int(spec.n_classes or 0)
            cat_list = list(cats) if cats is not None else list(range(n_classes))
            if len(cat_list) < n_classes:
                warnings.warn(
                    "Column '%s' produced %d unseen categorical levels during sampling; extending the category list "
                    "with numeric placeholders." % (column, n_classes - len(cat_list)), RuntimeWarning, stacklevel=2)
                cat_list.extend(range(len(cat_list), n_classes))
            data[column] = pd.Categorical.from_codes(c, categories=cat_list, ordered=(spec.type == "ordinal"))
    return pd.DataFrame(data)
