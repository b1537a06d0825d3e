"""Data layer: normalisation statistics, the packed-table writer (host reference path and the device path that
``SUAVE`` uses: ``pack_frame_device``) and the generation-side frame assembly (``FrameAssembler``).

Semantics of the reference's ``suave/data.py:133-266`` (``standardize``) and
``suave/model.py:2784-2922`` (``_prepare_training_tensors``) /
``:3116-3199`` (``_apply_training_normalization``), but emitting the packed
table (float32 numerics, int16 codes, uint8 masks) directly instead of dense
one-hot / thermometer tensors -- no per-row Python loop (the reference builds
the ordinal thermometer row by row, model.py:2890-2891).
"""

from __future__ import annotations

import warnings
import ctypes as C
from typing import Dict, List, Mapping, Optional, Tuple

import numpy as np
import pandas as pd

from .types import Schema

_EPS = 1e-6


def _fit_norm_stats_sharded(X: pd.DataFrame, schema: Schema) -> Dict[str, Dict[str, object]]:
    """``fit_norm_stats`` of a table whose rows are block-sharded over the ranks of torch.distributed: every rank passes
    its block and gets the statistics of the whole table -- two passes (mean, then squared deviations) like the numpy
    / pandas routines the one-process path calls, summed over ranks in float64; category lists are the sorted union."""

    from . import parallel

    schema.require_columns([c for c in X.columns if c in schema])
    cols = list(schema.feature_names)
    for column in cols:
        if column not in X:
            raise KeyError(f"column '{column}' of the schema is missing from the data")
    numeric: Dict[str, np.ndarray] = {}
    first = np.zeros((len(cols), 4), np.float64)  # count, sum, violation flag, any value <= 0
    for j, column in enumerate(cols):
        t = schema[column].type
        if t not in ("real", "pos", "count"):
            continue
        v = pd.to_numeric(X[column], errors="coerce").to_numpy(dtype=np.float64)
        if t == "pos":
            first[j, 2] = float(np.any(v[~np.isnan(v)] < -1.0 + _EPS))
            with np.errstate(invalid="ignore", divide="ignore"):
                v = np.log1p(v)
            v = np.where(np.isfinite(v), v, np.nan) if t == "pos" else v
        if t == "count":
            fin = v[~np.isnan(v)]
            first[j, 2] = float(np.any(fin < 0))
            first[j, 3] = float(np.any(fin <= 0))
        ok = ~np.isnan(v)
        first[j, 0], first[j, 1] = ok.sum(), v[ok].sum()
        numeric[column] = v
    tot = parallel.all_reduce_host(first[:, :2], "sum")
    flags = parallel.all_reduce_host(first[:, 2:], "max")
    second = np.zeros(len(cols), np.float64)
    means = np.zeros(len(cols), np.float64)
    for j, column in enumerate(cols):
        if column in numeric and tot[j, 0] > 0:
            means[j] = tot[j, 1] / tot[j, 0]
            d = numeric[column] - means[j]
            second[j] = np.nansum(d * d)
    ssq = parallel.all_reduce_host(second, "sum")
    cat_local = {c: X[c].astype("category").cat.categories.tolist() for c in cols if schema[c].type == "cat"}
    cat_all = parallel.all_gather_objects(cat_local)
    stats: Dict[str, Dict[str, object]] = {}
    for j, column in enumerate(cols):
        spec = schema[column]
        if spec.type in ("real", "pos"):
            if spec.type == "pos" and flags[j, 0] > 0:
                raise ValueError(f"Column '{column}' of type 'pos' must be >= -1 to apply log1p")
            n = tot[j, 0]
            mean = float(means[j]) if n > 0 else 0.0
            std = float(np.sqrt(ssq[j] / n)) if n > 0 else 1.0
            if not np.isfinite(std) or std < _EPS:
                std = 1.0
            if not np.isfinite(mean):
                mean = 0.0
            stats[column] = ({"type": "real", "mean": mean, "std": std} if spec.type == "real" else
                             {"type": "pos", "mean_log": mean, "std_log": std})
        elif spec.type == "count":
            if flags[j, 0] > 0:
                raise ValueError(f"Column '{column}' of type 'count' must be non-negative")
            stats[column] = {"type": "count", "offset": 1.0 if flags[j, 1] > 0 else 0.0}
        elif spec.type == "cat":
            union = set()
            for part in cat_all:
                union.update(part[column])
            stats[column] = {"type": "cat", "categories": sorted(union)}
        else:
            n_classes = int(spec.n_classes or 0)
            cats = X[column].cat.categories.tolist() if isinstance(X[column].dtype, pd.CategoricalDtype) else list(range(n_classes))
            stats[column] = {"type": "ordinal", "n_classes": n_classes, "categories": cats}
    return stats


def fit_norm_stats(X: pd.DataFrame, schema: Schema, *, sharded: bool = False) -> Dict[str, Dict[str, object]]:
    """Per-column statistics, as ``standardize`` computes them (data.py:182-264).  ``sharded``: X is this rank's block
    of a row-sharded table and the statistics are those of the whole table (collective: every rank calls it)."""

    if sharded:
        return _fit_norm_stats_sharded(X, schema)
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


# ---------------------------------------------------------------------------------------------------------------
# Device path: the raw columns go to the GPU as they are and libsuave_hip.so
# writes the packed table (suave_pack_columns) / hands generated columns back as DataFrame-ready blocks
# (suave_unpack_columns).  The host keeps what is pandas' own: label -> category code maps and the frame object.
# ---------------------------------------------------------------------------------------------------------------

_KIND = {"real": 0, "pos": 1, "count": 2, "cat": 3, "ordinal": 4}
_RAW = {np.dtype(np.float64): 0, np.dtype(np.float32): 1, np.dtype(np.int64): 2, np.dtype(np.int32): 3,
        np.dtype(np.int16): 4, np.dtype(np.int8): 5}
_PACK_COL = np.dtype([("src", np.uint64), ("src_dtype", np.int32), ("kind", np.int32), ("n_classes", np.int32),
                      ("slot", np.int32), ("a", np.float64), ("b", np.float64)], align=True)   # suave_pack_col_t
_UNPACK_COL = np.dtype([("block", np.int32), ("src", np.int32), ("dst", np.int32), ("reserved", np.int32)])


def _raw_array(series: pd.Series) -> Tuple[np.ndarray, int]:
    """The column's values in a dtype the packer reads, without a copy when the frame already holds one; the second
    value counts entries that are present but not numeric (coerced to NaN, as ``pd.to_numeric(errors="coerce")``)."""

    dt = series.dtype
    if isinstance(dt, np.dtype) and dt in _RAW:
        return np.ascontiguousarray(series.to_numpy()), 0
    if isinstance(dt, np.dtype) and dt.kind in "iub":
        return np.ascontiguousarray(series.to_numpy(dtype=np.int64)), 0
    if isinstance(dt, np.dtype) and dt.kind == "f":
        return np.ascontiguousarray(series.to_numpy(dtype=np.float64)), 0
    v = pd.to_numeric(series, errors="coerce").to_numpy(dtype=np.float64, na_value=np.nan)
    return np.ascontiguousarray(v), int((series.notna().to_numpy() & np.isnan(v)).sum())


def pack_frame_device(X: pd.DataFrame, schema: Schema, stats: Mapping[str, Mapping[str, object]], engine,
                      chunk_rows: int = 1 << 22, info: Optional[Dict[str, int]] = None):
    """``pack_frame`` on the GPU: returns ``(xnum, codes, mask)`` as device tensors, same values (the float64
    normalisation arithmetic runs in the kernel; log1p / log may differ from numpy's in the last float32 bit).
    ``info`` (optional dict) receives ``beyond_fp16``: the number of real cells whose normalised value exceeds 60 000
    in magnitude (the packer's counter of real columns) -- such a table needs ``suave_gemm_arithmetic_set(3)``."""

    import torch

    names = list(schema.feature_names)
    n = len(X)
    n_num = sum(schema[c].type in ("real", "pos", "count") for c in names)
    n_disc = len(names) - n_num
    dev = engine.device
    xnum = torch.empty((n, max(n_num, 1)), dtype=torch.float32, device=dev)
    codes = torch.empty((n, max(n_disc, 1)), dtype=torch.int16, device=dev)
    mask = torch.empty((n, len(names)), dtype=torch.uint8, device=dev)
    rejected = torch.zeros(len(names), dtype=torch.int64, device=dev)
    non_numeric = np.zeros(len(names), np.int64)
    for start in range(0, n, chunk_rows):
        stop = min(n, start + chunk_rows)
        desc = np.zeros(len(names), _PACK_COL)
        keep = []
        i_num = i_disc = 0
        for j, column in enumerate(names):
            spec = schema[column]
            st = stats.get(column, {})
            series = X[column].iloc[start:stop]
            d = desc[j]
            d["kind"] = _KIND[spec.type]
            if spec.type == "cat":
                cats = st.get("categories")
                cat = pd.Categorical(series, categories=cats) if cats is not None else series.astype("category").array
                arr = np.ascontiguousarray(cat.codes)
                d["n_classes"] = int(spec.n_classes)
            else:
                arr, bad = _raw_array(series)
                if spec.type == "ordinal":
                    non_numeric[j] += bad
                    d["n_classes"] = int(st.get("n_classes", spec.n_classes or 0))
                elif spec.type == "real":
                    std = float(st.get("std", 1.0))
                    d["a"], d["b"] = float(st.get("mean", 0.0)), (std if std > 0 else 1.0)
                elif spec.type == "pos":
                    d["a"], d["b"] = float(st.get("mean_log", 0.0)), float(st.get("std_log", 1.0))
                else:
                    d["a"] = float(st.get("offset", 0.0))
            if spec.type in ("real", "pos", "count"):
                d["slot"] = i_num
                i_num += 1
            else:
                d["slot"] = i_disc
                i_disc += 1
            with warnings.catch_warnings():  # (frames hand out read-only views; the packer only reads)
                warnings.filterwarnings("ignore", message="The given NumPy array is not writable")
                t = torch.from_numpy(arr).to(dev, non_blocking=True)
            keep.append(t)
            d["src"] = t.data_ptr()
            d["src_dtype"] = _RAW[arr.dtype]
        desc_t = torch.from_numpy(desc.view(np.uint8)).to(dev)
        engine.pack_columns(desc_t, stop - start, xnum[start:], codes[start:], mask[start:], rejected)
        torch.cuda.current_stream(dev).synchronize()  # the raw columns may go now
        del keep
    rej = rejected.cpu().numpy()
    if info is not None:
        info["beyond_fp16"] = int(sum(int(rej[j]) for j, column in enumerate(names) if schema[column].type == "real"))
    for j, column in enumerate(names):
        t = schema[column].type
        if t == "pos" and rej[j]:
            raise ValueError(f"Column '{column}' of type 'pos' must be >= -1 to apply log1p")
        if t == "count" and rej[j]:
            raise ValueError(f"Column '{column}' of type 'count' must be non-negative")
        if t == "ordinal" and (rej[j] or non_numeric[j]):
            n_classes = int(stats.get(column, {}).get("n_classes", schema[column].n_classes or 0))
            warnings.warn(
                f"Column '{column}' contains ordinal values outside [0, {n_classes - 1}] or non-numeric entries; "
                "they will be treated as missing.", UserWarning, stacklevel=2)
    return xnum[:, :n_num], codes[:, :n_disc], mask


class FrameAssembler:
    """Generated rows -> DataFrame without per-column device copies, dtype conversions or ``pd.concat``: every chunk's
    row-major decode outputs are regrouped on the device into columnar blocks (float32 | int64 counts | int16 / int8
    codes: ``suave_unpack_columns``) and copied straight into their place in host blocks sized for the whole request
    (pinned while they are small enough); the frame wraps rows of those blocks.  Values and dtypes are those of
    ``frame_from_packed``."""

    _PIN_LIMIT = 2 << 30

    def __init__(self, schema: Schema, stats: Mapping[str, Mapping[str, object]], engine, n_rows: int, chunk_rows: int):
        import torch

        self.schema, self.stats, self.engine, self.n = schema, stats, engine, int(n_rows)
        self.plan: List[Tuple[str, int, int]] = []  # (column, block, row in block)
        counts = [0, 0, 0, 0]
        desc = []
        i_num = i_disc = 0
        for column in schema.feature_names:
            spec = schema[column]
            if spec.type in ("real", "pos", "count"):
                blk, src = (1 if spec.type == "count" else 0), i_num
                i_num += 1
            else:
                blk, src = (3 if int(spec.n_classes or 0) <= 127 else 2), i_disc
                i_disc += 1
            self.plan.append((column, blk, counts[blk]))
            desc.append((blk, src, counts[blk], 0))
            counts[blk] += 1
        dts = (torch.float32, torch.int64, torch.int16, torch.int8)
        total = sum(c * self.n * torch.empty((), dtype=d).element_size() for c, d in zip(counts, dts))
        pin = total <= self._PIN_LIMIT
        self.host = [torch.empty((c, self.n), dtype=d, pin_memory=pin and c > 0) for c, d in zip(counts, dts)]
        self.dev = [torch.empty((max(c, 1), chunk_rows), dtype=d, device=engine.device) for c, d in zip(counts, dts)]
        self.desc = torch.from_numpy(np.array(desc, dtype=_UNPACK_COL).view(np.uint8)).to(engine.device)
        self.n_out = len(desc)
        self.chunk_rows = int(chunk_rows)

    def add(self, start: int, out_num, out_codes) -> None:
        """Rows [start, start + nb) of the request: the decode outputs of one chunk (device tensors)."""

        nb = int(out_num.shape[0]) if out_num.numel() else int(out_codes.shape[0])
        self.engine.unpack_columns(self.desc, self.n_out, nb, out_num, out_codes, self.dev, self.chunk_rows)
        for h, d in zip(self.host, self.dev):
            if not h.shape[0]:
                continue
            if nb == self.n and nb == self.chunk_rows:  # one chunk: both sides are whole, contiguous blocks
                h.copy_(d[: h.shape[0]], non_blocking=True)
                continue
            # a chunk of a larger request: [columns, nb] is a strided window of both blocks, which torch would stage
            # through a pageable temporary; one column's rows are contiguous on both sides -- a plain async DMA each
            for r in range(h.shape[0]):
                h[r, start:start + nb].copy_(d[r, :nb], non_blocking=True)

    def frame(self) -> pd.DataFrame:
        import torch

        torch.cuda.current_stream(self.engine.device).synchronize()
        blocks = [h.numpy() for h in self.host]
        data: Dict[str, object] = {}
        for column, blk, row in self.plan:
            spec = self.schema[column]
            if blk <= 1:
                data[column] = blocks[blk][row]
                continue
            cats = self.stats.get(column, {}).get("categories")
            n_classes = int(spec.n_classes or 0)
            cat_list = list(cats) if cats is not None else list(range(n_classes))
            if len(cat_list) < n_classes:
                warnings.warn(
                    "Column '%s' produced %d unseen categorical levels during sampling; extending the category list "
                    "with numeric placeholders." % (column, n_classes - len(cat_list)), RuntimeWarning, stacklevel=2)
                cat_list.extend(range(len(cat_list), n_classes))
            data[column] = pd.Categorical.from_codes(blocks[blk][row], categories=cat_list,
                                                     ordered=(spec.type == "ordinal"))
        return pd.DataFrame(data, copy=False)
