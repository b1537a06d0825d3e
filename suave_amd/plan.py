"""Schema plan: flat parameter layout, column table and ctypes mirrors of
``include/suave_hip.h``.

The flat float32 parameter buffer is a bijection of the reference's
``state_dict`` tensors: ``encoder.backbone.<i>.weight`` ...
``decoder.heads.<col>.mean_layer.weight`` plus the prior tensors.  Weight
matrices are stored ``[n_out, ld]`` row-major with ``ld`` = input width rounded
up to 4 floats (pad entries stay zero: their gradient is identically zero and
Adam leaves them untouched); the encoder's three output linears
(``component_logits``, ``mu_layer``, ``logvar_layer``, encoder.py:56-58) are
stacked into one ``[K + 2KD, H]`` matrix so the head is a single GEMM.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

from .types import DISCRETE_TYPES, NUMERIC_TYPES, Schema

MAX_HIDDEN = 8
MAX_COMPONENTS = 32
TYPE_CODE = {"real": 0, "pos": 1, "count": 2, "cat": 3, "ordinal": 4}
TYPE_ORDER = ("real", "pos", "count", "cat", "ordinal")  # encoder-input order, model.py:2813-2900
WIDE_CAT_MIN_CLASSES = 17  # SUAVE_WIDE_CAT_MIN_CLASSES in include/suave_hip.h


class ColT(C.Structure):
    _fields_ = [("type", C.c_int32), ("n_classes", C.c_int32), ("y_off", C.c_int32), ("y_dim", C.c_int32),
                ("x_index", C.c_int32), ("enc_off", C.c_int32), ("enc_mask", C.c_int32), ("w_off", C.c_int32),
                ("w2_off", C.c_int32), ("scale", C.c_float), ("shift", C.c_float), ("enc_embed", C.c_int32)]


class LayerT(C.Structure):
    _fields_ = [("w_off", C.c_int32), ("b_off", C.c_int32), ("n_out", C.c_int32), ("n_in", C.c_int32),
                ("ld", C.c_int32)]


class ModelT(C.Structure):
    _fields_ = [("n_cols", C.c_int32), ("n_num", C.c_int32), ("n_disc", C.c_int32), ("n_components", C.c_int32),
                ("latent_dim", C.c_int32), ("n_hidden", C.c_int32), ("enc_in", C.c_int32), ("total_y", C.c_int32),
                ("behaviour", C.c_int32), ("n_params", C.c_int32), ("dropout", C.c_float),
                ("enc", LayerT * MAX_HIDDEN), ("enc_head", LayerT), ("dec", LayerT * MAX_HIDDEN), ("y_proj", LayerT),
                ("head_off", C.c_int32), ("head_len", C.c_int32), ("prior_logits_off", C.c_int32),
                ("prior_mu_off", C.c_int32), ("prior_logvar_off", C.c_int32), ("pm_w_off", C.c_int32),
                ("pm_b_off", C.c_int32), ("y_rows", C.c_int32)]


class BatchT(C.Structure):
    _fields_ = [("xnum", C.c_void_p), ("codes", C.c_void_p), ("mask", C.c_void_p), ("row_index", C.c_void_p),
                ("row_begin", C.c_int64), ("n_rows", C.c_int32), ("ld_num", C.c_int32), ("ld_disc", C.c_int32),
                ("ld_mask", C.c_int32)]


class NoiseT(C.Structure):
    _fields_ = [("exp_noise", C.c_void_p), ("eps", C.c_void_p), ("enc_keep", C.c_void_p * MAX_HIDDEN),
                ("dec_keep", C.c_void_p * MAX_HIDDEN), ("seed", C.c_uint64), ("step", C.c_uint64)]


class ElboOutT(C.Structure):
    _fields_ = [("reconstruction", C.c_void_p), ("categorical_kl", C.c_void_p), ("gaussian_kl", C.c_void_p),
                ("latent", C.c_void_p), ("component", C.c_void_p), ("enc_out", C.c_void_p), ("sums", C.c_void_p),
                ("log_px", C.c_void_p), ("log_px_missing", C.c_void_p)]


class HeadT(C.Structure):  # suave_head_t
    _fields_ = [("n_classes", C.c_int32), ("training", C.c_int32), ("params", C.c_void_p), ("grads", C.c_void_p),
                ("class_weight", C.c_void_p), ("targets", C.c_void_p), ("weight_sum", C.c_void_p),
                ("loss_scale", C.c_float), ("loss_rows", C.c_float), ("loss_sum", C.c_void_p), ("dropout", C.c_float),
                ("dropout_key", C.c_uint64), ("adam_lr", C.c_float), ("adam_step", C.c_int64), ("adam_m", C.c_void_p),
                ("adam_v", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


def _r4(x: int) -> int:
    return (x + 3) // 4 * 4


@dataclass
class TensorSlot:
    """Where one ``state_dict`` tensor lives in the flat buffer."""

    name: str
    offset: int
    shape: Tuple[int, ...]
    ld: int  # row stride in the flat buffer (== shape[-1] for unpadded tensors)
    row0: int = 0  # first row inside a stacked matrix
    row_index: Optional[np.ndarray] = None  # tensor row i lives at buffer row row_index[i] (y_projection, wide schemas)
    # first encoder layer of wide schemas: buffer column i of the dense matrix holds tensor column col_index[i];
    # the remaining tensor columns (one-hot blocks of the wide categoricals) live transposed in embedding blocks
    # [(tensor column start, n_classes, buffer offset, row stride)]
    col_index: Optional[np.ndarray] = None
    embed: Optional[List[Tuple[int, int, int, int]]] = None


@dataclass
class ModelLayout:
    schema: Schema
    n_components: int
    latent_dim: int
    hidden: Tuple[int, ...]
    dropout: float
    behaviour: str
    norm_stats: Mapping[str, Mapping[str, object]]
    slots: Dict[str, TensorSlot] = field(default_factory=dict)
    cols: List[ColT] = field(default_factory=list)
    model: Optional[ModelT] = None

    # derived
    enc_in: int = 0      # width of the dense encoder input the first GEMM multiplies
    enc_in_ref: int = 0  # I of the reference (model.py:2905): enc_in + the one-hot widths of the embedded columns
    total_y: int = 0
    y_rows: int = 0  # rows of the internal y buffer / y_projection matrix (== total_y without wide categoricals)
    y_index: Optional[np.ndarray] = None  # reference y_full column -> internal y column
    n_num: int = 0
    n_disc: int = 0
    n_params: int = 0
    numeric_columns: Tuple[str, ...] = ()
    discrete_columns: Tuple[str, ...] = ()
    total_classes: int = 0
    class_offsets: Dict[str, int] = field(default_factory=dict)
    group_ranges: Dict[str, Tuple[int, int]] = field(default_factory=dict)

    def __post_init__(self) -> None:
        self._build()

    # ------------------------------------------------------------------
    def _build(self) -> None:
        schema, K, D = self.schema, int(self.n_components), int(self.latent_dim)
        if not (1 <= K <= MAX_COMPONENTS):
            raise ValueError(f"n_components must lie in 1..{MAX_COMPONENTS}")
        if len(self.hidden) > MAX_HIDDEN:
            raise ValueError(f"at most {MAX_HIDDEN} hidden layers are supported")
        names = list(schema.feature_names)
        ydims = schema.y_dimensions()
        self.numeric_columns = tuple(n for n in names if schema[n].type in NUMERIC_TYPES)
        self.discrete_columns = tuple(n for n in names if schema[n].type in DISCRETE_TYPES)
        self.n_num, self.n_disc = len(self.numeric_columns), len(self.discrete_columns)
        # encoder-input layout: values by type, then masks by type (model.py:2813-2905).  A wide categorical column
        # (C >= WIDE_CAT_MIN_CLASSES) contributes W1[:, its one-hot block] @ onehot = ONE column of W1 per row, so
        # its block is not part of the dense input at all: the C weight columns are kept transposed as an embedding
        # block [C, H1] that the first layer gathers from (forward) and scatters into (weight gradient).
        def is_wide(spec) -> bool:
            return spec.type == "cat" and int(spec.n_classes) >= WIDE_CAT_MIN_CLASSES

        embed_cols = [n for n in names if is_wide(schema[n]) and len(self.hidden) > 0 and int(self.hidden[0]) <= 1024]
        by_type = [n for t in TYPE_ORDER for n in names if schema[n].type == t]
        enc_off: Dict[str, int] = {}
        ref_off: Dict[str, int] = {}
        dense_ref_cols: List[int] = []
        pos = ref_pos = 0
        for n in by_type:
            width = 1 if schema[n].type in NUMERIC_TYPES else int(schema[n].n_classes)
            ref_off[n] = ref_pos
            if n in embed_cols:
                enc_off[n] = -1
            else:
                enc_off[n] = pos
                dense_ref_cols.extend(range(ref_pos, ref_pos + width))
                pos += width
            ref_pos += width
        enc_mask = {n: pos + i for i, n in enumerate(by_type)}
        dense_ref_cols.extend(range(ref_pos, ref_pos + len(by_type)))
        self.enc_in = pos + len(by_type)
        self.enc_in_ref = ref_pos + len(by_type)
        self.total_y = int(sum(ydims.values()))
        step = 3 if self.dropout > 0 else 2  # index step of nn.Sequential(Linear, ReLU[, Dropout])

        off = 0
        m = ModelT()

        def add_layer(layer: LayerT, prefix_w: str, prefix_b: str, n_out: int, n_in: int) -> None:
            nonlocal off
            ld = _r4(n_in)
            layer.w_off, layer.n_out, layer.n_in, layer.ld = off, n_out, n_in, ld
            self.slots[prefix_w] = TensorSlot(prefix_w, off, (n_out, n_in), ld)
            off += _r4(n_out * ld)
            layer.b_off = off
            self.slots[prefix_b] = TensorSlot(prefix_b, off, (n_out,), n_out)
            off += _r4(n_out)

        g0 = off
        prev = self.enc_in
        enc_embed: Dict[str, int] = {}
        for i, h in enumerate(self.hidden):
            add_layer(m.enc[i], f"encoder.backbone.{i * step}.weight", f"encoder.backbone.{i * step}.bias", h, prev)
            if i == 0 and embed_cols:
                sl = self.slots["encoder.backbone.0.weight"]
                sl.shape = (h, self.enc_in_ref)
                sl.col_index = np.asarray(dense_ref_cols, np.int64)
                sl.embed = []
                for n in embed_cols:
                    Cn = int(schema[n].n_classes)
                    enc_embed[n] = off
                    sl.embed.append((ref_off[n], Cn, off, _r4(h)))
                    off += Cn * _r4(h)
            prev = h
        # stacked encoder head
        E = K + 2 * K * D
        ld = _r4(prev)
        m.enc_head.w_off, m.enc_head.n_out, m.enc_head.n_in, m.enc_head.ld = off, E, prev, ld
        for nm, r0, rows in (("component_logits", 0, K), ("mu_layer", K, K * D), ("logvar_layer", K + K * D, K * D)):
            self.slots[f"encoder.{nm}.weight"] = TensorSlot(f"encoder.{nm}.weight", off + r0 * ld, (rows, prev), ld, r0)
        off += _r4(E * ld)
        m.enc_head.b_off = off
        for nm, r0, rows in (("component_logits", 0, K), ("mu_layer", K, K * D), ("logvar_layer", K + K * D, K * D)):
            self.slots[f"encoder.{nm}.bias"] = TensorSlot(f"encoder.{nm}.bias", off + r0, (rows,), rows, r0)
        off += _r4(E)
        self.group_ranges["encoder"] = (g0, off)
        g0 = off
        prev = D
        for i, h in enumerate(self.hidden):
            add_layer(m.dec[i], f"decoder.backbone.{i * step}.weight", f"decoder.backbone.{i * step}.bias", h, prev)
            prev = h
        # internal y layout: the slices of the narrow columns are contiguous in schema order (as in the reference,
        # decoder.py:306-314); every wide categorical column (C >= WIDE_CAT_MIN_CLASSES) then gets a 16-byte aligned
        # slot [y (C) | assignment s (K) | pad] behind them, so that its logits_layer is one aligned MFMA GEMM on
        # the reference's own feature vector [y_col ; s] (decoder.py:201-206)
        def is_wide(spec) -> bool:
            return spec.type == "cat" and int(spec.n_classes) >= WIDE_CAT_MIN_CLASSES

        y_start: Dict[str, int] = {}
        pos_y = 0
        for n in names:
            if not is_wide(schema[n]):
                y_start[n] = pos_y
                pos_y += int(ydims[n])
        pos_y = _r4(pos_y)
        for n in names:
            if is_wide(schema[n]):
                y_start[n] = pos_y
                pos_y += _r4(int(ydims[n]) + K)
        self.y_rows = max(pos_y, 1) if any(is_wide(schema[n]) for n in names) else self.total_y
        self.y_index = np.concatenate([np.arange(y_start[n], y_start[n] + int(ydims[n])) for n in names]).astype(np.int64)
        add_layer(m.y_proj, "decoder.y_projection.weight", "decoder.y_projection.bias", self.y_rows, prev)
        if self.y_rows != self.total_y:
            for key in ("decoder.y_projection.weight", "decoder.y_projection.bias"):
                sl = self.slots[key]
                sl.shape = (self.total_y,) + tuple(sl.shape[1:])
                sl.row_index = self.y_index
        # likelihood heads (all bias-free, decoder.py:61-62,105-106,153,190-192,233-236): narrow columns contiguous in
        # schema order, then the wide categorical weight matrices [C-1, ld] with ld = (C + K) rounded up to 4 floats
        m.head_off = off
        i_num = i_disc = 0
        cls_off = 0
        wide_cols = []
        for n in names:
            spec = schema[n]
            yd = int(ydims[n])
            col = ColT()
            col.type = TYPE_CODE[spec.type]
            col.n_classes = int(spec.n_classes or 0)
            col.y_off, col.y_dim = y_start[n], yd
            col.enc_off, col.enc_mask = enc_off[n], enc_mask[n]
            col.enc_embed = enc_embed.get(n, -1)
            col.w_off, col.w2_off = off, -1
            col.scale, col.shift = 1.0, 0.0
            stats = self.norm_stats.get(n, {}) if self.norm_stats else {}
            p = f"decoder.heads.{n}."
            if spec.type in ("real", "pos"):
                if spec.type == "real":
                    col.scale, col.shift = float(stats.get("std", 1.0)), float(stats.get("mean", 0.0))
                else:
                    col.scale, col.shift = float(stats.get("std_log", 1.0)), float(stats.get("mean_log", 0.0))
                self.slots[p + "mean_layer.weight"] = TensorSlot(p + "mean_layer.weight", off, (1, yd + K), yd + K)
                off += yd + K
                col.w2_off = off
                self.slots[p + "var_layer.weight"] = TensorSlot(p + "var_layer.weight", off, (1, K), K)
                off += K
            elif spec.type == "count":
                col.shift = float(stats.get("offset", 0.0))
                self.slots[p + "rate_layer.weight"] = TensorSlot(p + "rate_layer.weight", off, (1, yd + K), yd + K)
                off += yd + K
            elif spec.type == "cat":
                Cn = int(spec.n_classes)
                if is_wide(spec):
                    wide_cols.append((n, col))
                else:
                    self.slots[p + "logits_layer.weight"] = TensorSlot(p + "logits_layer.weight", off, (Cn - 1, yd + K), yd + K)
                    off += (Cn - 1) * (yd + K)
            else:
                Cn = int(spec.n_classes)
                self.slots[p + "partition_layer.weight"] = TensorSlot(p + "partition_layer.weight", off, (Cn - 1, K), K)
                off += (Cn - 1) * K
                col.w2_off = off
                self.slots[p + "mean_layer.weight"] = TensorSlot(p + "mean_layer.weight", off, (1, yd + K), yd + K)
                off += yd + K
            if spec.type in NUMERIC_TYPES:
                col.x_index = i_num
                i_num += 1
            else:
                col.x_index = i_disc
                i_disc += 1
                self.class_offsets[n] = cls_off
                cls_off += int(spec.n_classes)
            self.cols.append(col)
        m.head_len = off - m.head_off
        off = _r4(off)
        for n, col in wide_cols:
            Cn, yd = int(schema[n].n_classes), int(ydims[n])
            ldw = _r4(yd + K)
            col.w_off = off
            key = f"decoder.heads.{n}.logits_layer.weight"
            self.slots[key] = TensorSlot(key, off, (Cn - 1, yd + K), ldw)
            off += (Cn - 1) * ldw
        self.total_classes = cls_off
        self.group_ranges["decoder"] = (g0, off)
        g0 = off
        m.prior_logits_off = m.prior_mu_off = m.prior_logvar_off = m.pm_w_off = m.pm_b_off = -1
        if self.behaviour == "supervised":  # model.py:543-551
            m.prior_logits_off = off
            self.slots["prior.logits"] = TensorSlot("prior.logits", off, (K,), K)
            off += _r4(K)
            m.prior_mu_off = off
            self.slots["prior.mu"] = TensorSlot("prior.mu", off, (K, D), D)
            off += _r4(K * D)
            m.prior_logvar_off = off
            self.slots["prior.logvar"] = TensorSlot("prior.logvar", off, (K, D), D)
            off += _r4(K * D)
        else:  # prior.py:20-39
            m.pm_w_off = off
            self.slots["prior_mean.linear.weight"] = TensorSlot("prior_mean.linear.weight", off, (D, K), K)
            off += _r4(D * K)
            m.pm_b_off = off
            self.slots["prior_mean.linear.bias"] = TensorSlot("prior_mean.linear.bias", off, (D,), D)
            off += _r4(D)
        self.group_ranges["prior"] = (g0, off)
        self.n_params = off
        m.n_cols, m.n_num, m.n_disc = len(names), self.n_num, self.n_disc
        m.n_components, m.latent_dim, m.n_hidden = K, D, len(self.hidden)
        m.enc_in, m.total_y, m.y_rows = self.enc_in, self.total_y, self.y_rows
        m.behaviour = 0 if self.behaviour == "supervised" else 1
        m.n_params = off
        m.dropout = float(self.dropout)
        self.model = m

    # ------------------------------------------------------------------
    def cols_array(self):
        arr = (ColT * len(self.cols))()
        for i, c in enumerate(self.cols):
            arr[i] = c
        return arr

    def pack(self, tensors: Mapping[str, np.ndarray]) -> np.ndarray:
        """state_dict-style tensors -> flat float32 buffer."""

        flat = np.zeros(self.n_params, np.float32)
        missing = [k for k in self.slots if k not in tensors]
        if missing:
            raise KeyError(f"missing parameter tensors: {missing[:4]}...")
        for name, slot in self.slots.items():
            t = np.asarray(tensors[name], np.float32).reshape(slot.shape)
            if len(slot.shape) == 1:
                if slot.row_index is not None:
                    flat[slot.offset + slot.row_index] = t
                else:
                    flat[slot.offset : slot.offset + slot.shape[0]] = t
            elif slot.col_index is not None:
                rows = slot.shape[0]
                view = np.lib.stride_tricks.as_strided(
                    flat[slot.offset :], shape=(rows, len(slot.col_index)), strides=(slot.ld * 4, 4), writeable=True)
                view[...] = t[:, slot.col_index]
                for c0, n_cls, e_off, e_ld in slot.embed or []:
                    blk = np.lib.stride_tricks.as_strided(
                        flat[e_off:], shape=(n_cls, rows), strides=(e_ld * 4, 4), writeable=True)
                    blk[...] = t[:, c0:c0 + n_cls].T
            else:
                rows, cols = slot.shape
                n_rows = rows if slot.row_index is None else int(slot.row_index.max()) + 1
                view = np.lib.stride_tricks.as_strided(
                    flat[slot.offset :], shape=(n_rows, cols), strides=(slot.ld * 4, 4), writeable=True)
                if slot.row_index is not None:
                    view[slot.row_index] = t
                else:
                    view[...] = t
        return flat

    def unpack(self, flat: np.ndarray) -> Dict[str, np.ndarray]:
        """flat buffer (parameters or gradients) -> state_dict-style tensors."""

        flat = np.ascontiguousarray(flat, np.float32)
        out: Dict[str, np.ndarray] = {}
        for name, slot in self.slots.items():
            if len(slot.shape) == 1:
                if slot.row_index is not None:
                    out[name] = flat[slot.offset + slot.row_index].copy()
                else:
                    out[name] = flat[slot.offset : slot.offset + slot.shape[0]].copy()
            elif slot.col_index is not None:
                rows = slot.shape[0]
                t = np.zeros(slot.shape, np.float32)
                view = np.lib.stride_tricks.as_strided(flat[slot.offset :], shape=(rows, len(slot.col_index)),
                                                       strides=(slot.ld * 4, 4))
                t[:, slot.col_index] = view
                for c0, n_cls, e_off, e_ld in slot.embed or []:
                    blk = np.lib.stride_tricks.as_strided(flat[e_off:], shape=(n_cls, rows), strides=(e_ld * 4, 4))
                    t[:, c0:c0 + n_cls] = blk.T
                out[name] = t
            else:
                rows, cols = slot.shape
                n_rows = rows if slot.row_index is None else int(slot.row_index.max()) + 1
                view = np.lib.stride_tricks.as_strided(flat[slot.offset :], shape=(n_rows, cols), strides=(slot.ld * 4, 4))
                out[name] = view[slot.row_index].copy() if slot.row_index is not None else view.copy()
        return out

    def init_params(self, seed: int = 0) -> Dict[str, np.ndarray]:
        """torch.nn.Linear default init (uniform +-1/sqrt(fan_in) for weight and bias);
        prior tensors start at zero (model.py:533-551, prior.py:21-22)."""

        rng = np.random.default_rng(seed)
        out: Dict[str, np.ndarray] = {}
        for name, slot in self.slots.items():
            if name.startswith("prior"):
                out[name] = np.zeros(slot.shape, np.float32)
                continue
            if name.endswith(".weight"):
                fan_in = slot.shape[-1]
            else:
                fan_in = self.slots[name[: -len("bias")] + "weight"].shape[-1]
            bound = 1.0 / np.sqrt(max(fan_in, 1))
            out[name] = rng.uniform(-bound, bound, size=slot.shape).astype(np.float32)
        return out
