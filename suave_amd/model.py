"""``SUAVE``-compatible estimator driving the HIP engine.

Mirrors the public surface of the reference's ``suave/model.py`` for the hot
path: constructor kwargs (model.py:277-312), ``fit`` (:883-901), ``encode``
(:4174), ``sample`` (:4284), ``impute`` (:3512), ``predict_proba(attr=...)``
(:3720) -- with every device computation done by ``libsuave_hip.so``.

What is covered: the ELBO warm-up phase for both behaviours (the whole
unsupervised ``fit`` is this phase, model.py:1382-1560), per-epoch validation
ELBO (:2497-2559), posterior-statistics caches (:1669-1741, :2725-2761),
generation, imputation, the Monte-Carlo posterior predictive, and the
supervised schedule: classifier-head phase
(:1743-1864), joint fine-tuning with early stopping (:1866-2086), the four
decoder-refine modes incl. the closed-form EM prior refresh (:2088-2495),
``predict`` / ``predict_proba`` / ``calibrate`` / ``predict_confidence_interval``
and ``save`` / ``load``.  Every encoder / decoder / likelihood computation runs
in ``libsuave_hip.so``; the tiny classifier head and the temperature scaler stay
on PyTorch-ROCm (``supervised.py``) and exchange ``latent`` / ``dlatent`` with
the library.
"""

from __future__ import annotations

import math
import warnings
from typing import Dict, Iterable, Literal, Optional

import numpy as np
import pandas as pd
import torch

from . import data as data_utils
from . import parallel
from . import supervised as sup
from .engine import Engine, PackedTable
from .plan import ModelLayout
from .types import Schema

_DEFAULTS = dict(latent_dim=32, n_components=1, beta=1.5, hidden_dims=(256, 128), dropout=0.1, learning_rate=1e-3,
                 batch_size=128, kl_warmup_epochs=10, warmup_epochs=10, val_split=0.2, random_state=0,
                 gumbel_temperature=1.0, head_epochs=5, finetune_epochs=10, joint_decoder_lr_scale=0.1,
                 early_stop_patience=5)  # model.py:51-68
_REFINE_MODES = ("decoder_only", "decoder_prior", "prior_em_only", "prior_em_decoder")
_ROW_CHUNK = 262144  # rows per device call of the inference methods (bounds the workspace; results do not depend on it)


class _NonFiniteSums(Exception):
    """An epoch of the first attempt of ``fit`` gave non-finite ELBO sums on the fp16-plane arithmetic."""


class DrawReplay:
    """Recorded draws of a reference run, consumed in call order by the training phases (parity tests): the
    per-epoch permutations and, per ELBO call (training batches and validation batches alike), the Exp(1) draws
    behind the Gumbel noise and the reparameterisation eps.  ``SUAVE._replay`` is None in production: the phases
    then draw ``torch.randperm`` on the device and the kernels draw their noise from the counter-based generators."""

    def __init__(self, perms, steps) -> None:
        self.perms = [np.asarray(p, np.int64) for p in perms]
        self.steps = list(steps)
        self.grads = []  # filled by the phases: the flat gradient of every optimisation step, before Adam

    def next_perm(self, n: int, device) -> torch.Tensor:
        p = self.perms.pop(0)
        if len(p) != n:
            raise ValueError(f"replayed permutation has {len(p)} rows, the epoch has {n}")
        return torch.from_numpy(p).to(device)

    def next_noise(self, n_rows: int, device) -> Dict[str, torch.Tensor]:
        st = self.steps.pop(0)
        if st["exp"].shape[0] != n_rows:
            raise ValueError(f"replayed noise has {st['exp'].shape[0]} rows, the batch has {n_rows}")
        return {"exp_noise": torch.from_numpy(np.ascontiguousarray(st["exp"], np.float32)).to(device),
                "eps": torch.from_numpy(np.ascontiguousarray(st["eps"], np.float32)).to(device)}


class SUAVE:
    def __init__(self, schema: Optional[Schema] = None, *, behaviour: Literal["supervised", "unsupervised"] = "supervised",
                 latent_dim: Optional[int] = None, n_components: int = 1, beta: float = 1.5,
                 hidden_dims: Optional[Iterable[int]] = None, head_hidden_dims: Iterable[int] = (),
                 classification_loss_weight: Optional[float] = None, dropout: Optional[float] = None,
                 learning_rate: Optional[float] = None, batch_size: Optional[int] = None,
                 warmup_epochs: Optional[int] = None, kl_warmup_epochs: Optional[int] = None,
                 head_epochs: Optional[int] = None, finetune_epochs: Optional[int] = None,
                 decoder_refine_epochs: Optional[int] = None, decoder_refine_mode: str = "decoder_prior",
                 decoder_refine_prior_lambda: float = 1e-4, early_stop_patience: Optional[int] = None,
                 joint_decoder_lr_scale: float = 0.1, val_split: float = 0.2, stratify: bool = True,
                 random_state: int = 0, gumbel_temperature: float = 1.0,
                 tau_start: float = 1.0, tau_min: float = 1e-3, tau_decay: float = 0.01,
                 device: Optional[str] = None) -> None:
        behaviour = str(behaviour).lower()
        if behaviour not in {"supervised", "unsupervised"}:
            raise ValueError("behaviour must be either 'supervised' or 'unsupervised'")
        if n_components <= 0:
            raise ValueError("n_components must be positive")
        self.schema = schema
        self.behaviour = behaviour
        self.n_components = int(n_components)
        self.beta = float(beta)
        self.latent_dim = int(_DEFAULTS["latent_dim"] if latent_dim is None else latent_dim)
        if self.latent_dim <= 0:
            raise ValueError("latent_dim must be positive")
        self.hidden_dims = tuple(int(h) for h in (_DEFAULTS["hidden_dims"] if hidden_dims is None else hidden_dims))
        if any(h <= 0 for h in self.hidden_dims):
            raise ValueError("hidden_dims must contain positive integers")
        self.dropout = float(_DEFAULTS["dropout"] if dropout is None else dropout)
        if not 0 <= self.dropout < 1:
            raise ValueError("dropout must lie in the [0, 1) range")
        self.learning_rate = float(_DEFAULTS["learning_rate"] if learning_rate is None else learning_rate)
        if self.learning_rate <= 0:
            raise ValueError("learning_rate must be positive")
        self.batch_size = int(_DEFAULTS["batch_size"] if batch_size is None else batch_size)
        if self.batch_size <= 0:
            raise ValueError("batch_size must be positive")
        self.warmup_epochs = int(_DEFAULTS["warmup_epochs"] if warmup_epochs is None else warmup_epochs)
        self.kl_warmup_epochs = int(_DEFAULTS["kl_warmup_epochs"] if kl_warmup_epochs is None else kl_warmup_epochs)
        self.head_hidden_dims = tuple(int(h) for h in head_hidden_dims)
        if any(h <= 0 for h in self.head_hidden_dims):
            raise ValueError("head_hidden_dims must contain positive integers")
        self._classification_loss_weight_user = classification_loss_weight is not None
        self.classification_loss_weight = None if classification_loss_weight is None else float(classification_loss_weight)
        self.head_epochs = int(_DEFAULTS["head_epochs"] if head_epochs is None else head_epochs)
        self.finetune_epochs = int(_DEFAULTS["finetune_epochs"] if finetune_epochs is None else finetune_epochs)
        self._decoder_refine_epochs_user = decoder_refine_epochs is not None
        self.decoder_refine_epochs = None if decoder_refine_epochs is None else int(decoder_refine_epochs)
        if decoder_refine_mode not in _REFINE_MODES:
            raise ValueError("decoder_refine_mode must be one of " + ", ".join(_REFINE_MODES))
        self.decoder_refine_mode = decoder_refine_mode
        self.decoder_refine_prior_lambda = float(decoder_refine_prior_lambda)
        self.early_stop_patience = int(_DEFAULTS["early_stop_patience"] if early_stop_patience is None else early_stop_patience)
        self.joint_decoder_lr_scale = float(joint_decoder_lr_scale)
        self.stratify = bool(stratify)
        self.val_split = float(val_split)
        self.random_state = int(random_state)
        self.gumbel_temperature = float(gumbel_temperature)
        self._tau_start, self._tau_min, self._tau_decay = float(tau_start), float(tau_min), float(tau_decay)
        self._inference_tau = 1.0
        self._device = device
        self._engine: Optional[Engine] = None
        self._layout: Optional[ModelLayout] = None
        self._norm_stats_per_col: Dict[str, Dict[str, object]] = {}
        self._is_fitted = False
        self._classes: Optional[np.ndarray] = None
        self._class_to_index: Optional[Dict[object, int]] = None
        self._train_cache: Dict[str, torch.Tensor] = {}
        self._warmup_val_history: list = []
        self._train_history: list = []
        self._classifier: Optional[sup.ClassificationHead] = None
        self._temperature_scaler = sup.TemperatureScaler()
        self._is_calibrated = False
        self._joint_val_metrics: Optional[Dict[str, float]] = None
        self._joint_history: list = []
        self._refine_history: list = []
        self._step = 0
        self._replay: Optional[DrawReplay] = None
        self._dp: Optional[parallel.DataParallel] = None  # set by fit when torch.distributed has > 1 rank
        self._sharded = False  # fit(sharded=True): every rank holds its own block of the table's rows
        # which constructor arguments were left to the defaults (model.py:460-473; persisted by save)
        self._auto_configured: Dict[str, bool] = {
            "latent_dim": latent_dim is None, "hidden_dims": hidden_dims is None, "head_hidden_dims": False,
            "classification_loss_weight": classification_loss_weight is None, "dropout": dropout is None,
            "learning_rate": learning_rate is None, "batch_size": batch_size is None,
            "kl_warmup_epochs": kl_warmup_epochs is None, "warmup_epochs": warmup_epochs is None,
            "head_epochs": head_epochs is None, "finetune_epochs": finetune_epochs is None,
            "early_stop_patience": early_stop_patience is None}

    # ------------------------------------------------------------------
    def _noise_for(self, n_rows: int):
        return None if self._replay is None else self._replay.next_noise(n_rows, self._engine.device)

    def _head_keys(self, rows: torch.Tensor) -> Dict[str, object]:
        """Keys of the classifier head's dropout masks for one batch: (table rows, seed x batch counter) -- the same on
        every rank and for every sharding of the batch.  Replayed parity runs keep torch's generator (the reference's)."""

        self._head_batch = getattr(self, "_head_batch", 0) + 1
        if self._replay is not None:
            return {}
        return {"rows": rows, "key": (self.random_state * 1000003 + self._head_batch) & 0x7FFFFFFF}

    @staticmethod
    def _rows_here(X, row_offset: Optional[int], *others):
        """(global index of the first row computed here, this rank's rows of X, the same rows of ``others``).
        ``row_offset`` given: the caller sharded X itself; otherwise X is the same on every rank of
        torch.distributed and rank r takes ``parallel.shard_rows`` of it (everything on one rank)."""

        g0, n_loc = parallel.generation_span(len(X), row_offset)
        if row_offset is not None or n_loc == len(X):
            return (g0, X) + tuple(others)
        sl = slice(g0, g0 + n_loc)
        cut = lambda o: None if o is None else (o.iloc[sl] if hasattr(o, "iloc") else np.asarray(o)[sl])
        return (g0, X.iloc[sl]) + tuple(cut(o) for o in others)

    @staticmethod
    def _gathered(out, gather: bool, row_offset: Optional[int]):
        """``gather=True``: every rank gets all ranks' rows, in row order (host objects: meant for modest outputs)."""

        if not gather or row_offset is not None or not parallel.active():
            return out
        parts = parallel.all_gather_objects(out)
        if isinstance(out, pd.DataFrame):
            return pd.concat(parts)
        if isinstance(out, torch.Tensor):
            return torch.cat(parts)
        if isinstance(out, dict):
            cat = lambda vs: torch.cat(vs) if isinstance(vs[0], torch.Tensor) else np.concatenate(vs)
            return {k: cat([p[k] for p in parts]) for k in out}
        return np.concatenate(parts)

    def _record_step_gradient(self) -> None:
        if self._replay is not None:
            self._replay.grads.append(self._engine.grads.clone())

    def _shard(self, rows: torch.Tensor, table: Optional[PackedTable] = None) -> torch.Tensor:
        """This rank's rows of a global batch (global row ids): all of them on one GPU; with a replicated table a
        contiguous share of the batch; with a row-sharded table the batch's rows that live here."""

        if self._dp is None:
            return rows
        if table is not None and self._sharded:
            return table.local_rows(rows)
        return self._dp.shard(rows)

    def _range(self, table: PackedTable, start: int, count: int):
        """This rank's part [begin, end) of the global row range [start, start + count) of ``table``."""

        if self._dp is None:
            return start, start + count
        if self._sharded:
            return table.local_range(start, count)
        b, e = self._dp.span(count)
        return start + b, start + e

    def _all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        return t if self._dp is None else self._dp.all_reduce(t)

    def _backward(self, table: PackedTable, idx: torch.Tensor, **kw) -> None:
        """One ELBO forward + backward on this rank's rows ``idx`` of a global batch, gradients reduced over ranks."""

        dp = self._dp
        self._engine.elbo(table, 0, int(idx.numel()), row_index=idx, grad=True, seed=self.random_state, step=self._step,
                          accumulate_sums=True, ready_event=None if dp is None else dp.ready, **kw)
        if dp is not None:
            dp.reduce_gradients(self._engine.grads)

    @staticmethod
    def _arithmetic() -> int:
        import ctypes as C

        from . import lib as _lib

        mode = C.c_int(0)
        _lib.check(_lib.load().suave_gemm_arithmetic_get(C.byref(mode)), "suave_gemm_arithmetic_get")
        return int(mode.value)

    def _use_wide_arithmetic(self, why: str) -> bool:
        """Switch the library's matrix products from the default two fp16 planes per operand (range 65504) to three
        bf16 planes (fp32's exponent range, ~1.6x slower products) -- for the whole process, like
        ``suave_gemm_arithmetic_set``; logged once.  False when the arithmetic was not the default one (nothing to do)."""

        from . import lib as _lib

        if self._arithmetic() != 4:
            return False
        _lib.check(_lib.load().suave_gemm_arithmetic_set(3), "suave_gemm_arithmetic_set")
        warnings.warn(f"{why}: matrix products switched from two fp16 planes to three bf16 planes for this process "
                      "(suave_gemm_arithmetic_set(3)); results keep the 1e-4 parity bar, products run ~1.6x slower",
                      RuntimeWarning, stacklevel=3)
        return True

    def _select_device(self) -> torch.device:
        if self._device is not None:
            return torch.device(self._device)
        if not torch.cuda.is_available():
            raise RuntimeError("suave_amd needs an AMD GPU (no CPU fallback); none is visible")
        return torch.device("cuda:0")

    def _gumbel_temperature_for_epoch(self, epoch: int) -> float:
        """model.py:620-631."""

        if self.behaviour != "unsupervised":
            return 1.0
        return max(self._tau_min, self._tau_start - self._tau_decay * float(epoch))

    def _pack(self, X: pd.DataFrame):
        """DataFrame -> packed table columns on the device (normalisation fused: ``suave_pack_columns``)."""

        info: Dict[str, int] = {}
        out = data_utils.pack_frame_device(X, self.schema, self._norm_stats_per_col, self._engine, info=info)
        if info.get("beyond_fp16", 0):
            self._use_wide_arithmetic(f"{info['beyond_fp16']} normalised cells exceed fp16's range")
        return out

    def _table(self, X: pd.DataFrame, *, row_base: int = 0, n_global: Optional[int] = None) -> PackedTable:
        xnum, codes, mask = self._pack(X.reset_index(drop=True))
        return PackedTable(xnum, codes, mask, self._engine.device, row_base=row_base, n_global=n_global)

    def _frame(self, out_num: torch.Tensor, out_codes: torch.Tensor) -> pd.DataFrame:
        """Decoded columns (device, row-major) -> DataFrame (``sampling.decoder_outputs_to_frame``)."""

        n = int(out_num.shape[0]) if self._layout.n_num else int(out_codes.shape[0])
        asm = data_utils.FrameAssembler(self.schema, self._norm_stats_per_col, self._engine, n, max(n, 1))
        if n:
            asm.add(0, out_num, out_codes)
        return asm.frame()

    def _require_fitted(self, what: str) -> None:
        if not self._is_fitted or self._engine is None:
            raise RuntimeError(f"Model must be fitted before calling {what}")

    # ------------------------------------------------------------------
    def fit(self, X: pd.DataFrame, y=None, *, schema: Optional[Schema] = None, epochs: Optional[int] = None,
            batch_size: Optional[int] = None, kl_warmup_epochs: Optional[int] = None,
            warmup_epochs: Optional[int] = None, head_epochs: Optional[int] = None,
            finetune_epochs: Optional[int] = None, decoder_refine_epochs: Optional[int] = None,
            joint_decoder_lr_scale: Optional[float] = None, early_stop_patience: Optional[int] = None,
            sharded: bool = False) -> "SUAVE":
        """The reference's staged schedule (model.py:883-1380): ELBO warm-up, then for behaviour="supervised" the
        classifier-head phase, joint fine-tuning and decoder refinement.  ``epochs`` overrides the warm-up length.

        Data parallelism (no counterpart in the reference, which is single-device): when ``torch.distributed`` is
        initialised with more than one rank, every rank calls ``fit`` with the SAME ``X`` / ``y``; each global batch
        is sharded by rows over the ranks (``parallel.DataParallel``), gradients are summed over RCCL with the
        decoder-side bucket overlapped with the encoder backward, epoch / validation sums and the EM statistics are
        all-reduced, and every rank ends with bit-identical parameters equal (to fp32 summation order) to those of
        a one-GPU fit on the whole batches.  The classifier head's dropout masks are keyed by the table row as well
        (``supervised.ClassificationHead``), so they too are the one-GPU fit's.

        ``sharded=True`` (with more than one rank): ``X`` / ``y`` are THIS RANK'S BLOCK of the table's rows -- rank r's
        rows follow rank r-1's -- and only that block is packed and kept in HBM (BASELINE config #3: 2 M rows per GPU
        of a 16 M-row table).  The train / validation split, the normalisation statistics, the per-epoch permutation
        and every noise key are those of the concatenated table: a global batch is a slice of one permutation of the
        global train rows, and each rank computes the rows of it that it holds.  Labels (small) are gathered on every
        rank; so are the posterior caches at the end of the fit (conditional sampling draws from all training rows)."""

        if schema is not None:
            self.schema = schema
        if self.schema is None:
            raise ValueError("A schema must be provided to fit the model")
        missing = [c for c in self.schema.feature_names if c not in X.columns]
        if missing:
            raise ValueError(f"Missing columns in input data: {missing}")
        if self.behaviour == "supervised" and y is None:
            raise ValueError("Targets must be provided when behaviour='supervised'")
        bs = int(batch_size or self.batch_size)
        n_epochs = int(epochs if epochs is not None else (warmup_epochs if warmup_epochs is not None else self.warmup_epochs))
        kl_epochs = int(self.kl_warmup_epochs if kl_warmup_epochs is None else kl_warmup_epochs)
        n_head = int(self.head_epochs if head_epochs is None else head_epochs)
        n_joint = int(self.finetune_epochs if finetune_epochs is None else finetune_epochs)
        if decoder_refine_epochs is not None:
            n_refine = int(decoder_refine_epochs)
        elif self._decoder_refine_epochs_user:
            n_refine = int(self.decoder_refine_epochs)
        else:
            n_refine = n_epochs  # model.py:1160-1166
        lr_scale = float(self.joint_decoder_lr_scale if joint_decoder_lr_scale is None else joint_decoder_lr_scale)
        patience = int(self.early_stop_patience if early_stop_patience is None else early_stop_patience)
        X = X.loc[:, list(self.schema.feature_names)].reset_index(drop=True)
        self._sharded = bool(sharded) and parallel.active()
        n_local = len(X)
        y_arr = None if y is None else np.asarray(y).reshape(-1)
        if y_arr is not None and len(y_arr) != n_local:
            raise ValueError("X and y must have matching first dimensions")
        row_lo = 0  # global index of this rank's first row
        if self._sharded:
            sizes = [int(v) for v in parallel.all_gather_objects(n_local)]
            row_lo = int(sum(sizes[: torch.distributed.get_rank()]))
            n = int(sum(sizes))
            if y_arr is not None:  # labels are small: every rank holds all of them (split, class weights, conditional draws)
                y_arr = np.concatenate(parallel.all_gather_objects(y_arr))
        else:
            n = n_local
        rng = np.random.default_rng(self.random_state)
        n_val = int(round(n * self.val_split)) if n > 1 else 0  # split_train_val (data.py:17-104)
        n_val = min(max(n_val, 0), max(n - 1, 0))
        if y_arr is not None and self.stratify and n_val > 0:  # per-class share of the validation rows
            val_parts = []
            for c in np.unique(y_arr):
                idx_c = rng.permutation(np.flatnonzero(y_arr == c))
                val_parts.append(idx_c[: int(round(len(idx_c) * self.val_split))])
            val_idx = np.sort(np.concatenate(val_parts)) if val_parts else np.empty(0, np.int64)
            train_idx = np.setdiff1d(np.arange(n), val_idx)
            if len(train_idx) == 0:
                train_idx, val_idx = val_idx[:1], val_idx[1:]
        else:
            order = rng.permutation(n)
            val_idx, train_idx = np.sort(order[:n_val]), np.sort(order[n_val:])
        n_val = len(val_idx)
        # this rank's rows of the two splits: positions [base, base + count) of the sorted global split
        tr_b, tr_e = np.searchsorted(train_idx, [row_lo, row_lo + n_local])
        va_b, va_e = np.searchsorted(val_idx, [row_lo, row_lo + n_local])
        X_train, X_val = X.iloc[train_idx[tr_b:tr_e] - row_lo], X.iloc[val_idx[va_b:va_e] - row_lo]
        self._norm_stats_per_col = data_utils.fit_norm_stats(X_train, self.schema, sharded=self._sharded)  # train split
        y_train_t = y_val_t = None
        class_weights = None
        if y_arr is not None:
            self._classes = np.unique(y_arr[train_idx])
            self.classes_ = self._classes
            self._class_to_index = {c.item() if isinstance(c, np.generic) else c: i for i, c in enumerate(self._classes)}
            self._train_target_indices = self._map_targets(y_arr[train_idx])
            counts = np.bincount(self._train_target_indices, minlength=len(self._classes)).astype(np.float32)
            class_weights = counts.sum() / (len(counts) * counts)  # model.py:1116-1122
        self._layout = ModelLayout(self.schema, self.n_components, self.latent_dim, self.hidden_dims, self.dropout,
                                   self.behaviour, self._norm_stats_per_col)
        self._engine = eng = Engine(self._layout, self._select_device())
        eng.load_state(self._layout.init_params(seed=self.random_state))
        if y_arr is not None:
            y_train_t = torch.from_numpy(self._train_target_indices).to(eng.device)
            if n_val:
                y_val_t = torch.from_numpy(self._map_targets(y_arr[val_idx])).to(eng.device)
        train = self._table(X_train, row_base=int(tr_b), n_global=len(train_idx))
        val = self._table(X_val, row_base=int(va_b), n_global=n_val) if n_val else None
        self._dp = parallel.DataParallel.from_environment(self._layout, eng.device)
        if self._dp is not None:
            # one arithmetic on every rank: a rank whose block of a sharded table left fp16's range switched by itself
            if any(int(m) != 4 for m in parallel.all_gather_objects(self._arithmetic())):
                self._use_wide_arithmetic("another rank's rows exceed fp16's range")
            if self._replay is not None:
                raise RuntimeError("recorded draws (parity tests) are per batch row and cannot be replayed on sharded batches")
            if not self._sharded:
                self._dp.check_replicated(train.mask, train.xnum, train.codes)
        # The schedule runs once; if an epoch's ELBO sums come out non-finite under the default arithmetic (two fp16
        # planes per operand: an activation or scaled gradient beyond 65504 overflows where the reference's fp32 does
        # not), the arithmetic is switched to three bf16 planes (fp32's exponent range) and the schedule runs again
        # from the same initial state and seeds.  The sums are all-reduced, so every rank takes the same decision.
        for _attempt in (0, 1):
            self._nonfinite = False
            self._head_batch = 0
            self._val_calls = 0
            self._abort_on_nonfinite = _attempt == 0 and self._replay is None and self._arithmetic() == 4
            if _attempt:
                eng.load_state(self._layout.init_params(seed=self.random_state))
            try:
                self._run_schedule(train, val, y_train_t, y_val_t, class_weights, n_epochs, n_head, n_joint, n_refine, bs,
                                   kl_epochs, lr_scale, patience)
            except _NonFiniteSums:
                self._use_wide_arithmetic("an epoch of the fit produced non-finite ELBO sums")
                continue
            break
        if self._nonfinite:
            warnings.warn("non-finite ELBO sums during fit (not an overflow of the fp16 planes: the matrix products "
                          "run, or were re-run, on an arithmetic with fp32's exponent range)", RuntimeWarning, stacklevel=2)
        self._cache_training_statistics(train, bs)
        self._temperature_scaler = sup.TemperatureScaler()
        self._is_calibrated = False
        self._is_fitted = True
        return self

    def _run_schedule(self, train, val, y_train_t, y_val_t, class_weights, n_epochs, n_head, n_joint, n_refine, bs,
                      kl_epochs, lr_scale, patience) -> None:
        """The staged schedule on packed tables, from freshly initialised parameters (model.py:1168-1380)."""

        eng = self._engine
        self._gen = torch.Generator(device=eng.device)
        self._gen.manual_seed(self.random_state)
        self._step = 0
        self._classifier = None
        self._warmup_phase(train, val, n_epochs, bs, kl_epochs)
        supervised = self.behaviour == "supervised" and y_train_t is not None
        if supervised:
            torch.manual_seed(self.random_state)
            self._classifier = sup.ClassificationHead(self.latent_dim, max(len(self._classes), 2), class_weight=(
                class_weights if len(self._classes) >= 2 else None), dropout=self.dropout,
                hidden_dims=self.head_hidden_dims).to(eng.device)
            self._train_head_phase(self._posterior_mean(train), y_train_t, n_head, bs)
            warmup_scale = self._warmup_val_history[-1]["nll"] if self._warmup_val_history else None
            ce_scale = None
            if val is not None and y_val_t is not None:
                ce_scale = self._validation_scores(val, bs, y_val_t, None).get("classification_loss")
            if not (self._classification_loss_weight_user and self.classification_loss_weight is not None):
                self.classification_loss_weight = sup.derive_classification_loss_weight(warmup_scale, ce_scale)
            self._joint_finetune(train, val, y_train_t, y_val_t, n_joint, bs, lr_scale, patience,
                                 float(self.classification_loss_weight))
            self._refine_decoder(train, val, n_refine, bs, patience)
        else:
            self._joint_val_metrics = self._warmup_val_history[-1] if self._warmup_val_history else None

    def _map_targets(self, values) -> np.ndarray:
        """model.py `_map_targets_to_indices`: labels -> class indices, unknown labels raise."""

        out = np.empty(len(values), np.int64)
        for i, v in enumerate(np.asarray(values).reshape(-1)):
            key = v.item() if isinstance(v, np.generic) else v
            if key not in self._class_to_index:
                raise ValueError(f"Unknown class label '{key}'")
            out[i] = self._class_to_index[key]
        return out

    # ------------------------------------------------------------------
    def _epoch_batches(self, n_rows: int, eff: int):
        if self._replay is not None:
            perm = self._replay.next_perm(n_rows, self._engine.device)
        else:
            perm = torch.randperm(n_rows, device=self._engine.device, generator=self._gen, dtype=torch.int64)
        perm32 = perm.to(torch.int32)
        for start in range(0, n_rows, max(eff, 1)):  # last batch ragged, no drop-last (model.py:1475-1476)
            yield perm[start:start + eff], perm32[start:start + eff]

    def _elbo_batches(self, table: PackedTable, eff: int):
        """One epoch of global batches over ``table``: (global row ids of the batch int64, this rank's rows of it int32)
        -- the whole batch on one GPU, a contiguous share of it with a replicated table, the rows held here with a
        row-sharded table (selected once per epoch: no per-step host sync)."""

        if self._dp is None or not self._sharded:
            for idx64, idx in self._epoch_batches(table.n_global, eff):
                yield idx64, self._shard(idx)
            return
        perm = torch.randperm(table.n_global, device=self._engine.device, generator=self._gen, dtype=torch.int64)
        for b, e, loc in parallel.sharded_epoch_batches(perm, eff, table.row_base, table.n_rows):
            yield perm[b:e], loc

    def _warmup_phase(self, train: PackedTable, val: Optional[PackedTable], n_epochs: int, bs: int, kl_epochs: int) -> None:
        """_run_warmup_phase (model.py:1382-1560)."""

        eng = self._engine
        n_train = train.n_global
        eff = min(bs, n_train) if n_train else bs
        n_batches = max(1, math.ceil(max(n_train, 1) / max(eff, 1)))
        kl_steps = max(1, kl_epochs * n_batches)  # model.py:1448
        eng.reset_optimizer()
        warm_step = 0
        self._train_history, self._warmup_val_history = [], []
        for epoch in range(n_epochs):
            tau = self._gumbel_temperature_for_epoch(epoch) if self.behaviour == "unsupervised" else self.gumbel_temperature
            if self.behaviour == "unsupervised":
                self._inference_tau = tau
            eng.sums.zero_()
            for idx, loc in self._elbo_batches(train, eff):
                warm_step += 1
                self._step += 1
                beta = self.beta * min(warm_step, kl_steps) / kl_steps  # losses.kl_warmup
                nb = int(idx.numel())
                self._backward(train, loc, beta_scale=beta, tau=tau, global_rows=nb,
                               noise=self._noise_for(nb))
                self._record_step_gradient()
                eng.adam_step(self.learning_rate)
            s = self._epoch_sums()  # ONE host sync per epoch (the reference syncs 4-5x per batch)
            cnt = max(s[3], 1.0)
            self._train_history.append({"reconstruction": s[0] / cnt, "categorical_kl": s[1] / cnt,
                                        "gaussian_kl": s[2] / cnt})
            if val is not None and val.n_global:
                self._warmup_val_history.append(self._elbo_on_table(val, bs, tau, seed=self.random_state + 1 + epoch))

    def _epoch_sums(self) -> np.ndarray:
        """The epoch's (sum reconstruction, sum categorical KL, sum Gaussian KL, rows) over all ranks: one host sync.
        A non-finite sum is recorded (``fit`` then repeats the schedule on the wide arithmetic, see there; the
        reference keeps training on NaNs silently)."""

        s = self._all_reduce(self._engine.sums).cpu().numpy()
        if not np.all(np.isfinite(s)):
            self._nonfinite = True
            if getattr(self, "_abort_on_nonfinite", False):
                raise _NonFiniteSums()
        return s

    def _posterior_mean(self, table: PackedTable) -> torch.Tensor:
        """posterior["mean"] of _collect_posterior_statistics (model.py:1669-1741), kept on the device."""

        tau = self._inference_tau if self.behaviour == "unsupervised" else None
        lo, hi = table.row_base, table.row_base + table.n_rows
        outs = [self._engine.encode(table, s0, min(65536, hi - s0), temperature=tau)["mean"] for s0 in range(lo, hi, 65536)]
        mean = torch.cat(outs) if outs else torch.empty((0, self.latent_dim), device=self._engine.device)
        return parallel.all_gather_rows(mean) if self._sharded else mean  # row-sharded table: every rank gets all rows

    def _train_head_phase(self, latent_mu: torch.Tensor, y_t: torch.Tensor, epochs: int, bs: int) -> None:
        """_train_head_phase (model.py:1743-1864): Adam on the classifier over cached latent means."""

        if self._classifier is None or epochs <= 0 or latent_mu.shape[0] == 0:
            return
        n = latent_mu.shape[0]
        eff = min(bs, n)
        self._classifier.train()
        self._head_history = []
        hd = self._device_head(y_t)
        if hd is not None:  # the phase in the library: two launches per batch, one host sync per epoch
            hd.reset_optimizer()
            latent_mu = latent_mu.contiguous()
            for _ in range(epochs):
                hd.loss_sum.zero_()
                for idx, idx32 in self._epoch_batches(n, eff):
                    key = self._head_keys(idx).get("key", 0)
                    self._engine.head_step(hd.struct(training=True, loss_scale=1.0, loss_rows=float(idx.numel()), key=key,
                                                     adam_lr=self.learning_rate, want_grads=False), latent_mu, idx32)
                self._head_history.append(float(hd.loss_sum.item()) / n)
            if self._dp is not None:
                self._dp.broadcast_module(self._classifier)
            self._classifier.eval()
            return
        opt = torch.optim.Adam(self._classifier.parameters(), lr=self.learning_rate)
        for _ in range(epochs):
            tot = torch.zeros((), device=latent_mu.device, dtype=torch.float64)
            for idx, _ in self._epoch_batches(n, eff):
                loss = self._classifier.loss(self._classifier(latent_mu[idx], **self._head_keys(idx)), y_t[idx])
                opt.zero_grad()
                loss.backward()
                opt.step()
                tot += loss.detach().double() * idx.numel()
            self._head_history.append(float(tot.item()) / n)
        if self._dp is not None:  # the phase is replicated (cached latents, same seed); rank 0's bits win
            self._dp.broadcast_module(self._classifier)
        self._classifier.eval()

    def _device_head(self, y_t: torch.Tensor):
        """The library-side head state (``supervised.DeviceHead``) when the classifier is one the HIP kernels serve --
        no hidden layers, <= 64 classes -- else None (the torch module computes the head; ``SUAVE_TORCH_HEAD=1`` forces
        that, for comparisons).  Replayed parity runs with head dropout keep torch's generator, as the reference."""

        import os

        clf = self._classifier
        if clf is None or os.environ.get("SUAVE_TORCH_HEAD") or not sup.DeviceHead.supports(clf):
            return None
        hd = getattr(self, "_dev_head", None)
        if hd is None or hd.clf is not clf or hd.targets.numel() != y_t.numel():
            hd = self._dev_head = sup.DeviceHead(clf, y_t)
        if self._replay is not None and hd.dropout > 0.0:
            return None
        return hd

    def _validation_scores(self, val: Optional[PackedTable], bs: int, y_val_t: Optional[torch.Tensor],
                           weight: Optional[float]) -> Dict[str, float]:
        """_compute_validation_scores (model.py:2561-2634)."""

        if val is None or val.n_global == 0:
            return {}
        tau = self._inference_tau if self.behaviour == "unsupervised" else self.gumbel_temperature
        self._val_calls = getattr(self, "_val_calls", 0) + 1
        m = dict(self._elbo_on_table(val, bs, tau, seed=self.random_state + 7919 * self._val_calls))
        for key in ("brier", "ece", "classification_loss", "auroc", "joint_objective"):
            m.setdefault(key, float("nan"))
        if self._classifier is None or y_val_t is None:
            if weight is not None and math.isfinite(m["nll"]):
                m["joint_objective"] = m["nll"]
            return m
        was_training = self._classifier.training
        self._classifier.eval()
        with torch.no_grad():  # logits, weighted CE and the three metrics stay on the device: one sync for all four
            hd = getattr(self, "_dev_head", None)
            if hd is not None and hd.clf is self._classifier and self._device_head(hd.targets) is not None:
                logits, ce = hd.evaluate(self._engine, self._posterior_mean(val), y_val_t)
            else:
                logits = self._classifier(self._posterior_mean(val))
                ce = self._classifier.loss(logits, y_val_t).to(torch.float64).reshape(1)
            vals = torch.cat([ce, sup.device_classification_metrics(torch.softmax(logits, dim=-1), y_val_t)]).cpu().numpy()
        for key, v in zip(("classification_loss", "brier", "ece", "auroc"), vals):
            m[key] = float(v)
        if weight is not None:
            ok = math.isfinite(m["classification_loss"]) and math.isfinite(m["nll"])
            m["joint_objective"] = m["nll"] + float(weight) * m["classification_loss"] if ok else float("nan")
        if was_training:
            self._classifier.train()
        return m

    def _capture_state(self) -> Dict[str, object]:
        """_capture_model_state (model.py:2680-2694): device-side copies."""

        st = {"params": self._engine.params.clone()}
        if self._classifier is not None:
            st["classifier"] = {k: v.detach().clone() for k, v in self._classifier.state_dict().items()}
        return st

    def _restore_state(self, st: Dict[str, object]) -> None:
        self._engine.params.copy_(st["params"])
        if self._classifier is not None and "classifier" in st:
            self._classifier.load_state_dict(st["classifier"])

    def _joint_finetune(self, train: PackedTable, val: Optional[PackedTable], y_t: torch.Tensor,
                        y_val_t: Optional[torch.Tensor], epochs: int, bs: int, lr_scale: float, patience: int,
                        weight: float) -> int:
        """_run_joint_finetune (model.py:1866-2086): ELBO + weight * CE(classifier(latent)); encoder and classifier
        at lr, decoder and prior at lr * lr_scale; early stopping on the validation metrics."""

        self._joint_history = []
        if epochs <= 0 or self._classifier is None:
            return 0
        eng, clf = self._engine, self._classifier
        weight = float(max(weight, 0.0))
        eng.reset_optimizer()  # the reference builds a fresh Adam per phase (model.py:1909)
        hd = self._device_head(y_t)
        if hd is not None:
            hd.reset_optimizer()
        clf_opt = torch.optim.Adam(clf.parameters(), lr=self.learning_rate) if hd is None else None
        lrs = {"encoder": self.learning_rate, "decoder": self.learning_rate * lr_scale, "prior": self.learning_rate * lr_scale}
        clf.train()
        n = train.n_global
        eff = min(bs, n) if n else bs
        best_state = self._capture_state()
        best = self._validation_scores(val, bs, y_val_t, weight) or None
        if best:
            self._joint_val_metrics = best
        bad, done = 0, 0
        tau = self.gumbel_temperature
        for _ in range(epochs):
            eng.sums.zero_()
            ce_tot = torch.zeros((), device=eng.device, dtype=torch.float64)
            if hd is not None:
                hd.loss_sum.zero_()
            for idx64, loc in self._elbo_batches(train, eff):
                nb = int(idx64.numel())  # the global batch; `loc` are this rank's rows of it
                self._step += 1
                nz = self._noise_for(nb)
                loc64 = loc.long()
                if hd is not None:
                    # one pass (suave_elbo_joint_backward): the head's CE on the step's own latent inside the backward
                    key = self._head_keys(loc64).get("key", 0)
                    wsum = None if self._dp is None else hd.batch_weight(idx64)  # the GLOBAL batch's class-weight sum
                    self._backward(train, loc, beta_scale=self.beta, tau=tau, global_rows=nb, noise=nz,
                                   head=hd.struct(training=True, loss_scale=weight, loss_rows=float(nb), key=key, weight_sum=wsum))
                    if self._dp is not None:
                        self._dp.all_reduce(hd.grads)
                    self._record_step_gradient()
                    eng.adam_step(lrs)
                    hd.t += 1
                    eng.adam_flat(hd.flat, hd.grads, hd.m, hd.v, hd.t, self.learning_rate)
                    continue
                fwd = eng.elbo(train, 0, int(loc.numel()), row_index=loc, beta_scale=self.beta, tau=tau, grad=False,
                               training=True, seed=self.random_state, step=self._step, want=("latent",), noise=nz)
                latent = fwd["latent"].requires_grad_(True)
                hk = self._head_keys(loc64)
                if self._dp is None:
                    class_loss = clf.loss(clf(latent, **hk), y_t[loc64])
                else:  # the weighted mean over the GLOBAL batch: this rank's terms over the global weight sum
                    class_loss = clf.loss_terms(clf(latent, **hk), y_t[loc64]) / clf.loss_weight(y_t[idx64])
                clf_opt.zero_grad()
                (class_loss * weight).backward()
                self._backward(train, loc, beta_scale=self.beta, tau=tau, global_rows=nb,
                               dlatent=latent.grad.contiguous(), noise=nz)
                if self._dp is not None:
                    self._dp.reduce_module_gradients(clf)
                self._record_step_gradient()
                eng.adam_step(lrs)
                clf_opt.step()
                ce_tot += class_loss.detach().double() * nb
            s = self._epoch_sums()
            cnt = max(s[3], 1.0)
            elbo_loss = -(s[0] - self.beta * (s[1] + s[2])) / cnt
            if hd is not None:
                ce_tot = hd.loss_sum[0].clone()
            ce = float(self._all_reduce(ce_tot).item()) / cnt
            metrics = self._validation_scores(val, bs, y_val_t, weight)
            self._joint_history.append({"train_elbo_loss": elbo_loss, "train_classification_loss": ce,
                                        "train_joint": elbo_loss + weight * ce, "val": metrics})
            done += 1
            if best is None or sup.is_better_metrics(metrics, best):
                best, best_state, bad = metrics, self._capture_state(), 0
            else:
                bad += 1
                if bad > patience:
                    break
        self._restore_state(best_state)
        self._joint_val_metrics = best
        clf.eval()
        return done

    def _prior_view(self, name: str) -> torch.Tensor:
        slot = self._layout.slots[name]
        return self._engine.params[slot.offset: slot.offset + int(np.prod(slot.shape))].view(*slot.shape)

    def _closed_form_prior_update(self, train: PackedTable) -> None:
        """_closed_form_prior_update_from_stats (model.py:2447-2495) from posterior responsibilities."""

        eng, K, D = self._engine, self.n_components, self.latent_dim
        w = torch.zeros(K, device=eng.device, dtype=torch.float64)
        m1 = torch.zeros((K, D), device=eng.device, dtype=torch.float64)
        m2 = torch.zeros((K, D), device=eng.device, dtype=torch.float64)
        r0, r1 = self._range(train, 0, train.n_global)  # sufficient statistics are sums over rows: shard, then all-reduce
        for s0 in range(r0, r1, 65536):
            o = eng.encode(train, s0, min(65536, r1 - s0))
            p = o["probs"].double()
            mu = o["enc_out"][:, K:K + K * D].reshape(-1, K, D).double()
            lv = o["enc_out"][:, K + K * D:].reshape(-1, K, D).double()
            w += p.sum(dim=0)
            m1 += (p.unsqueeze(-1) * mu).sum(dim=0)
            m2 += (p.unsqueeze(-1) * (torch.exp(lv) + mu * mu)).sum(dim=0)
        for t in (w, m1, m2):
            self._all_reduce(t)
        total = float(w.sum().item())
        if total <= 0.0:
            return
        cw = torch.clamp(w.unsqueeze(-1), min=1e-6)
        means = m1 / cw
        var = torch.clamp(m2 / cw - means * means, min=1e-6)
        self._prior_view("prior.logits").copy_(torch.log(torch.clamp(w / total, min=1e-6)).float())
        self._prior_view("prior.mu").copy_(means.float())
        self._prior_view("prior.logvar").copy_(torch.log(var).float())

    def _refine_decoder(self, train: PackedTable, val: Optional[PackedTable], epochs: int, bs: int, patience: int) -> int:
        """_refine_decoder_after_joint (model.py:2088-2445): encoder and classifier frozen."""

        self._refine_history = []
        if self.behaviour != "supervised" or epochs <= 0 or train.n_global == 0:
            return 0
        eng, mode = self._engine, self.decoder_refine_mode
        patience = max(int(patience), 0)
        w_cls = self.classification_loss_weight
        frozen_ce = (self._joint_val_metrics or {}).get("classification_loss")
        frozen_ce = float(frozen_ce) if frozen_ce is not None and math.isfinite(float(frozen_ce)) else None

        def decorate(m: Dict[str, float]) -> Optional[Dict[str, float]]:
            if not m:
                return None
            m = dict(m)
            if w_cls is not None and frozen_ce is not None:
                m["classification_loss"] = frozen_ce
                if math.isfinite(m.get("nll", float("nan"))):
                    m["joint_objective"] = m["nll"] + float(w_cls) * frozen_ce
            if m.get("joint_objective") is None:
                m["joint_objective"] = m.get("nll")
            m.setdefault("classification_loss", None)
            return m

        tau = self.gumbel_temperature
        self._val_calls = getattr(self, "_val_calls", 0)

        def val_elbo():
            if val is None or val.n_global == 0:
                return {}
            self._val_calls += 1
            return self._elbo_on_table(val, bs, tau, seed=self.random_state + 7919 * self._val_calls)

        consumed, after_em = 0, None
        if mode in ("prior_em_only", "prior_em_decoder"):
            self._closed_form_prior_update(train)
            after_em = val_elbo()
            consumed = 1
            if mode == "prior_em_only":
                return 1
        train_prior = mode == "decoder_prior"
        lrs = {"encoder": 0.0, "decoder": self.learning_rate, "prior": self.learning_rate if train_prior else 0.0}
        eng.reset_optimizer()
        anchors = None
        if train_prior:
            anchors = [(self._layout.slots[k], self._prior_view(k).clone())
                       for k in ("prior.logits", "prior.mu", "prior.logvar")]
        best_state = self._capture_state()
        best = decorate(after_em or val_elbo())
        bad, done = 0, 0
        n = train.n_global
        eff = min(bs, n)
        lam = float(self.decoder_refine_prior_lambda)
        for _ in range(epochs):
            eng.sums.zero_()
            for idx, loc in self._elbo_batches(train, eff):
                self._step += 1
                nb = int(idx.numel())
                self._backward(train, loc, beta_scale=self.beta, tau=tau, global_rows=nb,
                               noise=self._noise_for(nb))
                if anchors and lam != 0.0:  # + lambda * sum_t mse(prior_t, anchor_t)  (model.py:2338-2346)
                    for slot, snap in anchors:
                        nel = int(np.prod(slot.shape))
                        g = eng.grads[slot.offset: slot.offset + nel].view(*slot.shape)
                        g.add_((self._prior_view(slot.name) - snap) * (2.0 * lam / nel))
                self._record_step_gradient()
                eng.adam_step(lrs)
            done += 1
            s = self._epoch_sums()
            cnt = max(s[3], 1.0)
            m = decorate(val_elbo())
            self._refine_history.append({"train_elbo_loss": -(s[0] - self.beta * (s[1] + s[2])) / cnt, "val": m})
            if m is not None and (best is None or sup.is_better_metrics(m, best)):
                best, best_state, bad = m, self._capture_state(), 0
            else:
                bad += 1
                if patience > 0 and bad > patience:
                    break
        self._restore_state(best_state)
        return done + consumed

    def _elbo_on_table(self, table: PackedTable, batch_size: int, tau: float, seed: int) -> Dict[str, float]:
        """_compute_elbo_on_dataset (model.py:2497-2559): eval mode, beta = self.beta, fresh noise."""

        eng = self._engine
        eng.sums.zero_()
        eff = min(batch_size, table.n_global)
        for i, start in enumerate(range(0, table.n_global, eff)):
            nb = min(eff, table.n_global - start)
            b, e = self._range(table, start, nb)  # noise is keyed by (seed, step, table row): the sums do not depend on the sharding
            eng.elbo(table, b, e - b, beta_scale=self.beta, tau=tau, grad=False, training=False, seed=seed,
                     step=i, accumulate_sums=True, noise=self._noise_for(nb))
        s = self._epoch_sums()
        n = max(s[3], 1.0)
        return {"nll": float(((s[1] + s[2]) * self.beta - s[0]) / n), "reconstruction": float(s[0] / n),
                "categorical_kl": float(s[1] / n), "gaussian_kl": float(s[2] / n)}

    def _cache_training_statistics(self, train: PackedTable, batch_size: int) -> None:
        """_collect_posterior_statistics / _cache_training_statistics (model.py:1669-1741, 2725-2761);
        the caches stay on the device (the reference moves them to the CPU)."""

        eng = self._engine
        K, D = self.n_components, self.latent_dim
        tau = self._inference_tau if self.behaviour == "unsupervised" else None
        outs = []
        chunk = max(batch_size, 65536)
        lo, hi = train.row_base, train.row_base + train.n_rows
        for start in range(lo, hi, chunk):
            outs.append(eng.encode(train, start, min(chunk, hi - start), temperature=tau))
        E = K + 2 * K * D
        cat = lambda key, width: (torch.cat([o[key] for o in outs]) if outs else
                                  torch.empty((0, width), device=eng.device))
        enc, probs, mean, plv = cat("enc_out", E), cat("probs", K), cat("mean", D), cat("logvar", D)
        if self._sharded:  # a row-sharded fit: conditional sampling draws from ALL training rows -> every rank holds them
            enc, probs, mean, plv = (parallel.all_gather_rows(t) for t in (enc, probs, mean, plv))
        if enc.shape[0] == 0:
            self._train_cache = {}
            return
        self._train_cache = {
            "probs": probs.contiguous(),
            "mu": enc[:, K:K + K * D].reshape(-1, K, D).contiguous(),
            "logvar": enc[:, K + K * D:].reshape(-1, K, D).contiguous(),
            "mean": mean,
            "post_logvar": plv,                                          # _train_latent_logvar (model.py:2735-2737)
            "logits": enc[:, :K].contiguous(),                           # _train_component_logits
        }

    # ------------------------------------------------------------------
    def encode(self, X: pd.DataFrame, *, return_components: bool = False, row_offset: Optional[int] = None,
               gather: bool = False):
        """model.py:4174-4282.  ``row_offset`` / ``gather``: rows sharded over the ranks of torch.distributed as in
        ``sample`` (no randomness here: sharding only decides which rows this rank computes)."""

        self._require_fitted("encode")
        _, X = self._rows_here(X, row_offset)
        table = self._table(X.loc[:, list(self.schema.feature_names)])
        K, D = self.n_components, self.latent_dim
        if table.n_rows == 0:
            empty = np.empty((0, D), np.float32)
            if not return_components:
                return self._gathered(empty, gather, row_offset)
            return self._gathered({"mean": empty, "assignments": np.empty((0, K), np.float32), "component_mu": empty.copy(),
                                   "component_logvar": empty.copy()}, gather, row_offset)
        tau = self._inference_tau if self.behaviour == "unsupervised" else None
        parts: Dict[str, list] = {"mean": [], "assignments": [], "component_mu": [], "component_logvar": []}
        for s0 in range(0, table.n_rows, _ROW_CHUNK):  # the workspace is sized by the chunk, not by the table
            out = self._engine.encode(table, s0, min(_ROW_CHUNK, table.n_rows - s0), temperature=tau)
            parts["mean"].append(out["mean"].cpu().numpy())
            if not return_components:
                continue
            enc = out["enc_out"]
            k = enc[:, :K].argmax(dim=1)  # hard assignment on softmax(logits) (model.py:4254-4258)
            rows = torch.arange(enc.shape[0], device=enc.device)
            parts["assignments"].append(torch.nn.functional.one_hot(k, K).float().cpu().numpy())
            parts["component_mu"].append(enc[:, K:K + K * D].reshape(-1, K, D)[rows, k].cpu().numpy())
            parts["component_logvar"].append(enc[:, K + K * D:].reshape(-1, K, D)[rows, k].cpu().numpy())
        if not return_components:
            return self._gathered(np.concatenate(parts["mean"]), gather, row_offset)
        return self._gathered({key: np.concatenate(v) for key, v in parts.items()}, gather, row_offset)

    # ------------------------------------------------------------------
    def _prior_cdf(self) -> torch.Tensor:
        st = self._engine.state()
        if self.behaviour == "supervised":
            lg = np.asarray(st["prior.logits"], np.float64)
        else:
            lg = np.zeros(self.n_components)
        p = np.exp(lg - lg.max())
        p /= p.sum()
        return torch.from_numpy(np.cumsum(p).astype(np.float32)).to(self._engine.device)

    def sample(self, n_samples: int, conditional: bool = False, y=None, batch_size: Optional[int] = None,
               seed: Optional[int] = None, row_offset: Optional[int] = None, gather: bool = False,
               _noise: Optional[Dict[str, torch.Tensor]] = None) -> pd.DataFrame:
        """Generate rows (model.py:4284-4431).  Row ``i`` depends only on ``(seed, i)``: the output is independent
        of ``batch_size`` and of how rows are sharded over GPUs.

        Multi-GPU (no counterpart in the reference): with torch.distributed initialised on more than one rank, every
        rank makes the SAME call and generates rows ``parallel.shard_rows(n_samples, rank, world)`` of it, keyed by the
        global row index -- no collective on the data path; the returned frame is this rank's shard, indexed by the
        global row numbers (``gather=True``: every rank gets the whole frame).  A caller that shards by itself passes
        ``row_offset``: its ``n_samples`` rows (and ``y`` entries) are global rows ``row_offset ..``.

        ``_noise`` injects the draws instead of generating them from ``seed`` (parity tests): ``u`` [n] and ``eps``
        [n,D] of the prior draw, or ``u_row`` [n], ``e_comp`` [n,K] and ``eps`` of the conditional draw, and
        ``normal`` / ``uniform`` [n,n_cols], ``exp`` [n,total classes] of the head samplers."""

        self._require_fitted("sample")
        if n_samples <= 0:
            raise ValueError("n_samples must be positive")
        eng = self._engine
        seed = self.random_state if seed is None else int(seed)
        chunk = int(batch_size or max(self.batch_size, 262144))
        labels_t = None
        if conditional:
            if self.behaviour != "supervised" or self._class_to_index is None:
                raise RuntimeError("Conditional sampling requires supervised targets from the training data")
            if y is None:
                raise ValueError("Targets must be provided when conditional=True")
            y_arr = np.asarray(y, dtype=object).reshape(-1)
            if y_arr.shape[0] != n_samples:
                raise ValueError("y must have length equal to n_samples when conditional=True")
            if row_offset is None:  # this rank's rows of the request
                gb, gn = parallel.generation_span(n_samples, None)
                y_arr = y_arr[gb:gb + gn]
            try:
                lab = np.asarray([self._class_to_index[v.item() if isinstance(v, np.generic) else v] for v in y_arr],
                                 np.int32)
            except KeyError as exc:
                raise ValueError(f"Unknown class '{exc.args[0]}' supplied for conditional sampling") from None
            tt = self._train_target_indices
            n_cls = len(self._classes)
            class_rows = np.concatenate([np.flatnonzero(tt == c) for c in range(n_cls)]).astype(np.int64)
            class_off = np.concatenate([[0], np.cumsum([(tt == c).sum() for c in range(n_cls)])]).astype(np.int64)
            for c in np.unique(lab):
                if class_off[c + 1] == class_off[c]:
                    raise ValueError(f"No training samples available for class '{self._classes[c]}' to condition on")
            labels_t = torch.from_numpy(lab).to(eng.device)
            class_rows_t = torch.from_numpy(class_rows).to(eng.device)
            class_off_t = torch.from_numpy(class_off).to(eng.device)
        else:
            cdf = self._prior_cdf()
        g0, n_here = parallel.generation_span(n_samples, row_offset)
        asm = data_utils.FrameAssembler(self.schema, self._norm_stats_per_col, eng, n_here, min(chunk, max(n_here, 1)))
        nz = _noise or {}
        inj0 = 0 if row_offset is not None else g0  # injected draw tensors cover the whole request

        def rows_of(key, start, nb):  # the chunk's rows of an injected draw tensor
            t = nz.get(key)
            return None if t is None else t[inj0 + start:inj0 + start + nb].contiguous()

        for start in range(0, n_here, chunk):
            nb = min(chunk, n_here - start)
            if conditional:
                z, comp = eng.conditional_latents(labels_t[start:start + nb].contiguous(), class_off_t, class_rows_t,
                                                  self._train_cache["probs"], self._train_cache["mu"],
                                                  self._train_cache["logvar"], u_row=rows_of("u_row", start, nb),
                                                  e_comp=rows_of("e_comp", start, nb), eps=rows_of("eps", start, nb),
                                                  seed=seed, row_offset=g0 + start)
            else:
                z, comp = eng.prior_latents(nb, cdf, u=rows_of("u", start, nb), eps=rows_of("eps", start, nb),
                                            seed=seed, row_offset=g0 + start)
            onum, ocodes = eng.decode_sample(z, component=comp, seed=seed + 1, row_offset=g0 + start,
                                             noise_normal=rows_of("normal", start, nb),
                                             noise_uniform=rows_of("uniform", start, nb),
                                             noise_exp=rows_of("exp", start, nb))
            asm.add(start, onum, ocodes)
        out = asm.frame()
        if g0:
            out.index = pd.RangeIndex(g0, g0 + n_here)
        return self._gathered(out, gather, row_offset)

    # ------------------------------------------------------------------
    def impute(self, X: pd.DataFrame, *, only_missing: bool = True,
               assignment_strategy: Optional[Literal["soft", "hard", "sample"]] = None, seed: Optional[int] = None,
               row_offset: Optional[int] = None, gather: bool = False,
               _noise: Optional[Dict[str, torch.Tensor]] = None, _routing: Optional[Dict[str, torch.Tensor]] = None
               ) -> pd.DataFrame:
        """Single-pass imputation at the posterior mean latent (model.py:3512-3692).

        ``_noise`` injects the draws instead of generating them from ``seed`` (parity tests): ``assign_exp``
        [B,K] Exp(1) of the relaxed Gumbel assignment, ``normal`` / ``uniform`` [B,n_cols] and ``exp``
        [B,total classes] of the head samplers.  ``_routing`` (a dict) receives the (latent, assignments) pair
        handed to the decoder -- what the reference's tests spy on (tests/test_impute.py:141-146)."""

        self._require_fitted("impute")
        missing_cols = [c for c in self.schema.feature_names if c not in X.columns]
        if missing_cols:
            raise KeyError("Input is missing required columns: " + ", ".join(missing_cols))
        strategy = (assignment_strategy or ("hard" if self.behaviour == "supervised" else "soft")).lower()
        if strategy not in {"soft", "hard", "sample"}:
            raise ValueError("assignment_strategy must be one of 'soft', 'hard' or 'sample'")
        eng = self._engine
        seed = self.random_state if seed is None else int(seed)
        n_all = len(X)
        g0, X = self._rows_here(X, row_offset)  # rows sharded over ranks as in sample(): keys use the global row index
        aligned = X.loc[:, list(self.schema.feature_names)].reset_index(drop=True)
        xnum, codes, mask = self._pack(aligned)
        table = PackedTable(xnum, codes, mask, eng.device)
        K, D = self.n_components, self.latent_dim
        tau = self._inference_tau if self.behaviour == "unsupervised" else None
        nz = _noise or {}
        e_all = None
        if strategy == "sample":
            if nz.get("assign_exp") is not None:
                e_all = nz["assign_exp"].to(eng.device)
            else:
                # torch's generator is not keyed by row: a rank of a sharded call draws the whole request's tensor (the
                # same on every rank) and keeps its rows; a caller-sharded call (row_offset) gets a stream of its own
                g = torch.Generator(device=eng.device)
                g.manual_seed(seed if row_offset is None else (seed * 1000003 + int(row_offset)) & 0x7FFFFFFF)
                e_all = torch.empty((n_all, K), device=eng.device).exponential_(generator=g)
                if row_offset is None:
                    e_all = e_all[g0:g0 + table.n_rows]
        rows_of = lambda key, s0, nb: None if nz.get(key) is None else nz[key][s0:s0 + nb].contiguous()
        nums, cods, lat, asg = [], [], [], []
        for s0 in range(0, table.n_rows, _ROW_CHUNK):  # row chunks: the generators are keyed by the table row
            nb = min(_ROW_CHUNK, table.n_rows - s0)
            comp = assign = None
            if strategy == "hard":  # model.py:3627-3634: argmax of the posterior probabilities, that component's mean
                z, comp = eng.encode_hard(table, s0, nb, temperature=tau)
            else:
                out = eng.encode(table, s0, nb, temperature=tau)
                enc, probs = out["enc_out"], out["probs"]
                mu = enc[:, K:K + K * D].reshape(-1, K, D)
            if strategy == "hard":
                pass
            elif strategy == "soft":  # model.py:3635-3637
                assign = probs.contiguous()
                z = out["mean"]
            else:  # relaxed Gumbel-softmax weights, model.py:3638-3644
                t = max(float(self._inference_tau if self.behaviour == "unsupervised" else self.gumbel_temperature), 1e-6)
                assign = torch.softmax((enc[:, :K] - torch.log(e_all[s0:s0 + nb])) / t, dim=1).contiguous()
                z = (assign.unsqueeze(-1) * mu).sum(dim=1).contiguous()
            if _routing is not None:
                lat.append(z)
                asg.append(assign if assign is not None else torch.nn.functional.one_hot(comp.long(), K).float())
            onum, ocodes = eng.decode_sample(z, component=comp, assignments=assign, seed=seed + 1, row_offset=g0 + s0,
                                             noise_normal=rows_of("normal", s0, nb), noise_uniform=rows_of("uniform", s0, nb),
                                             noise_exp=rows_of("exp", s0, nb))
            nums.append(onum)
            cods.append(ocodes)
        if _routing is not None:
            _routing["latent"] = torch.cat(lat) if lat else torch.empty((0, D), device=eng.device)
            _routing["assignments"] = torch.cat(asg) if asg else torch.empty((0, K), device=eng.device)
        lay = self._layout
        onum_all = torch.cat(nums) if nums else torch.empty((0, max(lay.n_num, 1)), device=eng.device)
        ocodes_all = torch.cat(cods) if cods else torch.empty((0, max(lay.n_disc, 1)), dtype=torch.int16, device=eng.device)
        recon = self._frame(onum_all, ocodes_all)
        return self._gathered(self._merge_observed(X, aligned, recon, mask.cpu().numpy(), only_missing), gather, row_offset)

    # ------------------------------------------------------------------
    def _ensure_classifier(self, what: str) -> None:
        if self.behaviour == "unsupervised":
            raise RuntimeError(f"{what} is unavailable when behaviour='unsupervised': the classifier head is disabled")
        if not self._is_fitted or self._classifier is None or self._classes is None:
            raise RuntimeError(f"Model must be fitted before calling {what}")

    def _compute_logits(self, X: pd.DataFrame) -> np.ndarray:
        """_compute_logits (model.py:3449-3486): classifier on the posterior-mean latent."""

        table = self._table(X.loc[:, list(self.schema.feature_names)])
        if table.n_rows == 0:
            return np.empty((0, len(self._classes)), np.float32)
        self._classifier.eval()
        with torch.no_grad():
            return self._classifier(self._posterior_mean(table)).cpu().numpy()

    def _resolve_attribute(self, attr):
        names = list(self.schema.feature_names)
        if isinstance(attr, (int, np.integer)):
            if not 0 <= int(attr) < len(names):
                raise ValueError(f"Attribute index {attr} is out of range")
            name = names[int(attr)]
        else:
            if attr not in self.schema:
                raise ValueError(f"Unknown attribute '{attr}'")
            name = str(attr)
        return name, self.schema[name], names.index(name)

    def _masked_attr_table(self, X: pd.DataFrame, attr_name: str, mask) -> PackedTable:
        """_normalize_inputs (model.py:3257-3302): the attribute is always decoded as missing; ``mask`` (True =
        missing) overrides the NaN-derived pattern."""

        names = list(self.schema.feature_names)
        aligned = X.loc[:, names].reset_index(drop=True)
        xnum, codes, m = self._pack(aligned)
        dev = self._engine.device
        if mask is not None:
            mm = np.asarray(mask.loc[:, names] if isinstance(mask, pd.DataFrame) else mask, dtype=bool)
            m = m * (~torch.from_numpy(np.ascontiguousarray(mm)).to(dev)).to(torch.uint8)
        m[:, names.index(attr_name)] = 0
        is_num = [self.schema[nme].type in ("real", "pos", "count") for nme in names]
        num_idx = torch.tensor([jj for jj, f in enumerate(is_num) if f], dtype=torch.long, device=dev)
        disc_idx = torch.tensor([jj for jj, f in enumerate(is_num) if not f], dtype=torch.long, device=dev)
        xnum = xnum.masked_fill(m[:, num_idx] == 0, 0.0)      # masked cells are zero-filled / coded -1 (Q7)
        codes = codes.masked_fill(m[:, disc_idx] == 0, -1)
        return PackedTable(xnum, codes, m, dev)

    def predict_proba(self, X: pd.DataFrame, attr=None, *, mask=None, L: int = 100, seed: Optional[int] = None,
                      row_offset: Optional[int] = None, gather: bool = False):
        """attr=None: calibrated classifier probabilities (model.py:3797-3815).  Otherwise the Monte-Carlo posterior
        predictive of a categorical / ordinal attribute (model.py:3816-3867); ``row_offset`` / ``gather`` as in
        ``sample`` (rows sharded over the ranks of torch.distributed, draws keyed by the global row)."""

        if attr is None:
            self._ensure_classifier("predict_proba")
            logits = self._compute_logits(X)
            if self._is_calibrated:
                logits = self._temperature_scaler.transform(logits)
            logits = np.asarray(logits, np.float32)
            e = np.exp(logits - logits.max(axis=1, keepdims=True)) if logits.size else logits
            return e / e.sum(axis=1, keepdims=True) if logits.size else logits
        self._require_fitted("predict_proba")
        attr_name, spec, j = self._resolve_attribute(attr)
        if spec.type not in {"cat", "ordinal"}:
            raise ValueError("predict_proba is only defined for categorical or ordinal attributes")
        if L <= 0:
            raise ValueError("L must be a positive integer")
        eng = self._engine
        seed = self.random_state if seed is None else int(seed)
        g0, X, mask = self._rows_here(X, row_offset, mask)  # this rank's rows; keys use the global row index
        table = self._masked_attr_table(X, attr_name, mask)
        n_classes = int(spec.n_classes)
        outs = [torch.empty((0, n_classes), dtype=torch.float32)]
        chunk = max(1, (1 << 20) // int(L))  # at most ~1M decoded rows at a time, as predict_confidence_interval
        for s0 in range(0, table.n_rows, chunk):
            nb = min(chunk, table.n_rows - s0)
            enc = eng.encode(table, s0, nb)["enc_out"]
            z, comp = eng.posterior_draws(enc, int(L), seed=seed, row_offset=g0 + s0)
            probs = eng.decode_params(z, j, component=comp)
            probs = probs.view(int(L), nb, -1).mean(dim=0).clamp(min=1e-8)
            outs.append((probs / probs.sum(dim=-1, keepdim=True).clamp_min(1e-8)).cpu())
        return self._gathered(torch.cat(outs), gather, row_offset)

    def predict_confidence_interval(self, X: pd.DataFrame, attr, *, mask=None, L: int = 1000, ci: float = 0.95,
                                    statistic: Optional[str] = None, return_samples: bool = False,
                                    seed: Optional[int] = None, row_offset: Optional[int] = None,
                                    gather: bool = False) -> Dict[str, torch.Tensor]:
        """Monte-Carlo posterior predictive statistics of a real / pos / count attribute (model.py:3869-4013):
        L draws of (s, z) ~ q(.|x_obs), one draw of the attribute's likelihood per latent draw."""

        self._require_fitted("predict_confidence_interval")
        attr_name, spec, j = self._resolve_attribute(attr)
        if spec.type in {"cat", "ordinal"}:
            raise ValueError("predict_confidence_interval is only available for real, pos or count attributes")
        if L <= 0:
            raise ValueError("L must be a positive integer")
        if not (0.0 < ci < 1.0):
            raise ValueError("ci must lie strictly between 0 and 1")
        if statistic not in {None, "mean", "median"}:
            raise ValueError("statistic must be one of None, 'mean' or 'median'")
        eng = self._engine
        seed = self.random_state if seed is None else int(seed)
        g0, X, mask = self._rows_here(X, row_offset, mask)
        table = self._masked_attr_table(X, attr_name, mask)
        B = table.n_rows
        if B == 0:
            empty = torch.empty(0, dtype=torch.float32)
            out = {"point": empty, "lower": empty, "upper": empty, "std": empty}
            if return_samples:
                out["samples"] = empty.view(0, 0 if not parallel.active() else int(L))
            return self._gathered(out, gather, row_offset)
        i_num = [n for n in self.schema.feature_names if self.schema[n].type in ("real", "pos", "count")].index(attr_name)
        samples = torch.empty((int(L), B), dtype=torch.float32, device=eng.device)
        chunk = max(1, (1 << 20) // int(L))  # at most ~1M decoded rows at a time
        for s0 in range(0, B, chunk):
            nb = min(chunk, B - s0)
            enc = eng.encode(table, s0, nb)["enc_out"]
            z, comp = self._mc_draws(enc, int(L), seed, g0 + s0)
            onum, _ = eng.decode_sample(z, component=comp, seed=seed + 1, row_offset=(g0 + s0) * int(L))
            col = onum[:, i_num].view(nb, int(L)).transpose(0, 1)
            if spec.type == "count":  # decoder_outputs_to_frame rounds counts (sampling.py:160-163)
                col = col.round().clamp(min=0)
            samples[:, s0:s0 + nb] = col
        lo_q = (1.0 - ci) / 2.0
        lower = torch.quantile(samples, lo_q, dim=0)
        upper = torch.quantile(samples, 1.0 - lo_q, dim=0)
        if statistic == "mean" or (statistic is None and spec.type in {"real", "count"}):
            point = samples.mean(dim=0)
        else:
            point = samples.median(dim=0).values
        out = {"point": point.cpu(), "lower": lower.cpu(), "upper": upper.cpu(), "std": samples.std(dim=0, unbiased=False).cpu()}
        if return_samples:
            out["samples"] = samples.transpose(0, 1).contiguous().cpu()
        return self._gathered(out, gather, row_offset)

    def predict(self, X: pd.DataFrame, attr=None, *, mask=None, L: int = 50, mode: str = "point",
                seed: Optional[int] = None):
        """Class labels (attr=None, model.py:4062-4083) or attribute predictions (model.py:4085-4117)."""

        if attr is None:
            probs = self.predict_proba(X)
            return self._classes[np.argmax(probs, axis=1)] if len(probs) else np.empty(0, dtype=self._classes.dtype)
        attr_name, spec, _ = self._resolve_attribute(attr)
        if mode not in {"point", "sample"}:
            raise ValueError("mode must be either 'point' or 'sample'")
        seed = self.random_state if seed is None else int(seed)
        if spec.type in {"cat", "ordinal"}:
            probs = self.predict_proba(X, attr=attr_name, mask=mask, L=L, seed=seed)
            if probs.numel() == 0:
                return probs.new_empty((0,), dtype=torch.long)
            if mode == "point":
                return probs.argmax(dim=1)
            g = torch.Generator()
            g.manual_seed(seed)
            return torch.multinomial(probs, 1, generator=g).squeeze(1)
        stats = self.predict_confidence_interval(X, attr_name, mask=mask, L=max(1, L), statistic=None,
                                                 return_samples=(mode == "sample"), seed=seed)
        if mode == "point" or stats.get("samples") is None or stats["samples"].size(1) == 0:
            return stats["point"]
        g = torch.Generator()
        g.manual_seed(seed)
        pick = torch.randint(0, stats["samples"].size(1), (stats["samples"].size(0),), generator=g)
        return stats["samples"].gather(1, pick.view(-1, 1)).squeeze(1)

    def calibrate(self, X: pd.DataFrame, y) -> "SUAVE":
        """Temperature scaling on the classifier logits of ``X`` (model.py:4122-4169)."""

        self._ensure_classifier("calibrate")
        if len(X) != len(y):
            raise ValueError("X and y must have matching first dimensions")
        logits = self._compute_logits(X)
        self._temperature_scaler.fit(logits, self._map_targets(np.asarray(y)))
        self._is_calibrated = True
        return self

    # ------------------------------------------------------------------
    def _mc_draws(self, enc: torch.Tensor, L: int, seed: int, first_row: int):
        """L posterior draws per row of ``enc`` in ROW-major order (row b's draws are rows b L .. b L + L - 1): with
        it the decode's generator key of (row, draw) is (global row) * L + draw, whatever the chunking or sharding."""

        return self._engine.posterior_draws(enc, L, seed=seed, row_offset=first_row, row_major=True)

    def impute_mc(self, X: pd.DataFrame, *, L: int = 32, only_missing: bool = True, seed: Optional[int] = None,
                  row_offset: Optional[int] = None, gather: bool = False) -> pd.DataFrame:
        """Posterior-predictive Monte-Carlo imputation (BASELINE config #5): ``L`` draws of (s, z) ~ q(.|x_obs) per
        row, every cell decoded per draw; missing numeric cells get the mean over the draws (counts rounded),
        missing categorical / ordinal cells the most frequent drawn level.  This generalises the per-attribute MC
        path (model.py:3304-3421) to all attributes; the reference's own ``impute`` is the single pass above."""

        self._require_fitted("impute_mc")
        if L <= 0:
            raise ValueError("L must be a positive integer")
        eng = self._engine
        seed = self.random_state if seed is None else int(seed)
        names = list(self.schema.feature_names)
        g0, X = self._rows_here(X, row_offset)
        aligned = X.loc[:, names].reset_index(drop=True)
        xnum, codes, mask = self._pack(aligned)
        table = PackedTable(xnum, codes, mask, eng.device)
        B, L = table.n_rows, int(L)
        lay = self._layout
        onum_all = torch.empty((B, max(lay.n_num, 1)), device=eng.device)
        ocode_all = torch.empty((B, max(lay.n_disc, 1)), dtype=torch.int16, device=eng.device)
        is_count = torch.tensor([self.schema[n].type == "count" for n in lay.numeric_columns], device=eng.device)
        chunk = max(1, (1 << 20) // L)
        for s0 in range(0, B, chunk):
            nb = min(chunk, B - s0)
            enc = eng.encode(table, s0, nb)["enc_out"]
            z, comp = self._mc_draws(enc, L, seed, g0 + s0)
            onum, ocodes = eng.decode_sample(z, component=comp, seed=seed + 1, row_offset=(g0 + s0) * L)
            if lay.n_num:
                v = onum.view(nb, L, -1)
                v = torch.where(is_count, v.round().clamp(min=0), v)
                onum_all[s0:s0 + nb, : lay.n_num] = v.mean(dim=1)
            if lay.n_disc:
                ocode_all[s0:s0 + nb, : lay.n_disc] = torch.mode(ocodes.view(nb, L, -1), dim=1).values
        recon = self._frame(onum_all, ocode_all)
        return self._gathered(self._merge_observed(X, aligned, recon, mask.cpu().numpy(), only_missing), gather, row_offset)

    def _merge_observed(self, X, aligned, recon, mask, only_missing):
        if only_missing:
            imputed = recon.copy()
            for j, column in enumerate(recon.columns):
                obs = mask[:, j].astype(bool)
                if not obs.any():
                    continue
                vals = aligned.loc[obs, column]
                if isinstance(recon[column].dtype, pd.CategoricalDtype):
                    vals = vals.astype(recon[column].dtype)
                else:
                    try:
                        vals = vals.astype(recon[column].dtype, copy=False)
                    except (TypeError, ValueError):
                        vals = vals.to_numpy()
                imputed.loc[obs, column] = vals
        else:
            imputed = recon
        imputed.index = X.index
        extra = [c for c in X.columns if c not in self.schema.feature_names]
        if extra:
            imputed = pd.concat([imputed, X.loc[:, extra]], axis=1)
        return imputed.loc[:, X.columns]

    # ------------------------------------------------------------------
    def _payload(self) -> Dict[str, object]:
        """The reference's archive payload (model.py:4484-4578) for this model: same key tree, same state_dict key
        names, so `suave.SUAVE.load` reads an archive written here and `SUAVE.load` here reads the reference's."""

        from . import persist

        unsup = self.behaviour == "unsupervised"
        meta = dict(behaviour=self.behaviour, latent_dim=self.latent_dim, n_components=self.n_components, beta=self.beta,
                    hidden_dims=list(self.hidden_dims), head_hidden_dims=list(self.head_hidden_dims), dropout=self.dropout,
                    learning_rate=self.learning_rate, batch_size=self.batch_size, kl_warmup_epochs=self.kl_warmup_epochs,
                    val_split=self.val_split, stratify=self.stratify, random_state=self.random_state,
                    gumbel_temperature=self.gumbel_temperature, tau_start=self._tau_start if unsup else None,
                    tau_min=self._tau_min if unsup else None, tau_decay=self._tau_decay if unsup else None,  # model.py:476-492
                    inference_tau=self._inference_tau, warmup_epochs=self.warmup_epochs,
                    head_epochs=self.head_epochs, finetune_epochs=self.finetune_epochs,
                    decoder_refine_epochs=self.decoder_refine_epochs, decoder_refine_mode=self.decoder_refine_mode,
                    decoder_refine_prior_lambda=self.decoder_refine_prior_lambda,
                    joint_decoder_lr_scale=self.joint_decoder_lr_scale, early_stop_patience=self.early_stop_patience,
                    classification_loss_weight=self.classification_loss_weight,
                    auto_configured={**self._auto_configured,
                                     "classification_loss_weight": not self._classification_loss_weight_user},
                    auto_hyperparameters=None)
        clf_state = class_weight = None
        use_weight = False
        if self._classifier is not None:
            clf_state = {k: v.detach().cpu() for k, v in self._classifier.state_dict().items()}
            use_weight = bool(self._classifier._use_weight)
            class_weight = self._classifier.class_weight.detach().cpu()
        return persist.build_payload(
            metadata=meta, state=self._engine.state(), schema=self.schema, classifier_state=clf_state,
            class_weight=class_weight, use_weight=use_weight, temperature_scaler=self._temperature_scaler.state_dict(),
            norm_stats=self._norm_stats_per_col, classes=self._classes, class_to_index=self._class_to_index,
            is_calibrated=self._is_calibrated, train_cache=self._train_cache,
            train_target_indices=getattr(self, "_train_target_indices", None),
            warmup_val_history=self._warmup_val_history, joint_val_metrics=self._joint_val_metrics)

    def save(self, path):
        """Serialise the model in the reference's archive layout (model.py:4456-4579)."""

        from pathlib import Path

        self._require_fitted("saving")
        path = Path(path)
        torch.save(self._payload(), path)
        return path

    @classmethod
    def load(cls, path, *, device: Optional[str] = None) -> "SUAVE":
        """Restore a model from an archive in the reference's layout (model.py:4581-4950), written by either
        implementation.  The file is read with ``weights_only=True`` (persist.load_archive)."""

        from . import persist

        payload = persist.load_archive(path)
        if not isinstance(payload, dict) or "metadata" not in payload:
            raise ValueError("Unexpected model archive format")
        return cls._load_from_payload(payload, device=device)

    @classmethod
    def _load_from_payload(cls, payload, *, device: Optional[str] = None) -> "SUAVE":
        from . import persist

        meta, mods, art = payload.get("metadata", {}), payload.get("modules", {}), payload.get("artefacts", {})
        if not meta.get("schema"):
            raise ValueError("Schema is required to restore decoder state")
        kwargs = {}
        for key in ("latent_dim", "n_components", "beta", "hidden_dims", "head_hidden_dims", "classification_loss_weight",
                    "dropout", "learning_rate", "batch_size", "kl_warmup_epochs", "val_split", "stratify", "random_state",
                    "gumbel_temperature", "tau_start", "tau_min", "tau_decay", "warmup_epochs", "head_epochs",
                    "finetune_epochs", "decoder_refine_epochs", "decoder_refine_mode", "decoder_refine_prior_lambda",
                    "joint_decoder_lr_scale", "early_stop_patience"):  # model.py:4695-4751
            if meta.get(key) is not None:
                kwargs[key] = meta[key]
        model = cls(Schema(meta["schema"]), behaviour=str(meta.get("behaviour", "supervised")).lower(), device=device,
                    **kwargs)
        auto = meta.get("heuristic_overrides")
        if not isinstance(auto, dict):
            auto = meta.get("auto_configured")
        if isinstance(auto, dict):
            model._auto_configured = {k: bool(v) for k, v in auto.items()}
            model._classification_loss_weight_user = not bool(auto.get("classification_loss_weight", False))
        if meta.get("inference_tau") is not None:
            model._inference_tau = float(meta["inference_tau"])
        model._norm_stats_per_col = art.get("normalization", {})
        model._layout = ModelLayout(model.schema, model.n_components, model.latent_dim, model.hidden_dims, model.dropout,
                                    model.behaviour, model._norm_stats_per_col)
        model._engine = eng = Engine(model._layout, model._select_device())
        eng.load_state(persist.state_from_payload(payload))
        model._class_to_index = art.get("class_to_index")
        if art.get("classes") is not None:
            model._classes = np.asarray(art["classes"])
            model.classes_ = model._classes
            if model._class_to_index is None:
                model._class_to_index = {c.item() if isinstance(c, np.generic) else c: i
                                         for i, c in enumerate(model._classes)}
        if art.get("train_target_indices") is not None:
            tt = art["train_target_indices"]
            model._train_target_indices = np.asarray(tt.numpy() if isinstance(tt, torch.Tensor) else tt)
        clf = mods.get("classifier")
        if model.behaviour == "supervised" and clf:
            if model._classes is None:
                raise ValueError("Saved model is missing class labels")
            cw = clf.get("class_weight") if clf.get("use_weight") else None
            model._classifier = sup.ClassificationHead(model.latent_dim, int(model._classes.size),
                                                       class_weight=None if cw is None else np.asarray(cw, np.float32),
                                                       dropout=model.dropout, hidden_dims=model.head_hidden_dims).to(eng.device)
            if clf.get("state_dict") is not None:
                model._classifier.load_state_dict(dict(clf["state_dict"]))
            model._classifier.eval()
        tstate = mods.get("temperature_scaler") or {}
        model._temperature_scaler.load_state_dict(tstate)
        model._is_calibrated = bool(tstate.get("fitted", False))  # model.py:4889-4891
        cache = {"probs": art.get("train_component_probs"), "mu": art.get("train_component_mu"),
                 "logvar": art.get("train_component_logvar"), "mean": art.get("train_latent_mu"),
                 "post_logvar": art.get("train_latent_logvar"), "logits": art.get("train_component_logits")}
        model._train_cache = {k: torch.as_tensor(v).float().contiguous().to(eng.device) for k, v in cache.items()
                              if v is not None}
        model._warmup_val_history = list(art.get("warmup_val_history") or [])
        model._joint_val_metrics = art.get("joint_val_metrics")
        model._is_fitted = True
        return model
