"""Supervised extras around the HIP ELBO step: classification head, temperature
scaling and the validation metrics the training schedule compares.

These are the small pieces SURVEY.md §2 rows 16-17 keep on stock PyTorch-ROCm
(a ``[B, D] x [D, C]`` linear and a one-parameter fit): the head consumes the
``latent`` output of ``suave_elbo_forward`` and hands ``d loss / d latent``
back to ``suave_elbo_backward`` (``dlatent``), so the encoder / decoder math of
joint fine-tuning stays in the HIP library.

Reference: ``suave/modules/heads.py`` (ClassificationHead),
``suave/modules/calibrate.py`` (TemperatureScaler), ``suave/evaluate.py:88-297``
(AUROC / Brier / ECE as used by ``model.py:2609-2622``).
"""

from __future__ import annotations

import math
from typing import Dict, Iterable, Optional

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F


class RowKeyedDropout(nn.Module):
    """nn.Dropout whose keep decisions can be keyed by the TABLE ROW of each batch row instead of torch's generator:
    a stateless hash of (key, layer, row id, feature).  With keys a row draws the same mask on whichever rank
    computes it, so a data-parallel fit reproduces the one-GPU fit's head dropout (the reference has one device and
    one generator stream: suave/modules/heads.py:47-49).  Without keys (``ctx`` None) it is ``F.dropout``."""

    def __init__(self, p: float, layer: int = 0) -> None:
        super().__init__()
        self.p, self.layer = float(p), int(layer)
        self.ctx = None  # (row ids int64 [B], key int), set around a forward by ClassificationHead

    @staticmethod
    def _h32(x: torch.Tensor) -> torch.Tensor:  # lowbias32 on int64 lanes holding uint32 values
        m = 0xFFFFFFFF
        x = x ^ (x >> 16)
        x = (x * 0x7FEB352D) & m
        x = x ^ (x >> 15)
        x = (x * 0x846CA68B) & m
        return x ^ (x >> 16)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.training or self.p <= 0.0:
            return x
        if self.ctx is None:
            return F.dropout(x, self.p, True)
        rows, key = self.ctx
        m = 0xFFFFFFFF
        r = self._h32((rows.to(torch.int64) * 0x9E3779B1 + (int(key) & m) + self.layer * 0x85EBCA77) & m)
        cols = (torch.arange(x.shape[-1], device=x.device, dtype=torch.int64) * 0x632BE5AB + 0x165667B1) & m
        h = self._h32(r.unsqueeze(-1) ^ cols.unsqueeze(0))
        keep = ((h >> 8).to(torch.float32) * (1.0 / 16777216.0)) >= self.p
        return x * keep.to(x.dtype) * (1.0 / (1.0 - self.p))


class ClassificationHead(nn.Module):
    """heads.py:12-91: optional ``Linear -> ReLU -> Dropout`` stack, then ``linear``; weighted cross-entropy.
    Parameter names match the reference ``state_dict`` (``hidden_layers.<i>.*``, ``linear.*``, ``class_weight``)."""

    def __init__(self, in_features: int, n_classes: int, *, class_weight=None, dropout: float = 0.0,
                 hidden_dims: Iterable[int] = ()) -> None:
        super().__init__()
        if n_classes < 2:
            raise ValueError("Classification head requires at least two classes")
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError("dropout must lie in the interval [0, 1)")
        hidden_dims = tuple(int(d) for d in hidden_dims)
        if any(d <= 0 for d in hidden_dims):
            raise ValueError("hidden_dims must contain positive integers")
        layers = []
        prev = int(in_features)
        for h in hidden_dims:
            layers += [nn.Linear(prev, h), nn.ReLU()]
            if dropout:
                layers.append(RowKeyedDropout(float(dropout), layer=len(layers)))
            prev = h
        self.hidden_layers = nn.Sequential(*layers) if layers else None
        self._final_dropout = RowKeyedDropout(float(dropout)) if dropout and not hidden_dims else None
        self.linear = nn.Linear(prev, n_classes)
        if class_weight is not None:
            w = torch.as_tensor(np.asarray(class_weight), dtype=torch.float32)
            if w.numel() != n_classes:
                raise ValueError("class_weight must provide a value for every target class")
            self.register_buffer("class_weight", w)
            self._use_weight = True
        else:
            self.register_buffer("class_weight", torch.ones(n_classes))
            self._use_weight = False

    def forward(self, latents: torch.Tensor, rows: Optional[torch.Tensor] = None, key: int = 0) -> torch.Tensor:
        """``rows`` (table row id of each batch row) + ``key`` (seed / step) key the dropout masks by row; without
        them the masks come from torch's generator as in the reference."""

        drops = [mod for mod in self.modules() if isinstance(mod, RowKeyedDropout)]
        for d in drops:
            d.ctx = None if rows is None else (rows, key)
        try:
            if self.hidden_layers is not None:
                latents = self.hidden_layers(latents)
            elif self._final_dropout is not None:
                latents = self._final_dropout(latents)
            return self.linear(latents)
        finally:
            for d in drops:
                d.ctx = None

    def loss(self, logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        return F.cross_entropy(logits, targets, weight=self.class_weight if self._use_weight else None)

    # the same loss cut for row-sharded batches: loss == loss_terms(all rows) / loss_weight(all targets)
    def loss_terms(self, logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        return F.cross_entropy(logits, targets, weight=self.class_weight if self._use_weight else None, reduction="sum")

    def loss_weight(self, targets: torch.Tensor) -> torch.Tensor:
        if self._use_weight:
            return self.class_weight[targets].sum()
        return torch.tensor(float(targets.numel()), device=targets.device)


class DeviceHead:
    """Library-side training state of a ``ClassificationHead`` WITHOUT hidden layers (include/suave_hip.h,
    ``suave_head_t``): the module's ``linear`` parameters are re-homed as views of one flat device buffer
    (weight [C, D] then bias [C]) which the HIP kernels read and update in place, so ``state_dict`` / persistence /
    the eval-mode forward keep working on the module while the head phase (model.py:1743-1864) and the joint step
    (model.py:1973-1999) run in the library: no autograd, no optimizer object, no host sync per batch."""

    def __init__(self, clf: ClassificationHead, targets: torch.Tensor) -> None:
        import ctypes as C

        from . import lib as _lib

        if clf.hidden_layers is not None:
            raise ValueError("DeviceHead serves heads without hidden layers")
        w, b = clf.linear.weight, clf.linear.bias
        self.C, self.D = int(w.shape[0]), int(w.shape[1])
        dev = w.device
        self.flat = torch.cat([w.detach().reshape(-1), b.detach().reshape(-1)]).to(torch.float32).contiguous()
        w.data = self.flat[: self.C * self.D].view(self.C, self.D)
        b.data = self.flat[self.C * self.D:]
        self.clf = clf
        self.grads = torch.zeros_like(self.flat)
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.t = 0
        nbytes = C.c_size_t(0)
        _lib.check(_lib.load().suave_head_workspace_bytes(self.C, self.D, C.byref(nbytes)), "suave_head_workspace_bytes")
        self.ws = torch.empty(int(nbytes.value), dtype=torch.uint8, device=dev)
        self.loss_sum = torch.zeros(1, dtype=torch.float64, device=dev)
        self.class_weight = clf.class_weight.to(torch.float32).contiguous() if clf._use_weight else None
        self.targets = targets.to(device=dev, dtype=torch.int32).contiguous()
        drops = [mod for mod in clf.modules() if isinstance(mod, RowKeyedDropout)]
        self.dropout = float(drops[0].p) if drops else 0.0

    @staticmethod
    def supports(clf: ClassificationHead) -> bool:
        w = clf.linear.weight
        c, d = int(w.shape[0]), int(w.shape[1])
        lds = 4 * (2 * (c * d + c) + d * 65 + c * 64 + 2)  # k_head_loss at its smallest tile (64 rows)
        return clf.hidden_layers is None and 2 <= c <= 64 and 1 <= d <= 256 and lds <= 144 * 1024 and w.is_cuda

    def reset_optimizer(self) -> None:
        self.m.zero_()
        self.v.zero_()
        self.t = 0

    def struct(self, *, training: bool, loss_scale: float, loss_rows: float, key: int = 0,
               weight_sum: Optional[torch.Tensor] = None, adam_lr: float = 0.0, want_grads: bool = True,
               targets: Optional[torch.Tensor] = None):
        """A ``suave_head_t`` for one call.  ``adam_lr`` > 0 advances the head's Adam step counter.  ``targets``
        (int32, device) replaces the training targets (validation rows)."""

        from .plan import HeadT

        if adam_lr > 0.0:
            self.t += 1
        h = HeadT()
        h.n_classes, h.training = self.C, int(bool(training))
        h.params, h.grads = self.flat.data_ptr(), (self.grads.data_ptr() if want_grads else None)
        h.class_weight = None if self.class_weight is None else self.class_weight.data_ptr()
        h.targets = (self.targets if targets is None else targets).data_ptr()
        h.weight_sum = None if weight_sum is None else weight_sum.data_ptr()
        h.loss_scale, h.loss_rows = float(loss_scale), float(loss_rows)
        h.loss_sum = self.loss_sum.data_ptr()
        h.dropout, h.dropout_key = self.dropout, int(key) & 0xFFFFFFFF
        h.adam_lr, h.adam_step = float(adam_lr), int(self.t)
        h.adam_m, h.adam_v = self.m.data_ptr(), self.v.data_ptr()
        h.workspace, h.workspace_bytes = self.ws.data_ptr(), int(self.ws.numel())
        return h

    def evaluate(self, engine, latent: torch.Tensor, targets: torch.Tensor):
        """Eval-mode logits [n, C] and the class-weighted cross-entropy (a device scalar, float64) of ``latent`` [n, D]
        against ``targets`` [n] -- the library's head kernel instead of a torch linear (no BLAS handle, one launch pair)."""

        n = int(latent.shape[0])
        latent = latent.contiguous()
        t32 = targets.to(torch.int32).contiguous()
        rows = torch.arange(n, dtype=torch.int32, device=latent.device)
        logits = torch.empty((n, self.C), dtype=torch.float32, device=latent.device)
        keep = self.loss_sum.clone()
        self.loss_sum.zero_()
        engine.head_step(self.struct(training=False, loss_scale=1.0, loss_rows=1.0, want_grads=False, targets=t32), latent, rows,
                         logits=logits)
        ce = self.loss_sum.clone()
        self.loss_sum.copy_(keep)
        return logits, ce

    def batch_weight(self, rows: torch.Tensor) -> torch.Tensor:
        """Sum of the class weights over the targets of ``rows`` (the GLOBAL batch of a data-parallel step): a device scalar."""

        if self.class_weight is None:
            return torch.full((1,), float(rows.numel()), dtype=torch.float32, device=self.flat.device)
        return self.class_weight[self.targets[rows].long()].sum().reshape(1)


class TemperatureScaler:
    """calibrate.py:16-112: one log-temperature fitted by Adam on the cross-entropy of ``logits / T``."""

    def __init__(self, temperature: float = 1.0, fitted: bool = False) -> None:
        self.temperature = float(temperature)
        self.fitted = bool(fitted)

    def fit(self, logits, targets, *, max_iter: int = 100, learning_rate: float = 0.01) -> "TemperatureScaler":
        logits = np.asarray(logits, dtype=np.float32)
        targets = np.asarray(targets, dtype=np.int64)
        if logits.ndim != 2:
            raise ValueError("logits must be a 2D array")
        if logits.shape[0] != targets.shape[0]:
            raise ValueError("logits and targets must have matching rows")
        if targets.size == 0:
            raise ValueError("calibration targets must be non-empty")
        lt, tt = torch.from_numpy(logits), torch.from_numpy(targets)
        log_t = torch.tensor(math.log(max(self.temperature, 1e-3)), dtype=lt.dtype, requires_grad=True)
        opt = torch.optim.Adam([log_t], lr=learning_rate)
        for _ in range(max_iter):
            opt.zero_grad()
            loss = F.cross_entropy(lt / torch.exp(log_t), tt)
            if not torch.isfinite(loss):
                break
            loss.backward()
            opt.step()
        self.temperature = float(torch.exp(log_t.detach()).clamp(min=1e-3, max=1e3).item())
        self.fitted = True
        return self

    def transform(self, logits):
        if self.temperature <= 0:
            raise ValueError("temperature must be positive")
        if isinstance(logits, torch.Tensor):
            return logits / float(self.temperature)
        return np.asarray(logits, dtype=np.float32) / float(self.temperature)

    __call__ = transform

    def state_dict(self) -> Dict[str, object]:
        return {"temperature": float(self.temperature), "fitted": bool(self.fitted)}

    def load_state_dict(self, state) -> None:
        t = float(state.get("temperature", 1.0))
        if t <= 0:
            raise ValueError("Loaded temperature must be positive")
        self.temperature = t
        self.fitted = bool(state.get("fitted", False))


# ---------------------------------------------------------------------------
# validation metrics (evaluate.py:88-297 semantics: binary -> positive column, multi-class -> macro OvR)
# ---------------------------------------------------------------------------
def _auc_binary(score: np.ndarray, y: np.ndarray) -> float:
    pos, neg = int((y == 1).sum()), int((y == 0).sum())
    if pos == 0 or neg == 0:
        raise ValueError("AUROC needs both classes")
    order = np.argsort(score, kind="mergesort")
    s = score[order]
    ranks = np.empty(len(s), np.float64)
    i = 0
    while i < len(s):  # average ranks over ties
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty(len(s), np.float64)
    r[order] = ranks
    return float((r[y == 1].sum() - pos * (pos + 1) / 2.0) / (pos * neg))


def compute_auroc(probabilities: np.ndarray, targets: np.ndarray) -> float:
    p = np.asarray(probabilities, np.float64)
    t = np.asarray(targets).astype(np.int64).reshape(-1)
    if p.ndim == 1 or p.shape[1] == 1:
        return _auc_binary(p.reshape(-1), t)
    if p.shape[1] == 2:
        return _auc_binary(p[:, 1], t)
    vals = []
    for c in range(p.shape[1]):
        if (t == c).any() and (t != c).any():
            vals.append(_auc_binary(p[:, c], (t == c).astype(np.int64)))
    if not vals:
        raise ValueError("AUROC undefined")
    return float(np.mean(vals))


def compute_brier(probabilities: np.ndarray, targets: np.ndarray) -> float:
    p = np.asarray(probabilities, np.float64)
    t = np.asarray(targets).astype(np.int64).reshape(-1)
    if p.ndim == 2 and p.shape[1] == 2:
        return float(np.mean((p[:, 1] - t) ** 2))
    onehot = np.eye(p.shape[1])[t]
    return float(np.mean(np.sum((p - onehot) ** 2, axis=1)))


def compute_ece(probabilities: np.ndarray, targets: np.ndarray, n_bins: int = 10) -> float:
    p = np.asarray(probabilities, np.float64)
    t = np.asarray(targets).astype(np.int64).reshape(-1)
    if p.ndim == 2 and p.shape[1] == 2:
        conf, correct = p[:, 1], (t == 1).astype(np.float64)
    else:
        conf, correct = p.max(axis=1), (p.argmax(axis=1) == t).astype(np.float64)
    edges = np.linspace(0.0, 1.0, n_bins + 1)
    ece = 0.0
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (conf > lo) & (conf <= hi) if lo > 0 else (conf >= lo) & (conf <= hi)
        if sel.any():
            ece += sel.mean() * abs(correct[sel].mean() - conf[sel].mean())
    return float(ece)


def device_classification_metrics(probabilities: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """``[brier, ece, auroc]`` (float64, on the tensors' device) of ``compute_brier`` / ``compute_ece`` /
    ``compute_auroc`` above -- the validation metrics of model.py:2609-2622 (evaluate.py:88-297) without moving the
    probabilities to the host: squared errors and calibration bins are reductions, AUROC is the rank statistic with
    average ranks over ties (one sort per class; ``unique_consecutive`` gives the tie groups).  An undefined AUROC
    (one class only) is NaN, where the host function raises.  One small tensor: one sync for the caller."""

    p = probabilities.detach().to(torch.float64)
    t = targets.reshape(-1).to(torch.int64)
    n, C = int(p.shape[0]), int(p.shape[1])
    nan = torch.full((), float("nan"), dtype=torch.float64, device=p.device)
    if n == 0:
        return torch.stack([nan, nan, nan])
    binary = C == 2
    if binary:
        brier = ((p[:, 1] - t.to(torch.float64)) ** 2).mean()
        conf, correct = p[:, 1], (t == 1).to(torch.float64)
    else:
        brier = ((p - F.one_hot(t, C).to(torch.float64)) ** 2).sum(dim=1).mean()
        conf, arg = p.max(dim=1)
        correct = (arg == t).to(torch.float64)
    # bins (lo, hi] (the first one closed at 0): bucketize against the nine inner edges of np.linspace(0, 1, 11)
    inner = torch.from_numpy(np.linspace(0.0, 1.0, 11)[1:-1].copy()).to(p.device)
    idx = torch.bucketize(conf.contiguous(), inner, right=False)
    cnt = torch.bincount(idx, minlength=10).to(torch.float64)
    s_conf = torch.bincount(idx, weights=conf, minlength=10)
    s_corr = torch.bincount(idx, weights=correct, minlength=10)
    safe = cnt.clamp_min(1.0)
    ece = torch.where(cnt > 0, (cnt / n) * (s_corr / safe - s_conf / safe).abs(), torch.zeros_like(cnt)).sum()

    def auc(score: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        pos = y.sum().to(torch.float64)
        neg = float(n) - pos
        srt, order = torch.sort(score)
        _, inv, counts = torch.unique_consecutive(srt, return_inverse=True, return_counts=True)
        cd = counts.to(torch.float64)
        avg_rank = cd.cumsum(0) - (cd - 1.0) * 0.5           # 1-based ranks, averaged over a tie group
        rsum = (avg_rank[inv] * y[order].to(torch.float64)).sum()
        val = (rsum - pos * (pos + 1.0) * 0.5) / (pos * neg).clamp_min(1.0)
        return torch.where((pos > 0) & (neg > 0), val, nan)

    if binary:
        auroc = auc(p[:, 1], t == 1)
    else:
        vals = torch.stack([auc(p[:, c], t == c) for c in range(C)])
        ok = ~torch.isnan(vals)
        auroc = torch.where(ok.any(), torch.where(ok, vals, torch.zeros_like(vals)).sum() / ok.sum().clamp_min(1), nan)
    return torch.stack([brier, ece, auroc])


def is_better_metrics(candidate: Dict[str, float], best: Dict[str, float]) -> bool:
    """model.py:2637-2678: lexicographic comparison joint_objective, nll, classification_loss, brier, ece (min), auroc (max)."""

    for key, mode in (("joint_objective", "min"), ("nll", "min"), ("classification_loss", "min"), ("brier", "min"),
                      ("ece", "min"), ("auroc", "max")):
        c, b = candidate.get(key), best.get(key)
        c = float(c) if c is not None else float("nan")
        b = float(b) if b is not None else float("nan")
        cf, bf = math.isfinite(c), math.isfinite(b)
        if not cf and not bf:
            continue
        if cf and not bf:
            return True
        if not cf and bf:
            return False
        if mode == "min":
            if c < b - 1e-8:
                return True
            if c > b + 1e-8:
                return False
        else:
            if c > b + 1e-8:
                return True
            if c < b - 1e-8:
                return False
    return False


def derive_classification_loss_weight(elbo_scale: Optional[float], ce_scale: Optional[float]) -> float:
    """model.py:634-664."""

    if (elbo_scale is None or not math.isfinite(float(elbo_scale)) or ce_scale is None
            or not math.isfinite(float(ce_scale)) or float(ce_scale) <= 0.0):
        return 1.0
    return float(np.clip(float(elbo_scale) / max(float(ce_scale), 1e-8), 0.1, 1000.0))
