"""``SUAVE``-compatible estimator driving the HIP engine.

Mirrors the public surface of the reference's ``suave/model.py`` for the hot
path: constructor kwargs (model.py:277-312), ``fit`` (:883-901), ``encode``
(:4174), ``sample`` (:4284), ``impute`` (:3512), ``predict_proba(attr=...)``
(:3720) -- with every device computation done by ``libsuave_hip.so``.

What is covered: the ELBO warm-up phase for both behaviours (the whole
unsupervised ``fit`` is this phase, model.py:1382-1560), per-epoch validation
ELBO (:2497-2559), posterior-statistics caches (:1669-1741, :2725-2761),
generation, imputation and the Monte-Carlo posterior predictive.  The
supervised classifier head / joint fine-tune / decoder-refine phases
(model.py:1743-2445) and temperature calibration are outside the hot path
 and are not implemented here.
"""

from __future__ import annotations

import math
from typing import Dict, Iterable, Literal, Optional

import numpy as np
import pandas as pd
import torch

from . import data as data_utils
from .engine import Engine, PackedTable
from .plan import ModelLayout
from .types import Schema

_DEFAULTS = dict(latent_dim=32, n_components=1, beta=1.5, hidden_dims=(256, 128), dropout=0.1, learning_rate=1e-3,
                 batch_size=128, kl_warmup_epochs=10, warmup_epochs=10, val_split=0.2, random_state=0,
                 gumbel_temperature=1.0)  # model.py:51-68


class SUAVE:
    def __init__(self, schema: Optional[Schema] = None, *, behaviour: Literal["supervised", "unsupervised"] = "supervised",
                 latent_dim: Optional[int] = None, n_components: int = 1, beta: float = 1.5,
                 hidden_dims: Optional[Iterable[int]] = None, dropout: Optional[float] = None,
                 learning_rate: Optional[float] = None, batch_size: Optional[int] = None,
                 warmup_epochs: Optional[int] = None, kl_warmup_epochs: Optional[int] = None,
                 val_split: float = 0.2, random_state: int = 0, gumbel_temperature: float = 1.0,
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

    # ------------------------------------------------------------------
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

    def _table(self, X: pd.DataFrame) -> PackedTable:
        xnum, codes, mask = data_utils.pack_frame(X.reset_index(drop=True), self.schema, self._norm_stats_per_col)
        return PackedTable(xnum, codes, mask, self._engine.device)

    def _require_fitted(self, what: str) -> None:
        if not self._is_fitted or self._engine is None:
            raise RuntimeError(f"Model must be fitted before calling {what}")

    # ------------------------------------------------------------------
    def fit(self, X: pd.DataFrame, y=None, *, schema: Optional[Schema] = None, epochs: Optional[int] = None,
            batch_size: Optional[int] = None, kl_warmup_epochs: Optional[int] = None,
            warmup_epochs: Optional[int] = None) -> "SUAVE":
        """ELBO warm-up training (the whole ``fit`` for behaviour="unsupervised")."""

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
        X = X.loc[:, list(self.schema.feature_names)].reset_index(drop=True)
        n = len(X)
        rng = np.random.default_rng(self.random_state)
        order = rng.permutation(n)
        n_val = int(round(n * self.val_split)) if n > 1 else 0  # split_train_val (data.py:17-104), unstratified
        n_val = min(max(n_val, 0), max(n - 1, 0))
        val_idx, train_idx = np.sort(order[:n_val]), np.sort(order[n_val:])
        X_train, X_val = X.iloc[train_idx], X.iloc[val_idx]
        self._norm_stats_per_col = data_utils.fit_norm_stats(X_train, self.schema)  # statistics from the train split
        if y is not None:
            y_arr = np.asarray(y).reshape(-1)
            self._classes = np.unique(y_arr)
            self._class_to_index = {c.item() if isinstance(c, np.generic) else c: i for i, c in enumerate(self._classes)}
            self._train_target_indices = np.asarray([self._class_to_index[v.item() if isinstance(v, np.generic) else v]
                                                     for v in y_arr[train_idx]], dtype=np.int64)
        self._layout = ModelLayout(self.schema, self.n_components, self.latent_dim, self.hidden_dims, self.dropout,
                                   self.behaviour, self._norm_stats_per_col)
        self._engine = eng = Engine(self._layout, self._select_device())
        eng.load_state(self._layout.init_params(seed=self.random_state))
        train = self._table(X_train)
        val = self._table(X_val) if n_val else None
        n_train = train.n_rows
        eff = min(bs, n_train) if n_train else bs
        n_batches = max(1, math.ceil(max(n_train, 1) / max(eff, 1)))
        kl_steps = max(1, kl_epochs * n_batches)  # model.py:1448
        gen = torch.Generator(device=eng.device)
        gen.manual_seed(self.random_state)
        step = 0
        self._train_history, self._warmup_val_history = [], []
        for epoch in range(n_epochs):
            tau = self._gumbel_temperature_for_epoch(epoch) if self.behaviour == "unsupervised" else self.gumbel_temperature
            if self.behaviour == "unsupervised":
                self._inference_tau = tau
            perm = torch.randperm(n_train, device=eng.device, generator=gen, dtype=torch.int64).to(torch.int32)
            eng.sums.zero_()
            for start in range(0, n_train, max(eff, 1)):  # last batch ragged, no drop-last (model.py:1475-1476)
                idx = perm[start:start + eff]
                step += 1
                beta = self.beta * min(step, kl_steps) / kl_steps  # losses.kl_warmup
                eng.elbo(train, 0, int(idx.numel()), row_index=idx, beta_scale=beta, tau=tau, grad=True,
                         seed=self.random_state, step=step, accumulate_sums=True)
                eng.adam_step(self.learning_rate)
            s = eng.sums.cpu().numpy()  # ONE host sync per epoch (the reference syncs 4-5x per batch)
            cnt = max(s[3], 1.0)
            self._train_history.append({"reconstruction": s[0] / cnt, "categorical_kl": s[1] / cnt,
                                        "gaussian_kl": s[2] / cnt})
            if val is not None and val.n_rows:
                self._warmup_val_history.append(self._elbo_on_table(val, bs, tau, seed=self.random_state + 1 + epoch))
        self._cache_training_statistics(train, bs)
        self._is_fitted = True
        return self

    def _elbo_on_table(self, table: PackedTable, batch_size: int, tau: float, seed: int) -> Dict[str, float]:
        """_compute_elbo_on_dataset (model.py:2497-2559): eval mode, beta = self.beta, fresh noise."""

        eng = self._engine
        eng.sums.zero_()
        eff = min(batch_size, table.n_rows)
        for i, start in enumerate(range(0, table.n_rows, eff)):
            nb = min(eff, table.n_rows - start)
            eng.elbo(table, start, nb, beta_scale=self.beta, tau=tau, grad=False, training=False, seed=seed, step=i,
                     accumulate_sums=True)
        s = eng.sums.cpu().numpy()
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
        for start in range(0, train.n_rows, chunk):
            outs.append(eng.encode(train, start, min(chunk, train.n_rows - start), temperature=tau))
        if not outs:
            self._train_cache = {}
            return
        enc = torch.cat([o["enc_out"] for o in outs])
        self._train_cache = {
            "probs": torch.cat([o["probs"] for o in outs]).contiguous(),
            "mu": enc[:, K:K + K * D].reshape(-1, K, D).contiguous(),
            "logvar": enc[:, K + K * D:].reshape(-1, K, D).contiguous(),
            "mean": torch.cat([o["mean"] for o in outs]),
        }

    # ------------------------------------------------------------------
    def encode(self, X: pd.DataFrame, *, return_components: bool = False):
        self._require_fitted("encode")
        table = self._table(X.loc[:, list(self.schema.feature_names)])
        K, D = self.n_components, self.latent_dim
        if table.n_rows == 0:
            empty = np.empty((0, D), np.float32)
            if not return_components:
                return empty
            return {"mean": empty, "assignments": np.empty((0, K), np.float32), "component_mu": empty.copy(),
                    "component_logvar": empty.copy()}
        tau = self._inference_tau if self.behaviour == "unsupervised" else None
        out = self._engine.encode(table, 0, table.n_rows, temperature=tau)
        mean = out["mean"].cpu().numpy()
        if not return_components:
            return mean
        enc = out["enc_out"]
        k = enc[:, :K].argmax(dim=1)  # hard assignment on softmax(logits) (model.py:4254-4258)
        rows = torch.arange(enc.shape[0], device=enc.device)
        mu = enc[:, K:K + K * D].reshape(-1, K, D)[rows, k]
        lv = enc[:, K + K * D:].reshape(-1, K, D)[rows, k]
        return {"mean": mean, "assignments": torch.nn.functional.one_hot(k, K).float().cpu().numpy(),
                "component_mu": mu.cpu().numpy(), "component_logvar": lv.cpu().numpy()}

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
               seed: Optional[int] = None) -> pd.DataFrame:
        """Generate rows (model.py:4284-4431).  Row ``i`` depends only on ``(seed, i)``: the output is independent
        of ``batch_size`` and of how rows are sharded over GPUs."""

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
        frames = []
        for start in range(0, n_samples, chunk):
            nb = min(chunk, n_samples - start)
            if conditional:
                z, comp = eng.conditional_latents(labels_t[start:start + nb].contiguous(), class_off_t, class_rows_t,
                                                  self._train_cache["probs"], self._train_cache["mu"],
                                                  self._train_cache["logvar"], seed=seed, row_offset=start)
            else:
                z, comp = eng.prior_latents(nb, cdf, seed=seed, row_offset=start)
            onum, ocodes = eng.decode_sample(z, component=comp, seed=seed + 1, row_offset=start)
            frames.append(data_utils.frame_from_packed(onum.cpu().numpy(), ocodes.cpu().numpy(), self.schema,
                                                       self._norm_stats_per_col))
        return pd.concat(frames, ignore_index=True) if len(frames) > 1 else frames[0]

    # ------------------------------------------------------------------
    def impute(self, X: pd.DataFrame, *, only_missing: bool = True,
               assignment_strategy: Optional[Literal["soft", "hard", "sample"]] = None, seed: Optional[int] = None
               ) -> pd.DataFrame:
        """Single-pass imputation at the posterior mean latent (model.py:3512-3692)."""

        self._require_fitted("impute")
        missing_cols = [c for c in self.schema.feature_names if c not in X.columns]
        if missing_cols:
            raise KeyError("Input is missing required columns: " + ", ".join(missing_cols))
        strategy = (assignment_strategy or ("hard" if self.behaviour == "supervised" else "soft")).lower()
        if strategy not in {"soft", "hard", "sample"}:
            raise ValueError("assignment_strategy must be one of 'soft', 'hard' or 'sample'")
        eng = self._engine
        seed = self.random_state if seed is None else int(seed)
        aligned = X.loc[:, list(self.schema.feature_names)].reset_index(drop=True)
        xnum, codes, mask = data_utils.pack_frame(aligned, self.schema, self._norm_stats_per_col)
        table = PackedTable(xnum, codes, mask, eng.device)
        K, D = self.n_components, self.latent_dim
        tau = self._inference_tau if self.behaviour == "unsupervised" else None
        out = eng.encode(table, 0, table.n_rows, temperature=tau)
        enc, probs = out["enc_out"], out["probs"]
        mu = enc[:, K:K + K * D].reshape(-1, K, D)
        comp = assign = None
        if strategy == "hard":  # model.py:3627-3634
            comp = probs.argmax(dim=1).to(torch.int32)
            z = mu[torch.arange(mu.shape[0], device=mu.device), comp.long()].contiguous()
        elif strategy == "soft":  # model.py:3635-3637
            assign = probs.contiguous()
            z = out["mean"]
        else:  # relaxed Gumbel-softmax weights, model.py:3638-3644
            t = max(float(self._inference_tau if self.behaviour == "unsupervised" else self.gumbel_temperature), 1e-6)
            g = torch.Generator(device=eng.device)
            g.manual_seed(seed)
            e = torch.empty_like(probs).exponential_(generator=g)
            assign = torch.softmax((enc[:, :K] - torch.log(e)) / t, dim=1).contiguous()
            z = (assign.unsqueeze(-1) * mu).sum(dim=1).contiguous()
        onum, ocodes = eng.decode_sample(z, component=comp, assignments=assign, seed=seed + 1)
        recon = data_utils.frame_from_packed(onum.cpu().numpy(), ocodes.cpu().numpy(), self.schema,
                                             self._norm_stats_per_col)
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
    def predict_proba(self, X: pd.DataFrame, attr=None, *, mask=None, L: int = 100, seed: Optional[int] = None):
        """Monte-Carlo posterior predictive of a categorical / ordinal attribute (model.py:3816-3867)."""

        self._require_fitted("predict_proba")
        if attr is None:
            raise RuntimeError("classifier probabilities (attr=None) need the supervised head phases, which are "
                               "outside this package's hot path; pass attr=<categorical or ordinal column>")
        names = list(self.schema.feature_names)
        if isinstance(attr, (int, np.integer)):
            if not 0 <= int(attr) < len(names):
                raise ValueError(f"Attribute index {attr} is out of range")
            attr_name = names[int(attr)]
        else:
            if attr not in self.schema:
                raise ValueError(f"Unknown attribute '{attr}'")
            attr_name = str(attr)
        spec = self.schema[attr_name]
        if spec.type not in {"cat", "ordinal"}:
            raise ValueError("predict_proba is only defined for categorical or ordinal attributes")
        if L <= 0:
            raise ValueError("L must be a positive integer")
        eng = self._engine
        seed = self.random_state if seed is None else int(seed)
        aligned = X.loc[:, names].reset_index(drop=True)
        xnum, codes, m = data_utils.pack_frame(aligned, self.schema, self._norm_stats_per_col)
        if mask is not None:  # True marks missing (model.py:3748-3756)
            mm = np.asarray(mask.loc[:, names] if isinstance(mask, pd.DataFrame) else mask, dtype=bool)
            m = (m.astype(bool) & ~mm).astype(np.uint8)
        j = names.index(attr_name)
        m[:, j] = 0  # the attribute is always decoded as missing
        i_disc = [n for n in names if self.schema[n].type in ("cat", "ordinal")].index(attr_name)
        codes[:, i_disc] = -1
        i_num = 0
        for jj, nme in enumerate(names):
            if self.schema[nme].type in ("real", "pos", "count"):
                xnum[m[:, jj] == 0, i_num] = 0.0
                i_num += 1
        table = PackedTable(xnum, codes, m, eng.device)
        n_classes = int(spec.n_classes)
        if table.n_rows == 0:
            return torch.empty((0, n_classes), dtype=torch.float32)
        enc = eng.encode(table, 0, table.n_rows)["enc_out"]
        z, comp = eng.posterior_draws(enc, int(L), seed=seed)
        probs = eng.decode_params(z, j, component=comp)
        probs = probs.view(int(L), table.n_rows, -1).mean(dim=0).clamp(min=1e-8)
        probs = probs / probs.sum(dim=-1, keepdim=True).clamp_min(1e-8)
        return probs.cpu()
