"""The reference's model archive (``SUAVE.save`` / ``SUAVE.load``, suave/model.py:4456-4579, :4581-4950) as
plain functions on tensors -- no GPU needed, so the payload layout is testable anywhere.

Payload written by the reference's ``save`` and read by its ``_load_from_payload``::

    metadata   constructor arguments + schema dict + inference_tau + auto-configuration flags
    modules    encoder / decoder   state_dicts under the reference's key names
               classifier          {"state_dict", "use_weight", "class_weight"} or None
               temperature_scaler  {"temperature", "fitted"}
               prior               {"logits", "logvar", "mu"} (+ "mean_state_dict" when unsupervised)
    artefacts  classes, class_to_index, normalization, feature_layout, is_calibrated, the training posterior
               caches (train_latent_mu / _logvar, train_component_logits / _mu / _logvar / _probs),
               train_target_indices, warmup_val_history, joint_val_metrics, the (unused here) prediction caches

Archives written here hold only tensors and plain Python containers, so ``torch.load(weights_only=True)``
reads them; the reference stores ``classes`` / ``train_target_indices`` as numpy arrays and reads lists or
tensors just as well (``np.asarray`` on load, model.py:4780-4828).
"""

from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, Mapping, Optional

import numpy as np
import torch

METADATA_KEYS = (
    "schema", "behaviour", "latent_dim", "n_components", "beta", "hidden_dims", "head_hidden_dims", "dropout",
    "learning_rate", "batch_size", "kl_warmup_epochs", "val_split", "stratify", "random_state", "gumbel_temperature",
    "tau_start", "tau_min", "tau_decay", "inference_tau", "warmup_epochs", "head_epochs", "finetune_epochs",
    "decoder_refine_epochs", "decoder_refine_mode", "decoder_refine_prior_lambda", "joint_decoder_lr_scale",
    "early_stop_patience", "classification_loss_weight", "auto_configured", "heuristic_overrides",
    "auto_hyperparameters")  # model.py:4484-4522
MODULE_KEYS = ("encoder", "decoder", "classifier", "temperature_scaler", "prior")  # model.py:4545-4551
ARTEFACT_KEYS = (
    "classes", "class_to_index", "normalization", "feature_layout", "temperature_scaler_state", "is_calibrated",
    "train_latent_mu", "train_latent_logvar", "train_component_logits", "train_component_mu",
    "train_component_logvar", "train_component_probs", "train_target_indices", "cached_logits",
    "cached_probabilities", "logits_cache_key", "probability_cache_key", "warmup_val_history",
    "joint_val_metrics")  # model.py:4552-4572


def _t(a) -> torch.Tensor:
    return a.detach().cpu().clone() if isinstance(a, torch.Tensor) else torch.from_numpy(np.array(a, copy=True))


def feature_layout(schema) -> Dict[str, Dict[str, int]]:
    """``SUAVE._feature_layout`` (model.py:2813-2921): per type, column -> width in the encoder input."""

    out: Dict[str, Dict[str, int]] = {"real": {}, "pos": {}, "count": {}, "cat": {}, "ordinal": {}}
    for name in schema.feature_names:
        spec = schema[name]
        out[spec.type][name] = 1 if spec.type in ("real", "pos", "count") else int(spec.n_classes)
    return out


def prior_tensors(state: Mapping[str, np.ndarray], behaviour: str, K: int, D: int):
    """prior logits / means / log-variances as the reference's accessors return them (model.py:595-618)."""

    if behaviour == "supervised":
        return state["prior.logits"], state["prior.mu"], state["prior.logvar"]
    w, b = np.asarray(state["prior_mean.linear.weight"]), np.asarray(state["prior_mean.linear.bias"])
    return np.zeros(K, np.float32), (w.T + b[None, :]).astype(np.float32), np.zeros((K, D), np.float32)


def build_payload(*, metadata: Mapping[str, Any], state: Mapping[str, np.ndarray], schema, classifier_state=None,
                  class_weight=None, use_weight: bool = False, temperature_scaler: Mapping[str, Any],
                  norm_stats: Mapping[str, Any], classes=None, class_to_index=None, is_calibrated: bool = False,
                  train_cache: Optional[Mapping[str, torch.Tensor]] = None, train_target_indices=None,
                  warmup_val_history=None, joint_val_metrics=None) -> Dict[str, Any]:
    """The dictionary ``SUAVE.save`` hands to ``torch.save`` (model.py:4573-4578).  ``state`` maps the reference's
    parameter names (``encoder.*``, ``decoder.*``, ``prior.*`` / ``prior_mean.*``) to arrays."""

    behaviour = metadata["behaviour"]
    K, D = int(metadata["n_components"]), int(metadata["latent_dim"])
    enc = OrderedDict((k[len("encoder."):], _t(v)) for k, v in state.items() if k.startswith("encoder."))
    dec = OrderedDict((k[len("decoder."):], _t(v)) for k, v in state.items() if k.startswith("decoder."))
    logits, mu, logvar = prior_tensors(state, behaviour, K, D)
    prior: Dict[str, Any] = {"logits": _t(logits), "logvar": _t(logvar), "mu": _t(mu)}
    if behaviour == "unsupervised":
        prior["mean_state_dict"] = OrderedDict((k[len("prior_mean."):], _t(v)) for k, v in state.items()
                                               if k.startswith("prior_mean."))
    clf = None
    if classifier_state is not None:
        sd = OrderedDict((k, _t(v)) for k, v in classifier_state.items())
        clf = {"state_dict": sd, "use_weight": bool(use_weight),
               "class_weight": _t(class_weight) if (use_weight and class_weight is not None) else None}
    meta = {k: metadata.get(k) for k in METADATA_KEYS}
    meta["schema"] = schema.to_dict()
    meta["hidden_dims"] = [int(h) for h in metadata["hidden_dims"]]
    meta["head_hidden_dims"] = [int(h) for h in metadata.get("head_hidden_dims") or ()]
    meta["auto_configured"] = dict(metadata.get("auto_configured") or {})
    meta["heuristic_overrides"] = dict(metadata.get("heuristic_overrides") or meta["auto_configured"])
    cache = train_cache or {}
    get = lambda k: None if cache.get(k) is None else _t(cache[k])
    art = {
        "classes": None if classes is None else [c.item() if isinstance(c, np.generic) else c for c in classes],
        "class_to_index": None if class_to_index is None else dict(class_to_index),
        "normalization": norm_stats, "feature_layout": feature_layout(schema), "temperature_scaler_state": None,
        "is_calibrated": bool(is_calibrated),
        "train_latent_mu": get("mean"), "train_latent_logvar": get("post_logvar"),
        "train_component_logits": get("logits"), "train_component_mu": get("mu"),
        "train_component_logvar": get("logvar"), "train_component_probs": get("probs"),
        "train_target_indices": None if train_target_indices is None else _t(np.asarray(train_target_indices, np.int64)),
        "cached_logits": None, "cached_probabilities": None, "logits_cache_key": None, "probability_cache_key": None,
        "warmup_val_history": list(warmup_val_history or []), "joint_val_metrics": joint_val_metrics,
    }
    return {"metadata": meta, "modules": {"encoder": enc, "decoder": dec, "classifier": clf,
                                          "temperature_scaler": dict(temperature_scaler), "prior": prior},
            "artefacts": art}


def state_from_payload(payload: Mapping[str, Any]) -> Dict[str, np.ndarray]:
    """Parameter arrays by the reference's names out of a payload (either implementation's), following
    ``_load_from_payload`` (model.py:4834-4947): module state dicts, then the prior block; an unsupervised archive
    without ``mean_state_dict`` falls back to the stored component means (``PriorMean.load_component_means``,
    prior.py:41-51: weight = mu^T, bias = 0)."""

    meta, mods = payload["metadata"], payload["modules"]
    npy = lambda v: np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, np.float32)
    out: Dict[str, np.ndarray] = {}
    for mod in ("encoder", "decoder"):
        sd = mods.get(mod)
        if sd is None:
            raise ValueError(f"archive holds no {mod} state")
        for k, v in sd.items():
            out[f"{mod}.{k}"] = npy(v)
    prior = mods.get("prior") or {}
    behaviour = str(meta.get("behaviour", "supervised")).lower()
    K, D = int(meta["n_components"]), int(meta["latent_dim"])
    if behaviour == "supervised":
        for name in ("logits", "mu", "logvar"):
            shape = (K,) if name == "logits" else (K, D)
            v = prior.get(name)
            if v is None:
                out["prior." + name] = np.zeros(shape, np.float32)  # the constructor's zeros (model.py:533-551)
                continue
            v = npy(v)
            if v.shape != shape:
                what = {"logits": "logits do", "mu": "means do", "logvar": "log-variance does"}[name]
                raise ValueError(f"Saved prior {what} not match the model configuration")
            out["prior." + name] = v
    else:
        ms = prior.get("mean_state_dict")
        if ms is not None:
            for k, v in ms.items():
                out["prior_mean." + k] = npy(v)
        elif prior.get("mu") is not None:
            mu = npy(prior["mu"])
            out["prior_mean.linear.weight"] = np.ascontiguousarray(mu.T)
            out["prior_mean.linear.bias"] = np.zeros(D, np.float32)
        else:
            out["prior_mean.linear.weight"] = np.zeros((D, K), np.float32)
            out["prior_mean.linear.bias"] = np.zeros(D, np.float32)
    return out


def load_archive(path) -> Dict[str, Any]:
    """Read an archive without executing anything from it: ``weights_only=True``; numpy arrays (the reference
    stores ``classes`` and ``train_target_indices`` that way) are admitted through torch's allow-list of their
    reconstruction helpers only."""

    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as first:  # noqa: BLE001 - retry once with numpy's array types allow-listed
        safe = [np.ndarray, np.dtype]
        try:
            from numpy._core import multiarray as _ma
        except ImportError:  # numpy < 2
            from numpy.core import multiarray as _ma
        safe += [_ma._reconstruct, _ma.scalar]
        safe += [type(np.dtype(t)) for t in ("int64", "int32", "float32", "float64", "bool", "uint8", "object", "<U8")]
        try:
            with torch.serialization.safe_globals(safe):
                return torch.load(path, map_location="cpu", weights_only=True)
        except Exception as second:  # noqa: BLE001 - the retry's error is the informative one (model.py:4669-4674 chains it too)
            raise ValueError(f"Unexpected model archive format: {second} (first attempt, without numpy's array types "
                             f"allow-listed: {first})") from second


def tree(obj, prefix: str = "") -> Dict[str, str]:
    """Flat description of a payload: path -> "tensor:<dtype>:<shape>" / Python type name.  Two archives are
    layout-compatible when their trees agree."""

    out: Dict[str, str] = {}
    if isinstance(obj, Mapping) and (prefix.count("/") < 2 or prefix.endswith(("state_dict", "prior", "classifier"))
                                     or prefix.split("/")[-1] in ("encoder", "decoder", "mean_state_dict")):
        if not obj:
            out[prefix] = "dict:empty"
        for k, v in obj.items():
            out.update(tree(v, f"{prefix}/{k}" if prefix else str(k)))
    elif isinstance(obj, torch.Tensor):
        out[prefix] = f"tensor:{str(obj.dtype).replace('torch.', '')}:{tuple(obj.shape)}"
    elif isinstance(obj, np.ndarray):
        out[prefix] = f"array:{tuple(obj.shape)}"
    else:
        out[prefix] = type(obj).__name__
    return out
