"""CPU restatement (numpy) of SUAVE's ELBO training step and generative decode.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``suave_amd/``) may
import, call, link or execute this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
there only as the checker / the timed CPU baseline.

Every function cites the file:line of the reference (xuxu-wei/SUAVE @
2025-10-03, mounted at ``/root/reference`` in the build container) that it
restates.  Parity of this restatement is PINNED: ``tests/golden/*.npz`` hold
outputs of the untouched reference run in the build container by
``tests/golden/make_golden.py`` (inputs, injected noise, per-row ELBO terms,
all parameter gradients, Adam-after-one-step parameters, sampled values), and
``tests/test_oracle_golden.py`` checks this module against them together with
the reference's own known-answer tests (``tests/test_third_party_alignment.py``
and friends, restated in ``tests/test_oracle_known_answers.py``).

Layout conventions are the reference's own (dense one-hot / thermometer
tensors, ``state_dict`` key names); the product's packed layouts are converted
to these by ``dense_from_packed`` below so the two sides see identical inputs.

Noise is always an explicit input (SURVEY.md §8a "noise inventory"): dropout
keep-masks, the Exp(1) draws behind the Gumbel noise, the N(0,1) ``eps`` of the
reparameterisation, and per-cell draws for sampling.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

try:  # scipy is only needed for lgamma (Poisson log-likelihood)
    from scipy.special import gammaln as _gammaln
except Exception:  # pragma: no cover
    _gammaln = np.vectorize(math.lgamma)

EPS = 1e-6  # suave/modules/distributions.py:18
LOG_2PI = math.log(2.0 * math.pi)  # suave/modules/distributions.py:19
LOGVAR_RANGE = (-15.0, 15.0)  # suave/modules/encoder.py:31

TYPE_ORDER = ("real", "pos", "count", "cat", "ordinal")  # model.py:2813-2900


# ---------------------------------------------------------------------------
# elementary functions (torch semantics)
# ---------------------------------------------------------------------------
def softplus(x: np.ndarray) -> np.ndarray:
    """torch.nn.functional.softplus, beta=1, threshold=20."""

    x = np.asarray(x)
    safe = np.minimum(x, 20.0)
    return np.where(x > 20.0, x, np.log1p(np.exp(safe))).astype(x.dtype)


def sigmoid(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x)
    one = x.dtype.type(1.0)
    return (one / (one + np.exp(-x))).astype(x.dtype)


def softplus_grad(x: np.ndarray) -> np.ndarray:
    """d softplus / dx with torch's threshold (1 above 20)."""

    return np.where(x > 20.0, np.ones_like(x), sigmoid(x))


def log_softmax(x: np.ndarray, axis: int = -1) -> np.ndarray:
    m = np.max(x, axis=axis, keepdims=True)
    z = x - m
    return z - np.log(np.sum(np.exp(z), axis=axis, keepdims=True))


def softmax(x: np.ndarray, axis: int = -1) -> np.ndarray:
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


# ---------------------------------------------------------------------------
# schema description (suave/types.py:9-33, :267-287)
# ---------------------------------------------------------------------------
@dataclass
class Column:
    name: str
    type: str
    n_classes: Optional[int] = None
    y_dim: Optional[int] = None
    stats: Dict[str, float] = field(default_factory=dict)

    def resolved_y_dim(self) -> int:
        if self.y_dim is not None:
            return int(self.y_dim)
        if self.type in ("real", "pos", "count"):
            return 1
        return int(self.n_classes)

    def width(self) -> int:
        """Width of the column inside ``encoder_inputs`` (model.py:2826-2900)."""

        return 1 if self.type in ("real", "pos", "count") else int(self.n_classes)


class OracleModel:
    """Shapes, parameter names and slices of one SUAVE model.

    Parameter dictionary keys follow the reference ``state_dict`` layout
    (SURVEY §8b): ``encoder.backbone.<i>.weight`` ... ``decoder.heads.<col>.
    mean_layer.weight`` plus ``prior.logits|mu|logvar`` (supervised,
    model.py:543-551) or ``prior_mean.linear.weight|bias`` (unsupervised,
    prior.py:20-39).
    """

    def __init__(
        self,
        columns: Sequence[Column],
        *,
        n_components: int,
        latent_dim: int,
        hidden: Sequence[int] = (256, 128),
        dropout: float = 0.0,
        behaviour: str = "supervised",
    ) -> None:
        self.columns = list(columns)
        self.K = int(n_components)
        self.D = int(latent_dim)
        self.hidden = tuple(int(h) for h in hidden)
        self.dropout = float(dropout)
        self.behaviour = behaviour
        # decoder.py:306-314 -- y slices in schema order
        self.y_slices: Dict[str, slice] = {}
        start = 0
        for col in self.columns:
            yd = col.resolved_y_dim()
            self.y_slices[col.name] = slice(start, start + yd)
            start += yd
        self.Y = start
        # model.py:2813-2905 -- encoder input: values by type, then masks by type
        self.enc_value_slices: Dict[str, slice] = {}
        self.enc_mask_index: Dict[str, int] = {}
        pos = 0
        ordered: List[Column] = []
        for t in TYPE_ORDER:
            for col in self.columns:
                if col.type == t:
                    ordered.append(col)
        for col in ordered:
            self.enc_value_slices[col.name] = slice(pos, pos + col.width())
            pos += col.width()
        for i, col in enumerate(ordered):
            self.enc_mask_index[col.name] = pos + i
        self.I = pos + len(ordered)
        self.by_type_order = [c.name for c in ordered]
        # backbone module indices: Linear, ReLU, (Dropout) -> step 3 or 2
        self.backbone_step = 3 if self.dropout > 0 else 2

    # -- parameter bookkeeping ------------------------------------------------
    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        shapes: Dict[str, Tuple[int, ...]] = {}
        prev = self.I
        for i, h in enumerate(self.hidden):  # encoder.py:46-53
            idx = i * self.backbone_step
            shapes[f"encoder.backbone.{idx}.weight"] = (h, prev)
            shapes[f"encoder.backbone.{idx}.bias"] = (h,)
            prev = h
        shapes["encoder.component_logits.weight"] = (self.K, prev)  # encoder.py:56
        shapes["encoder.component_logits.bias"] = (self.K,)
        shapes["encoder.mu_layer.weight"] = (self.K * self.D, prev)  # encoder.py:57
        shapes["encoder.mu_layer.bias"] = (self.K * self.D,)
        shapes["encoder.logvar_layer.weight"] = (self.K * self.D, prev)  # :58
        shapes["encoder.logvar_layer.bias"] = (self.K * self.D,)
        prev = self.D
        for i, h in enumerate(self.hidden):  # decoder.py:286-294
            idx = i * self.backbone_step
            shapes[f"decoder.backbone.{idx}.weight"] = (h, prev)
            shapes[f"decoder.backbone.{idx}.bias"] = (h,)
            prev = h
        shapes["decoder.y_projection.weight"] = (self.Y, prev)  # decoder.py:300
        shapes["decoder.y_projection.bias"] = (self.Y,)
        K = self.K
        for col in self.columns:  # decoder.py:56-265 (all head linears bias-free)
            yd = col.resolved_y_dim()
            p = f"decoder.heads.{col.name}."
            if col.type in ("real", "pos"):
                shapes[p + "mean_layer.weight"] = (1, yd + K)
                shapes[p + "var_layer.weight"] = (1, K)
            elif col.type == "count":
                shapes[p + "rate_layer.weight"] = (1, yd + K)
            elif col.type == "cat":
                shapes[p + "logits_layer.weight"] = (col.n_classes - 1, yd + K)
            elif col.type == "ordinal":
                shapes[p + "partition_layer.weight"] = (col.n_classes - 1, K)
                shapes[p + "mean_layer.weight"] = (1, yd + K)
            else:
                raise ValueError(col.type)
        if self.behaviour == "supervised":  # model.py:543-551
            shapes["prior.logits"] = (K,)
            shapes["prior.mu"] = (K, self.D)
            shapes["prior.logvar"] = (K, self.D)
        else:  # model.py:538-542, prior.py:20
            shapes["prior_mean.linear.weight"] = (self.D, K)
            shapes["prior_mean.linear.bias"] = (self.D,)
        return shapes

    def init_params(self, seed: int = 0, dtype=np.float32, scale: float = 1.0) -> Dict[str, np.ndarray]:
        """nn.Linear-style uniform(+-1/sqrt(fan_in)) init with a numpy generator."""

        rng = np.random.default_rng(seed)
        params: Dict[str, np.ndarray] = {}
        for name, shape in self.param_shapes().items():
            if name.startswith("prior"):
                params[name] = (0.1 * rng.standard_normal(shape)).astype(dtype)
                continue
            fan_in = shape[-1] if len(shape) == 2 else None
            if fan_in is None:
                w_name = name.replace(".bias", ".weight")
                fan_in = self.param_shapes()[w_name][-1]
            bound = scale / math.sqrt(max(fan_in, 1))
            params[name] = rng.uniform(-bound, bound, size=shape).astype(dtype)
        return params

    # -- prior accessors (model.py:595-618, prior.py:31-39) --------------------
    def prior_tensors(self, P: Mapping[str, np.ndarray]):
        if self.behaviour == "supervised":
            return P["prior.logits"], P["prior.mu"], P["prior.logvar"]
        w = P["prior_mean.linear.weight"]
        b = P["prior_mean.linear.bias"]
        mu = w.T + b[None, :]
        dt = mu.dtype
        return np.zeros(self.K, dt), mu, np.zeros((self.K, self.D), dt)


# ---------------------------------------------------------------------------
# packed (product) layout -> reference dense layout  (model.py:2784-2922)
# ---------------------------------------------------------------------------
def dense_from_packed(
    model: OracleModel,
    xnum: np.ndarray,
    codes: np.ndarray,
    mask: np.ndarray,
    dtype=np.float32,
):
    """Expand the product's packed batch into the reference's tensors.

    ``xnum [B, n_numeric]`` holds real/pos/count cells (schema order among the
    numeric columns), ``codes [B, n_discrete]`` holds the category code
    (``-1`` = all-zero one-hot, model.py:2866-2870) or the ordinal level
    (``-1`` = all-zero thermometer, model.py:2884-2891) and ``mask [B, n_cols]``
    holds 1 for observed cells in schema order.  Returns ``(encoder_inputs,
    data, masks)`` with ``data[col]`` of shape ``[B,1]`` or ``[B,C]`` and
    ``masks[col]`` of shape ``[B,1]``.
    """

    B = mask.shape[0]
    data: Dict[str, np.ndarray] = {}
    masks: Dict[str, np.ndarray] = {}
    i_num = 0
    i_disc = 0
    for j, col in enumerate(model.columns):
        masks[col.name] = mask[:, j : j + 1].astype(dtype)
        if col.type in ("real", "pos", "count"):
            data[col.name] = xnum[:, i_num : i_num + 1].astype(dtype)
            i_num += 1
        else:
            C = int(col.n_classes)
            code = codes[:, i_disc].astype(np.int64)
            i_disc += 1
            ar = np.arange(C)[None, :]
            if col.type == "cat":
                dense = (ar == code[:, None]) & (code[:, None] >= 0)
            else:
                dense = (ar <= code[:, None]) & (code[:, None] >= 0)
            data[col.name] = dense.astype(dtype)
    x_enc = np.zeros((B, model.I), dtype)
    for col in model.columns:
        x_enc[:, model.enc_value_slices[col.name]] = data[col.name]
        x_enc[:, model.enc_mask_index[col.name]] = masks[col.name][:, 0]
    return x_enc, data, masks


# ---------------------------------------------------------------------------
# MLP helpers (encoder.py:46-54, decoder.py:286-294)
# ---------------------------------------------------------------------------
def _backbone_forward(prefix, model, P, x, drop_masks, training):
    acts = [x]
    pre_masks = []
    h = x
    p = model.dropout
    for i, _ in enumerate(model.hidden):
        idx = i * model.backbone_step
        W = P[f"{prefix}.backbone.{idx}.weight"]
        b = P[f"{prefix}.backbone.{idx}.bias"]
        a = h @ W.T + b
        relu = a > 0
        h = np.where(relu, a, 0).astype(a.dtype)
        scale = np.ones_like(h)
        if training and p > 0:
            keep = drop_masks[i].astype(h.dtype)
            scale = keep / h.dtype.type(1.0 - p)  # F.dropout: mask / (1-p)
            h = h * scale
        pre_masks.append(relu.astype(h.dtype) * scale)
        acts.append(h)
    return acts, pre_masks


def _backbone_backward(prefix, model, P, acts, pre_masks, dh, grads, need_dx):
    for i in reversed(range(len(model.hidden))):
        idx = i * model.backbone_step
        W = P[f"{prefix}.backbone.{idx}.weight"]
        da = dh * pre_masks[i]
        grads[f"{prefix}.backbone.{idx}.weight"] = da.T @ acts[i]
        grads[f"{prefix}.backbone.{idx}.bias"] = da.sum(axis=0)
        if i > 0 or need_dx:
            dh = da @ W
    return dh


def encoder_forward(model, P, x_enc, drop_masks=None, training=False):
    """EncoderMLP.forward -- suave/modules/encoder.py:60-72."""

    acts, pre = _backbone_forward("encoder", model, P, x_enc, drop_masks, training)
    h = acts[-1]
    B = x_enc.shape[0]
    logits = h @ P["encoder.component_logits.weight"].T + P["encoder.component_logits.bias"]
    mu = h @ P["encoder.mu_layer.weight"].T + P["encoder.mu_layer.bias"]
    lv_raw = h @ P["encoder.logvar_layer.weight"].T + P["encoder.logvar_layer.bias"]
    logvar = np.clip(lv_raw, LOGVAR_RANGE[0], LOGVAR_RANGE[1])
    cache = dict(acts=acts, pre=pre, lv_raw=lv_raw.reshape(B, model.K, model.D))
    return logits, mu.reshape(B, model.K, model.D), logvar.reshape(B, model.K, model.D), cache


def decoder_trunk_forward(model, P, z, drop_masks=None, training=False):
    """Decoder backbone + y_projection -- suave/modules/decoder.py:343-344."""

    acts, pre = _backbone_forward("decoder", model, P, z, drop_masks, training)
    y = acts[-1] @ P["decoder.y_projection.weight"].T + P["decoder.y_projection.bias"]
    return y, dict(acts=acts, pre=pre)


# ---------------------------------------------------------------------------
# likelihood heads, forward + analytic backward  (decoder.py:56-265,
# distributions.py:42-164; gradient formulas: SURVEY Appendix A)
# ---------------------------------------------------------------------------
def head_forward(col: Column, K: int, P, y, s, x, mask, *, want_grad=False, upstream=None):
    """One likelihood head.

    Returns ``out`` with ``log_px``, ``log_px_missing``, ``params`` (reference
    keys, decoder.py:89,134-138,173,215,259) and -- if ``want_grad`` -- the
    gradients of ``sum_b upstream[b] * log_px[b]`` w.r.t. ``y``, ``s`` and the
    head weights.  Masked rows contribute exactly zero gradient
    (``_apply_observed_linear``, decoder.py:41-53).
    """

    dt = y.dtype
    pfx = f"decoder.heads.{col.name}."
    m = mask[:, 0].astype(dt)
    mm = (1.0 - m).astype(dt)
    f = np.concatenate([y, s], axis=1)
    yd = y.shape[1]
    out: Dict[str, object] = {}
    g: Dict[str, np.ndarray] = {}
    st = col.stats or {}
    if col.type in ("real", "pos"):
        Wm = P[pfx + "mean_layer.weight"]
        Wv = P[pfx + "var_layer.weight"]
        a = f @ Wm[0]
        bv = s @ Wv[0]
        v = softplus(bv) + dt.type(EPS)
        if col.type == "real":  # decoder.py:77-83
            shift = dt.type(st.get("mean", 0.0))
            scale = dt.type(st.get("std", 1.0))
        else:  # decoder.py:121-128
            shift = dt.type(st.get("mean_log", 0.0))
            scale = dt.type(st.get("std_log", 1.0))
        mean = a * scale + shift
        vs = v * (scale * scale)
        xr = x[:, 0] * scale + shift
        c = np.maximum(vs, dt.type(EPS))  # distributions.py:50 / :105
        r = xr - mean
        nll = 0.5 * (dt.type(LOG_2PI) + np.log(c) + r * r / c)
        if col.type == "pos":
            nll = nll + xr  # distributions.py:107
        out["log_px"] = -(nll * m)
        out["log_px_missing"] = -(nll * mm)
        if col.type == "real":
            out["params"] = {"mean": mean[:, None], "var": vs[:, None]}
        else:
            out["params"] = {
                "mean_log": mean[:, None],
                "var_log": vs[:, None],
                "mean": (np.exp(mean + 0.5 * np.maximum(vs, 0)) - 1.0)[:, None],
            }
        out["_raw"] = {"mean_raw": a, "var": v}
        if want_grad:
            u = upstream * m
            ga = u * (r / c) * scale
            pass_c = (vs >= dt.type(EPS)).astype(dt)
            gb = u * (-0.5) * (1.0 / c - r * r / (c * c)) * pass_c * (scale * scale) * softplus_grad(bv)
            g["f_mean"] = ga
            g[pfx + "mean_layer.weight"] = (ga[:, None] * f).sum(axis=0)[None, :]
            g[pfx + "var_layer.weight"] = (gb[:, None] * s).sum(axis=0)[None, :]
            g["dy"] = ga[:, None] * Wm[0, :yd][None, :]
            g["ds"] = ga[:, None] * Wm[0, yd:][None, :] + gb[:, None] * Wv[0][None, :]
    elif col.type == "count":  # decoder.py:163-173, distributions.py:120-127
        Wr = P[pfx + "rate_layer.weight"]
        rho = f @ Wr[0]
        rate = softplus(rho) + dt.type(EPS)
        offset = dt.type(st.get("offset", 0.0))
        counts = np.maximum(np.exp(x[:, 0]) - offset, 0).astype(dt)
        rc = np.maximum(rate, dt.type(EPS))
        with np.errstate(divide="ignore", invalid="ignore"):
            xlogy = np.where(counts == 0, 0, counts * np.log(rc)).astype(dt)
        lp = xlogy - rc - _gammaln(counts.astype(np.float64) + 1.0).astype(dt)
        out["log_px"] = lp * m
        out["log_px_missing"] = lp * mm
        out["params"] = {"rate": rate[:, None]}
        if want_grad:
            u = upstream * m
            gr = u * (counts / rc - 1.0) * softplus_grad(rho)
            g[pfx + "rate_layer.weight"] = (gr[:, None] * f).sum(axis=0)[None, :]
            g["dy"] = gr[:, None] * Wr[0, :yd][None, :]
            g["ds"] = gr[:, None] * Wr[0, yd:][None, :]
    elif col.type == "cat":  # decoder.py:202-216, distributions.py:57-64
        W = P[pfx + "logits_layer.weight"]
        partial = f @ W.T
        logits = np.concatenate([np.zeros((y.shape[0], 1), dt), partial], axis=1)
        logp = log_softmax(logits)
        lp = (x * logp).sum(axis=1)
        out["log_px"] = lp * m
        out["log_px_missing"] = lp * mm
        out["params"] = {"logits": logits, "probs": softmax(logits)}
        if want_grad:
            u = upstream * m
            gl = u[:, None] * (x - x.sum(axis=1, keepdims=True) * np.exp(logp))
            gp = gl[:, 1:]
            g[pfx + "logits_layer.weight"] = gp.T @ f
            df = gp @ W
            g["dy"] = df[:, :yd]
            g["ds"] = df[:, yd:]
    elif col.type == "ordinal":  # decoder.py:246-259, distributions.py:137-164
        Wp = P[pfx + "partition_layer.weight"]
        Wm = P[pfx + "mean_layer.weight"]
        part = s @ Wp.T
        mean_o = f @ Wm[0]
        spac = softplus(part) + dt.type(EPS)
        thr = np.cumsum(spac, axis=1, dtype=dt)
        sg = sigmoid(thr - mean_o[:, None])
        ones = np.ones((y.shape[0], 1), dt)
        zeros = np.zeros((y.shape[0], 1), dt)
        p_raw = np.concatenate([sg, ones], axis=1) - np.concatenate([zeros, sg], axis=1)
        probs = np.clip(p_raw, dt.type(EPS), dt.type(1.0))
        levels = np.rint(x.sum(axis=1)).astype(np.int64)
        tgt = np.maximum(levels - 1, 0)
        lp = np.log(np.maximum(probs, dt.type(EPS)))[np.arange(y.shape[0]), tgt]
        out["log_px"] = lp * m
        out["log_px_missing"] = lp * mm
        out["params"] = {"probs": probs, "thresholds": thr, "mean": mean_o[:, None]}
        if want_grad:
            C = probs.shape[1]
            u = upstream * m
            rows = np.arange(y.shape[0])
            pt_raw = p_raw[rows, tgt]
            passed = ((pt_raw >= dt.type(EPS)) & (pt_raw <= 1.0)).astype(dt)
            gpt = u * passed / probs[rows, tgt]
            dsg = np.zeros_like(sg)
            lo = tgt < C - 1
            dsg[rows[lo], tgt[lo]] += gpt[lo]
            hi = tgt > 0
            dsg[rows[hi], tgt[hi] - 1] -= gpt[hi]
            du = dsg * sg * (1.0 - sg)
            gmean = -du.sum(axis=1)
            dspac = np.cumsum(du[:, ::-1], axis=1, dtype=dt)[:, ::-1]
            gpart = dspac * softplus_grad(part)
            g[pfx + "partition_layer.weight"] = gpart.T @ s
            g[pfx + "mean_layer.weight"] = (gmean[:, None] * f).sum(axis=0)[None, :]
            g["dy"] = gmean[:, None] * Wm[0, :yd][None, :]
            g["ds"] = gmean[:, None] * Wm[0, yd:][None, :] + gpart @ Wp
    else:
        raise ValueError(col.type)
    if want_grad:
        out["grads"] = g
    return out


def heads_forward(model, P, y_full, s, data, masks, *, want_grad=False, upstream=None):
    """Decoder.forward column loop -- decoder.py:348-384 -- plus
    losses.sum_reconstruction_terms (losses.py:29-32)."""

    B = y_full.shape[0]
    dt = y_full.dtype
    recon = np.zeros(B, dt)
    recon_missing = np.zeros(B, dt)
    per_feature = {}
    dY = np.zeros_like(y_full) if want_grad else None
    dS = np.zeros_like(s) if want_grad else None
    wgrads: Dict[str, np.ndarray] = {}
    for col in model.columns:
        sl = model.y_slices[col.name]
        out = head_forward(
            col, model.K, P, y_full[:, sl], s, data[col.name], masks[col.name],
            want_grad=want_grad, upstream=upstream,
        )
        per_feature[col.name] = out
        recon = recon + out["log_px"]
        recon_missing = recon_missing + out["log_px_missing"]
        if want_grad:
            g = out["grads"]
            dY[:, sl] = g["dy"]
            dS += g["ds"]
            for k, v in g.items():
                if k.startswith("decoder."):
                    wgrads[k] = v
    return dict(recon=recon, recon_missing=recon_missing, per_feature=per_feature,
                dY=dY, dS=dS, wgrads=wgrads)


# ---------------------------------------------------------------------------
# KL terms (losses.py:51-57, :60-84, :87-97)
# ---------------------------------------------------------------------------
def kl_categorical(lq, lp):
    log_q = log_softmax(lq)
    log_p = log_softmax(np.broadcast_to(lp, lq.shape))
    return (np.exp(log_q) * (log_q - log_p)).sum(axis=-1)


def kl_normal_vs_normal(mu_q, lv_q, mu_p, lv_p):
    vp = np.maximum(np.exp(lv_p), mu_q.dtype.type(1e-6))
    d = mu_q - mu_p
    return 0.5 * (lv_p - lv_q + (np.exp(lv_q) + d * d) / vp - 1.0).sum(axis=-1)


def kl_normal_mixture(mu_q, lv_q, mu_p, lv_p, w):
    comp = kl_normal_vs_normal(mu_q, lv_q, mu_p[None], lv_p[None])
    return (w * comp).sum(axis=-1)


def kl_warmup(step: int, total_steps: int, beta_target: float) -> float:
    """losses.py:35-41."""

    if total_steps <= 0:
        return beta_target
    return beta_target * (min(max(step, 0), total_steps) / float(total_steps))


def gumbel_softmax(logits, exp_noise, tau):
    """torch F.gumbel_softmax: g = -log(Exp(1)); softmax((logits+g)/tau)."""

    g = -np.log(exp_noise)
    return softmax((logits + g) / logits.dtype.type(tau))


# ---------------------------------------------------------------------------
# supervised ELBO step  (model.py:1562-1618, 1656-1667)
# ---------------------------------------------------------------------------
def elbo_supervised(model, P, x_enc, data, masks, noise, *, beta_scale, tau=1.0,
                    training=True, want_grad=True, dlatent=None):
    """``SUAVE._forward_elbo_batch`` (behaviour="supervised") and, when
    ``want_grad``, the gradient of ``loss`` w.r.t. every parameter (what
    ``loss.backward()`` at model.py:1505 leaves in ``.grad``).  ``dlatent`` [B, D] is the gradient an extra
    loss term sends into ``outputs["latent"]`` -- the classifier's cross-entropy of joint fine-tuning
    (model.py:1987-1991)."""

    dt = x_enc.dtype
    B = x_enc.shape[0]
    K, D = model.K, model.D
    tau = max(float(tau), 1e-6)  # model.py:1587
    logits, mu, logvar, ecache = encoder_forward(model, P, x_enc, noise.get("enc_drop"), training)
    pl, pm, plv = model.prior_tensors(P)
    cat_kl = kl_categorical(logits, pl)
    y_soft = gumbel_softmax(logits, noise["exp"].astype(dt), tau)
    k = np.argmax(y_soft, axis=1)
    rows = np.arange(B)
    y_hard = np.zeros_like(y_soft)
    y_hard[rows, k] = 1.0
    s = (y_hard - y_soft) + y_soft  # straight-through value (Q4)
    mu_k = mu[rows, k]
    lv_k = logvar[rows, k]
    std = np.exp(0.5 * lv_k)
    eps = noise["eps"].astype(dt)
    z = mu_k + eps * std  # model.py:2924-2930
    y_full, dcache = decoder_trunk_forward(model, P, z, noise.get("dec_drop"), training)
    c = dt.type(-1.0 / B)  # d loss / d elbo_row
    up = np.full(B, c, dt)
    # heads: gradient of sum_b c * recon_b
    ho = heads_forward(model, P, y_full, s, data, masks, want_grad=want_grad, upstream=up)
    recon = ho["recon"]
    gauss_kl = kl_normal_vs_normal(mu_k, lv_k, pm[k], plv[k])
    total_unscaled = cat_kl + gauss_kl
    total_kl = dt.type(beta_scale) * total_unscaled
    elbo = recon - total_kl
    loss = -elbo.mean(dtype=dt)
    out = dict(loss=loss, reconstruction=recon, categorical_kl=cat_kl, gaussian_kl=gauss_kl,
               total_kl=total_kl, kl_unscaled=total_unscaled, latent=z, component_logits=logits,
               component_mu=mu, component_logvar=logvar, component_indices=k, assignments=s,
               y_soft=y_soft, y_full=y_full, per_feature=ho["per_feature"])
    if not want_grad:
        return out
    grads: Dict[str, np.ndarray] = dict(ho["wgrads"])
    bk = dt.type(beta_scale) / dt.type(B)  # d loss / d KL_row
    # decoder trunk backward
    dY = ho["dY"]
    hd = dcache["acts"][-1]
    grads["decoder.y_projection.weight"] = dY.T @ hd
    grads["decoder.y_projection.bias"] = dY.sum(axis=0)
    dh = dY @ P["decoder.y_projection.weight"]
    dz = _backbone_backward("decoder", model, P, dcache["acts"], dcache["pre"], dh, grads, True)
    if dlatent is not None:
        dz = dz + np.asarray(dlatent, dt)
    # Gaussian KL (selected component only) + reparameterisation
    vp_raw = np.exp(plv[k])
    vp = np.maximum(vp_raw, dt.type(1e-6))
    diff = mu_k - pm[k]
    e_lv = np.exp(lv_k)
    g_mu_k = dz + bk * diff / vp
    g_lv_k = dz * eps * (0.5 * std) + bk * 0.5 * (e_lv / vp - 1.0)
    g_pm_rows = -bk * diff / vp
    unclamped = (vp_raw >= dt.type(1e-6)).astype(dt)
    g_plv_rows = bk * 0.5 * (1.0 - unclamped * (e_lv + diff * diff) / vp)
    # straight-through Gumbel path + categorical KL
    dS = ho["dS"]
    inner = (y_soft * dS).sum(axis=1, keepdims=True)
    g_logits = (y_soft * (dS - inner)) / dt.type(tau)
    q = softmax(logits)
    log_q = log_softmax(logits)
    log_p = log_softmax(pl)
    g_logits = g_logits + bk * q * ((log_q - log_p[None]) - cat_kl[:, None])
    if model.behaviour == "supervised":
        p = np.exp(log_p)
        grads["prior.logits"] = (bk * (p[None] - q)).sum(axis=0)
        gpm = np.zeros((K, D), dt)
        gplv = np.zeros((K, D), dt)
        np.add.at(gpm, k, g_pm_rows)
        np.add.at(gplv, k, g_plv_rows)
        grads["prior.mu"] = gpm
        grads["prior.logvar"] = gplv
    # scatter into the full encoder-head gradients
    g_mu = np.zeros((B, K, D), dt)
    g_lv = np.zeros((B, K, D), dt)
    g_mu[rows, k] = g_mu_k
    lv_raw_k = ecache["lv_raw"][rows, k]
    in_range = ((lv_raw_k >= LOGVAR_RANGE[0]) & (lv_raw_k <= LOGVAR_RANGE[1])).astype(dt)
    g_lv[rows, k] = g_lv_k * in_range
    _encoder_backward(model, P, ecache, g_logits, g_mu.reshape(B, K * D), g_lv.reshape(B, K * D), grads)
    out["grads"] = grads
    out["dz"] = dz
    out["dS"] = dS
    out["dY"] = dY
    return out


def _encoder_backward(model, P, ecache, g_logits, g_mu, g_lv, grads):
    h = ecache["acts"][-1]
    grads["encoder.component_logits.weight"] = g_logits.T @ h
    grads["encoder.component_logits.bias"] = g_logits.sum(axis=0)
    grads["encoder.mu_layer.weight"] = g_mu.T @ h
    grads["encoder.mu_layer.bias"] = g_mu.sum(axis=0)
    grads["encoder.logvar_layer.weight"] = g_lv.T @ h
    grads["encoder.logvar_layer.bias"] = g_lv.sum(axis=0)
    dh = (g_logits @ P["encoder.component_logits.weight"]
          + g_mu @ P["encoder.mu_layer.weight"]
          + g_lv @ P["encoder.logvar_layer.weight"])
    _backbone_backward("encoder", model, P, ecache["acts"], ecache["pre"], dh, grads, False)


# ---------------------------------------------------------------------------
# unsupervised ELBO step  (model.py:1576-1580, 1619-1654)
# ---------------------------------------------------------------------------
def elbo_unsupervised(model, P, x_enc, data, masks, noise, *, beta_scale, temperature,
                      training=True, want_grad=True):
    """``_forward_elbo_batch`` for behaviour="unsupervised": all-K reparameterised
    latents, relaxed Gumbel weights, K decoder passes with one-hot assignments.

    ``noise["eps"]`` is ``[B,K,D]``; ``noise["dec_drop"]`` is a list over the K
    decoder passes of per-layer keep masks (the reference draws fresh dropout
    noise in every pass).
    """

    dt = x_enc.dtype
    B = x_enc.shape[0]
    K, D = model.K, model.D
    logits, mu, logvar, ecache = encoder_forward(model, P, x_enc, noise.get("enc_drop"), training)
    q = softmax(logits)
    if temperature is not None:
        tau = max(float(temperature), 1e-6)
        w = gumbel_softmax(logits, noise["exp"].astype(dt), tau)
    else:
        tau = None
        w = q
    pl, pm, plv = model.prior_tensors(P)
    cat_kl = kl_categorical(logits, pl)
    eps = noise["eps"].astype(dt)
    std = np.exp(0.5 * logvar)
    z_all = mu + eps * std
    c = dt.type(-1.0 / B)
    comp_lp = np.zeros((B, K), dt)
    caches = []
    for j in range(K):
        s = np.zeros((B, K), dt)
        s[:, j] = 1.0
        dm = noise.get("dec_drop")
        yj, dc = decoder_trunk_forward(model, P, z_all[:, j], dm[j] if dm else None, training)
        ho = heads_forward(model, P, yj, s, data, masks, want_grad=want_grad, upstream=c * w[:, j])
        comp_lp[:, j] = ho["recon"]
        caches.append((dc, ho))
    comp_kl = kl_normal_vs_normal(mu, logvar, pm[None], plv[None])  # [B,K]
    gauss_kl = (w * comp_kl).sum(axis=1)
    total_unscaled = cat_kl + gauss_kl
    total_kl = dt.type(beta_scale) * total_unscaled
    recon = (w * comp_lp).sum(axis=1)
    elbo = recon - total_kl
    loss = -elbo.mean(dtype=dt)
    latent = (w[:, :, None] * z_all).sum(axis=1)
    out = dict(loss=loss, reconstruction=recon, categorical_kl=cat_kl, gaussian_kl=gauss_kl,
               total_kl=total_kl, kl_unscaled=total_unscaled, latent=latent,
               component_logits=logits, component_mu=mu, component_logvar=logvar,
               posterior_weights=w, component_log_px=comp_lp)
    if not want_grad:
        return out
    grads: Dict[str, np.ndarray] = {}
    bk = dt.type(beta_scale) / dt.type(B)
    dz_all = np.zeros((B, K, D), dt)
    for j in range(K):
        dc, ho = caches[j]
        for name, gval in ho["wgrads"].items():
            grads[name] = grads.get(name, 0) + gval
        dY = ho["dY"]
        hd = dc["acts"][-1]
        grads["decoder.y_projection.weight"] = grads.get("decoder.y_projection.weight", 0) + dY.T @ hd
        grads["decoder.y_projection.bias"] = grads.get("decoder.y_projection.bias", 0) + dY.sum(axis=0)
        dh = dY @ P["decoder.y_projection.weight"]
        local: Dict[str, np.ndarray] = {}
        dz_all[:, j] = _backbone_backward("decoder", model, P, dc["acts"], dc["pre"], dh, local, True)
        for name, gval in local.items():
            grads[name] = grads.get(name, 0) + gval
    vp_raw = np.exp(plv)[None]
    vp = np.maximum(vp_raw, dt.type(1e-6))
    diff = mu - pm[None]
    e_lv = np.exp(logvar)
    wk = w[:, :, None]
    g_mu = dz_all + bk * wk * diff / vp
    g_lv = dz_all * eps * (0.5 * std) + bk * wk * 0.5 * (e_lv / vp - 1.0)
    in_range = ((ecache["lv_raw"] >= LOGVAR_RANGE[0]) & (ecache["lv_raw"] <= LOGVAR_RANGE[1])).astype(dt)
    g_lv = g_lv * in_range
    g_pm = (-bk * wk * diff / vp).sum(axis=0)  # [K,D]
    # d loss / d w_j = c * comp_lp_j + bk * comp_kl_j
    dw = c * comp_lp + bk * comp_kl
    if tau is not None:
        g_logits = (w * (dw - (w * dw).sum(axis=1, keepdims=True))) / dt.type(tau)
    else:
        g_logits = q * (dw - (q * dw).sum(axis=1, keepdims=True))
    log_q = log_softmax(logits)
    log_p = log_softmax(pl)
    g_logits = g_logits + bk * q * ((log_q - log_p[None]) - cat_kl[:, None])
    if model.behaviour == "supervised":
        p = np.exp(log_p)
        grads["prior.logits"] = (bk * (p[None] - q)).sum(axis=0)
        grads["prior.mu"] = g_pm
        unclamped = (vp_raw >= dt.type(1e-6)).astype(dt)
        grads["prior.logvar"] = (bk * wk * 0.5 * (1.0 - unclamped * (e_lv + diff * diff) / vp)).sum(axis=0)
    else:  # prior.py:20-39: mu_p[k,d] = W[d,k] + b[d]
        grads["prior_mean.linear.weight"] = g_pm.T.copy()
        grads["prior_mean.linear.bias"] = g_pm.sum(axis=0)
    _encoder_backward(model, P, ecache, g_logits, g_mu.reshape(B, K * D), g_lv.reshape(B, K * D), grads)
    out["grads"] = grads
    return out


# ---------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults; model.py:1443)
# ---------------------------------------------------------------------------
def adam_step(P, grads, state, *, lr, step, betas=(0.9, 0.999), eps=1e-8, lr_scale=None):
    """One torch.optim.Adam step (no weight decay / amsgrad) on every key of ``grads``."""

    b1, b2 = betas
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for name, g in grads.items():
        p = P[name]
        dt = p.dtype
        m = state.setdefault("m", {}).setdefault(name, np.zeros_like(p))
        v = state.setdefault("v", {}).setdefault(name, np.zeros_like(p))
        g = g.astype(dt).reshape(p.shape)
        m[...] = dt.type(b1) * m + dt.type(1 - b1) * g
        v[...] = dt.type(b2) * v + dt.type(1 - b2) * g * g
        scale = 1.0 if lr_scale is None else lr_scale.get(name, 1.0)
        step_size = dt.type(lr * scale / bc1)
        denom = np.sqrt(v) / dt.type(math.sqrt(bc2)) + dt.type(eps)
        P[name] = (p - step_size * (m / denom)).astype(dt)
    return P


# ---------------------------------------------------------------------------
# posterior statistics / encode  (model.py:2944-2964, 4238-4267)
# ---------------------------------------------------------------------------
def mixture_posterior_statistics(logits, mu, logvar, temperature=None):
    if temperature is not None:
        logits = logits / logits.dtype.type(max(float(temperature), 1e-6))
    p = softmax(logits)
    wmu = (p[:, :, None] * mu).sum(axis=1)
    second = (p[:, :, None] * (np.exp(logvar) + mu * mu)).sum(axis=1)
    var = np.maximum(second - wmu * wmu, mu.dtype.type(1e-6))
    return wmu, np.log(var), p


# ---------------------------------------------------------------------------
# rest of the supervised schedule (model.py:1866-2495)
# ---------------------------------------------------------------------------
def classifier_loss(latent, W, b, class_weight, targets):
    """``ClassificationHead`` without hidden layers (heads.py:78-91): logits = latent W^T + b and the weighted
    cross-entropy ``F.cross_entropy(logits, targets, weight)`` = sum_i w[t_i] * (-log p_i[t_i]) / sum_i w[t_i].
    Returns loss, logits and the gradients w.r.t. latent, W and b."""

    dt = latent.dtype
    logits = latent @ W.T + b
    lp = log_softmax(logits)
    n = latent.shape[0]
    rows = np.arange(n)
    w = np.asarray(class_weight, dt)[targets]
    loss = -(w * lp[rows, targets]).sum() / w.sum()
    g = np.exp(lp)
    g[rows, targets] -= 1.0
    g = g * (w / w.sum())[:, None]
    return loss, logits, g @ W, g.T @ latent, g.sum(axis=0)


def prior_anchor_gradient(P, anchors, lam):
    """Gradient of ``lam * sum_t mse(prior_t, anchor_t)`` (model.py:2338-2346, reduction="mean")."""

    return {k: (2.0 * lam / P[k].size) * (P[k] - anchors[k]) for k in anchors}


def closed_form_prior_update(probs, mu, logvar):
    """``_closed_form_prior_update_from_stats`` (model.py:2447-2495): prior logits, means and log-variances
    from posterior responsibilities."""

    dt = probs.dtype
    w = probs.sum(axis=0)
    total = float(w.sum())
    mixture = np.maximum(w / dt.type(total), dt.type(1e-6))
    cw = np.maximum(w[:, None], dt.type(1e-6))
    means = (probs[:, :, None] * mu).sum(axis=0) / cw
    second = (probs[:, :, None] * (np.exp(logvar) + mu * mu)).sum(axis=0) / cw
    var = np.maximum(second - means * means, dt.type(1e-6))
    return np.log(mixture), means, np.log(var)
