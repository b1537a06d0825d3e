#!/usr/bin/env python3
"""Generate golden vectors from the UNTOUCHED reference (build container only).

Usage (from the repo root, in the container that has /root/reference)::

    python tests/golden/make_golden.py

The reference source never leaves the build container; only the small ``.npz``
files written next to this script travel with the repository.  Each file holds
inputs (packed batch, masks, normalisation statistics), every parameter by its
``state_dict`` key, the recorded noise tensors in call order and
the reference's outputs (per-row ELBO terms, loss, all parameter gradients,
parameters after one Adam step, head parameters and sampled values).

Recipe follows SURVEY.md Appendix B: stub ``seaborn`` (only imported by the
Flask UI), disable distribution argument validation (quirk Q1), and replay
noise by wrapping ``torch.randn_like``, ``Tensor.exponential_``,
``F.dropout``, ``torch.multinomial`` and ``torch.poisson``.
"""

from __future__ import annotations

import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.distributions.Distribution.set_default_validate_args(False)
torch.set_num_threads(1)

from suave.model import SUAVE  # noqa: E402
from suave.modules.decoder import Decoder  # noqa: E402
from suave.modules.encoder import EncoderMLP  # noqa: E402
from suave.types import Schema  # noqa: E402
from suave import sampling as ref_sampling  # noqa: E402

from oracle import suave_oracle as orc  # noqa: E402  (test infrastructure)
from oracle import sampling_oracle as sorc  # noqa: E402


# ---------------------------------------------------------------------------
# noise recorder
# ---------------------------------------------------------------------------
class NoiseTape:
    """Record (or replay) every random draw the reference makes, in call order."""

    def __init__(self):
        self.randn = []
        self.exp = []
        self.dropout = []
        self.multinomial_exp = []
        self.poisson_u = []
        self.poisson_n = []
        self._orig = {}

    def __enter__(self):
        tape = self
        self._orig["randn_like"] = torch.randn_like
        self._orig["exponential_"] = torch.Tensor.exponential_
        self._orig["dropout"] = F.dropout
        self._orig["multinomial"] = torch.multinomial
        self._orig["poisson"] = torch.poisson

        def randn_like(t, *a, **k):
            out = tape._orig["randn_like"](t, *a, **k)
            tape.randn.append(out.detach().clone())
            return out

        def exponential_(t, *a, **k):
            out = tape._orig["exponential_"](t, *a, **k)
            tape.exp.append(t.detach().clone())
            return out

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            keep = (torch.rand_like(x) >= p).to(x.dtype)
            tape.dropout.append(keep.detach().clone())
            return x * keep / (1.0 - p)

        def multinomial(probs, num_samples, replacement=False, *, generator=None):
            # torch's single-draw path: q ~ Exp(1); argmax(probs / q)
            if num_samples == 1:
                q = torch.empty_like(probs)
                tape._orig["exponential_"](q)
                tape.multinomial_exp.append(q.detach().clone())
                return torch.argmax(probs / q, dim=-1, keepdim=True)
            # Categorical.sample((L,)) (model.py:3322-3323): torch inverts the CDF with float64 uniforms,
            # which cannot be injected in binary32; the build defines the draw as L independent
            # exponential races and the same map replaces the primitive here (like torch.poisson below)
            assert replacement
            q = torch.empty((num_samples,) + tuple(probs.shape), dtype=probs.dtype)
            tape._orig["exponential_"](q)
            tape.multinomial_exp.append(q.detach().clone())
            return torch.argmax(probs.unsqueeze(0) / q, dim=-1).T.contiguous()

        def poisson(rate, generator=None):
            # build-defined (uniform, normal) -> count map (torch.poisson consumes
            # a data-dependent number of uniforms and cannot be replayed)
            u = torch.rand_like(rate)
            n = tape._orig["randn_like"](rate)
            tape.poisson_u.append(u.detach().clone())
            tape.poisson_n.append(n.detach().clone())
            cnt = sorc.poisson_from_noise(
                rate.detach().numpy().astype(np.float32),
                u.numpy().astype(np.float32), n.numpy().astype(np.float32))
            return torch.from_numpy(cnt.astype(np.float32)).to(rate.dtype)

        torch.randn_like = randn_like
        torch.Tensor.exponential_ = exponential_
        F.dropout = dropout
        torch.nn.functional.dropout = dropout
        torch.multinomial = multinomial
        torch.poisson = poisson
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig["randn_like"]
        torch.Tensor.exponential_ = self._orig["exponential_"]
        F.dropout = self._orig["dropout"]
        torch.nn.functional.dropout = self._orig["dropout"]
        torch.multinomial = self._orig["multinomial"]
        torch.poisson = self._orig["poisson"]
        return False


def check_multinomial_equivalence():
    """torch.multinomial(p, 1) == argmax(p / Exp(1)) with the same generator state."""

    torch.manual_seed(123)
    p = torch.rand(257, 7)
    p = p / p.sum(-1, keepdim=True)
    torch.manual_seed(99)
    a = torch.multinomial(p, 1, True)
    torch.manual_seed(99)
    q = torch.empty_like(p).exponential_()
    b = torch.argmax(p / q, dim=-1, keepdim=True)
    assert torch.equal(a, b), "exponential-race equivalence of torch.multinomial failed"


# ---------------------------------------------------------------------------
# synthetic schemas / batches
# ---------------------------------------------------------------------------
def make_columns(kind: str):
    if kind == "mixed7":
        return [
            orc.Column("r0", "real", stats={"mean": 50.0, "std": 12.0}),
            orc.Column("c0", "cat", n_classes=3),
            orc.Column("p0", "pos", stats={"mean_log": 0.9, "std_log": 0.45}),
            orc.Column("n0", "count", stats={"offset": 1.0}),
            orc.Column("o0", "ordinal", n_classes=4),
            orc.Column("r1", "real", stats={"mean": 0.0, "std": 1.0}),
            orc.Column("c1", "cat", n_classes=5),
        ]
    if kind == "mixed12":
        cols = []
        for i in range(3):
            cols.append(orc.Column(f"r{i}", "real", stats={"mean": float(i), "std": 1.0 + 0.5 * i}))
            cols.append(orc.Column(f"p{i}", "pos", stats={"mean_log": 1.0 - 0.2 * i, "std_log": 0.5 + 0.1 * i}))
        cols.append(orc.Column("n0", "count", stats={"offset": 1.0}))
        cols.append(orc.Column("n1", "count", stats={"offset": 0.0}))
        cols.append(orc.Column("c0", "cat", n_classes=2))
        cols.append(orc.Column("c1", "cat", n_classes=8))
        cols.append(orc.Column("o0", "ordinal", n_classes=3))
        cols.append(orc.Column("o1", "ordinal", n_classes=5))
        return cols
    if kind == "ydim":  # explicit y_dim overrides (types.py:24-33)
        return [
            orc.Column("r0", "real", y_dim=2, stats={"mean": 1.0, "std": 2.0}),
            orc.Column("c0", "cat", n_classes=4, y_dim=3),
            orc.Column("p0", "pos", y_dim=2, stats={"mean_log": 0.3, "std_log": 0.8}),
            orc.Column("n0", "count", y_dim=2, stats={"offset": 1.0}),
            orc.Column("o0", "ordinal", n_classes=4, y_dim=1),
        ]
    if kind == "wide7":  # two wide categoricals (>= 17 classes: GEMM head path) between narrow columns
        return [
            orc.Column("r0", "real", stats={"mean": 3.0, "std": 2.0}),
            orc.Column("w0", "cat", n_classes=40),
            orc.Column("p0", "pos", stats={"mean_log": 0.9, "std_log": 0.45}),
            orc.Column("c0", "cat", n_classes=3),
            orc.Column("o0", "ordinal", n_classes=5),
            orc.Column("w1", "cat", n_classes=17),
            orc.Column("n0", "count", stats={"offset": 1.0}),
        ]
    raise ValueError(kind)


def make_batch(cols, B, mask_rate, rng):
    n_num = sum(c.type in ("real", "pos", "count") for c in cols)
    n_disc = len(cols) - n_num
    xnum = np.zeros((B, n_num), np.float32)
    codes = np.zeros((B, n_disc), np.int16)
    mask = (rng.random((B, len(cols))) >= mask_rate).astype(np.uint8)
    i_num = i_disc = 0
    for j, c in enumerate(cols):
        if c.type == "real":
            xnum[:, i_num] = rng.standard_normal(B)
        elif c.type == "pos":
            raw = rng.lognormal(1.0, 0.5, B)
            xnum[:, i_num] = (np.log1p(raw) - c.stats["mean_log"]) / c.stats["std_log"]
        elif c.type == "count":
            cnt = rng.poisson(3.0, B)
            off = c.stats["offset"]
            if off == 0.0:
                cnt = cnt + 1
            xnum[:, i_num] = np.log(cnt + off).astype(np.float32)
        if c.type in ("real", "pos", "count"):
            xnum[mask[:, j] == 0, i_num] = 0.0  # Q7: masked cells are zero-filled
            i_num += 1
        else:
            codes[:, i_disc] = rng.integers(0, c.n_classes, B)
            codes[mask[:, j] == 0, i_disc] = -1
            i_disc += 1
    return xnum, codes, mask


def build_reference(cols, K, D, hidden, dropout, behaviour, seed):
    schema = Schema({
        c.name: {"type": c.type,
                 **({"n_classes": c.n_classes} if c.n_classes else {}),
                 **({"y_dim": c.y_dim} if c.y_dim else {})}
        for c in cols
    })
    model = SUAVE(schema, behaviour=behaviour, latent_dim=D, n_components=K,
                  hidden_dims=hidden, dropout=dropout)
    om = orc.OracleModel(cols, n_components=K, latent_dim=D, hidden=hidden,
                         dropout=dropout, behaviour=behaviour)
    torch.manual_seed(seed)
    model._encoder = EncoderMLP(om.I, D, hidden=hidden, dropout=dropout, n_components=K)
    model._decoder = Decoder(D, schema, hidden=hidden, dropout=dropout, n_components=K)
    model._norm_stats_per_col = {c.name: dict(c.stats) for c in cols}
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        # make head weights / prior O(1) so every term is exercised
        for p in model._decoder.heads.parameters():
            p.copy_(0.5 * torch.randn(p.shape, generator=g))
        if behaviour == "supervised":
            model._prior_component_logits.copy_(0.3 * torch.randn(K, generator=g))
            model._prior_component_mu.copy_(0.3 * torch.randn(K, D, generator=g))
            model._prior_component_logvar.copy_(0.2 * torch.randn(K, D, generator=g))
        else:
            model._prior_mean_layer.linear.weight.copy_(0.3 * torch.randn(D, K, generator=g))
            model._prior_mean_layer.linear.bias.copy_(0.1 * torch.randn(D, generator=g))
    return model, om


def ref_params(model, behaviour):
    P = {}
    for k, v in model._encoder.state_dict().items():
        P["encoder." + k] = v.detach().numpy().copy()
    for k, v in model._decoder.state_dict().items():
        P["decoder." + k] = v.detach().numpy().copy()
    if behaviour == "supervised":
        P["prior.logits"] = model._prior_component_logits.detach().numpy().copy()
        P["prior.mu"] = model._prior_component_mu.detach().numpy().copy()
        P["prior.logvar"] = model._prior_component_logvar.detach().numpy().copy()
    else:
        for k, v in model._prior_mean_layer.state_dict().items():
            P["prior_mean." + k] = v.detach().numpy().copy()
    return P


def ref_named_parameters(model, behaviour):
    named = []
    for k, p in model._encoder.named_parameters():
        named.append(("encoder." + k, p))
    for k, p in model._decoder.named_parameters():
        named.append(("decoder." + k, p))
    if behaviour == "supervised":
        named.append(("prior.logits", model._prior_component_logits))
        named.append(("prior.mu", model._prior_component_mu))
        named.append(("prior.logvar", model._prior_component_logvar))
    else:
        for k, p in model._prior_mean_layer.named_parameters():
            named.append(("prior_mean." + k, p))
    return named


def to_torch_batches(om, xnum, codes, mask):
    x_enc, data, masks = orc.dense_from_packed(om, xnum, codes, mask)
    data_t = {t: {} for t in orc.TYPE_ORDER}
    mask_t = {t: {} for t in orc.TYPE_ORDER}
    for c in om.columns:
        data_t[c.type][c.name] = torch.from_numpy(data[c.name])
        mask_t[c.type][c.name] = torch.from_numpy(masks[c.name])
    return torch.from_numpy(x_enc), data_t, mask_t


def cols_json(cols):
    return json.dumps([
        {"name": c.name, "type": c.type, "n_classes": c.n_classes, "y_dim": c.y_dim, "stats": c.stats}
        for c in cols
    ])


# ---------------------------------------------------------------------------
# fixture writers
# ---------------------------------------------------------------------------
def gen_elbo(name, kind, K, D, hidden, B, mask_rate, dropout, behaviour, seed,
             beta_scale=0.75, tau=1.0, temperature=None, lr=1e-3):
    rng = np.random.default_rng(seed)
    cols = make_columns(kind)
    model, om = build_reference(cols, K, D, hidden, dropout, behaviour, seed)
    model.gumbel_temperature = tau
    xnum, codes, mask = make_batch(cols, B, mask_rate, rng)
    x_enc, data_t, mask_t = to_torch_batches(om, xnum, codes, mask)
    P0 = ref_params(model, behaviour)
    model._encoder.train()
    model._decoder.train()
    torch.manual_seed(seed + 7)
    named = ref_named_parameters(model, behaviour)
    opt = torch.optim.Adam([p for _, p in named], lr=lr)
    with NoiseTape() as tape:
        out = model._forward_elbo_batch(x_enc, data_t, mask_t, beta_scale=beta_scale,
                                        temperature=temperature)
        opt.zero_grad()
        out["loss"].backward()
    grads = {k: (p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(p.shape, np.float32))
             for k, p in named}
    opt.step()
    P1 = ref_params(model, behaviour)
    n_hidden = len(hidden)
    rec = {
        "meta": json.dumps(dict(kind=kind, K=K, D=D, hidden=list(hidden), B=B, mask_rate=mask_rate,
                                dropout=dropout, behaviour=behaviour, beta_scale=beta_scale, tau=tau,
                                temperature=temperature, lr=lr, seed=seed)),
        "columns": cols_json(cols),
        "xnum": xnum, "codes": codes, "mask": mask,
        "noise/exp": tape.exp[0].numpy() if tape.exp else np.zeros((B, K), np.float32),
    }
    if behaviour == "supervised":
        rec["noise/eps"] = tape.randn[0].numpy()
    else:
        rec["noise/eps"] = tape.randn[0].numpy()  # [B,K,D]
    if dropout > 0:
        for i in range(n_hidden):
            rec[f"noise/enc_drop{i}"] = tape.dropout[i].numpy().astype(np.uint8)
        n_pass = 1 if behaviour == "supervised" else K
        for j in range(n_pass):
            for i in range(n_hidden):
                rec[f"noise/dec_drop{j}_{i}"] = tape.dropout[n_hidden + j * n_hidden + i].numpy().astype(np.uint8)
    for k, v in P0.items():
        rec["param/" + k] = v
    for k, v in grads.items():
        rec["grad/" + k] = v
    for k, v in P1.items():
        rec["adam1/" + k] = v
    for key in ("loss", "reconstruction", "categorical_kl", "gaussian_kl", "total_kl",
                "kl_unscaled", "latent", "component_logits", "component_mu", "component_logvar"):
        rec["out/" + key] = out[key].detach().numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name, "loss", float(out["loss"]))


def gen_joint(name, kind, K, D, hidden, B, mask_rate, seed, n_classes=3, cls_weight=0.7):
    """One joint fine-tuning step (model.py:1973-1999): ELBO loss + weight * cross-entropy of
    ``classifier(outputs["latent"])``; gradients of the JOINT loss on every parameter."""

    from suave.modules.heads import ClassificationHead

    rng = np.random.default_rng(seed)
    cols = make_columns(kind)
    model, om = build_reference(cols, K, D, hidden, 0.0, "supervised", seed)
    torch.manual_seed(seed + 3)
    clf = ClassificationHead(D, n_classes, class_weight=[1.0, 2.0, 0.5][:n_classes])
    xnum, codes, mask = make_batch(cols, B, mask_rate, rng)
    x_enc, data_t, mask_t = to_torch_batches(om, xnum, codes, mask)
    targets = torch.from_numpy(rng.integers(0, n_classes, B).astype(np.int64))
    P0 = ref_params(model, "supervised")
    model._encoder.train()
    model._decoder.train()
    clf.train()
    named = ref_named_parameters(model, "supervised")
    with NoiseTape() as tape:
        out = model._forward_elbo_batch(x_enc, data_t, mask_t, beta_scale=model.beta, temperature=None)
        logits = clf(out["latent"])
        class_loss = clf.loss(logits, targets)
        joint = out["loss"] + class_loss * cls_weight
        for _, p in named:
            p.grad = None
        joint.backward()
    rec = {
        "meta": json.dumps(dict(kind=kind, K=K, D=D, hidden=list(hidden), B=B, mask_rate=mask_rate, dropout=0.0,
                                behaviour="supervised", beta_scale=float(model.beta), tau=1.0, seed=seed,
                                n_classes=n_classes, cls_weight=cls_weight)),
        "columns": cols_json(cols),
        "xnum": xnum, "codes": codes, "mask": mask,
        "noise/exp": tape.exp[0].numpy(), "noise/eps": tape.randn[0].numpy(),
        "targets": targets.numpy(),
        "clf/linear.weight": clf.linear.weight.detach().numpy(), "clf/linear.bias": clf.linear.bias.detach().numpy(),
        "clf/class_weight": clf.class_weight.numpy(),
        "clf_grad/linear.weight": clf.linear.weight.grad.numpy(), "clf_grad/linear.bias": clf.linear.bias.grad.numpy(),
        "out/class_loss": class_loss.detach().numpy(), "out/joint_loss": joint.detach().numpy(),
        "out/logits": logits.detach().numpy(),
    }
    for k, v in P0.items():
        rec["param/" + k] = v
    for k, p in named:
        rec["grad/" + k] = p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(p.shape, np.float32)
    for key in ("loss", "reconstruction", "categorical_kl", "gaussian_kl", "latent"):
        rec["out/" + key] = out[key].detach().numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name, "joint loss", float(joint))


def gen_decode(name, kind, K, D, hidden, B, seed):
    """Decoder forward on placeholder batches (sampling path, model.py:4397-4420):
    head params, log-likelihoods with both masks, injected draws and samples."""

    rng = np.random.default_rng(seed)
    cols = make_columns(kind)
    model, om = build_reference(cols, K, D, hidden, 0.0, "supervised", seed)
    model._decoder.eval()
    xnum, codes, mask = make_batch(cols, B, 0.3, rng)
    _, data_t, mask_t = to_torch_batches(om, xnum, codes, mask)
    z = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32))
    kk = rng.integers(0, K, B)
    s = F.one_hot(torch.from_numpy(kk), K).float()
    with NoiseTape() as tape, torch.no_grad():
        out = model._decoder(z, s, data_t, model._norm_stats_per_col, mask_t)
        frame = ref_sampling.decoder_outputs_to_frame(out["per_feature"], model.schema,
                                                      model._norm_stats_per_col)
    rec = {"columns": cols_json(cols),
           "meta": json.dumps(dict(kind=kind, K=K, D=D, hidden=list(hidden), B=B, seed=seed)),
           "xnum": xnum, "codes": codes, "mask": mask, "z": z.numpy(), "k": kk.astype(np.int64)}
    for k_, v in ref_params(model, "supervised").items():
        rec["param/" + k_] = v
    i_rn = i_mn = i_ps = 0
    for c in cols:
        pf = out["per_feature"][c.name]
        rec[f"out/{c.name}/log_px"] = pf["log_px"].numpy()
        rec[f"out/{c.name}/log_px_missing"] = pf["log_px_missing"].numpy()
        for pk, pv in pf["params"].items():
            rec[f"out/{c.name}/params/{pk}"] = pv.numpy()
        rec[f"out/{c.name}/sample"] = pf["sample"].numpy()
        col_vals = frame[c.name]
        if c.type in ("cat", "ordinal"):
            rec[f"out/{c.name}/frame"] = np.asarray(col_vals.cat.codes, dtype=np.int64)
        else:
            rec[f"out/{c.name}/frame"] = col_vals.to_numpy()
        if c.type in ("real", "pos"):
            rec[f"noise/{c.name}/normal"] = tape.randn[i_rn].numpy()
            i_rn += 1
        elif c.type in ("cat", "ordinal"):
            rec[f"noise/{c.name}/exp"] = tape.multinomial_exp[i_mn].numpy()
            i_mn += 1
        else:
            rec[f"noise/{c.name}/uniform"] = tape.poisson_u[i_ps].numpy()
            rec[f"noise/{c.name}/normal"] = tape.poisson_n[i_ps].numpy()
            i_ps += 1
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name)


def gen_posterior(name, kind, K, D, hidden, B, seed, L=8):
    """encode / posterior statistics (model.py:2944-2964), impute routing
    (model.py:3618-3644) and MC posterior draws (model.py:3304-3336)."""

    rng = np.random.default_rng(seed)
    cols = make_columns(kind)
    model, om = build_reference(cols, K, D, hidden, 0.0, "supervised", seed)
    model._encoder.eval()
    xnum, codes, mask = make_batch(cols, B, 0.3, rng)
    x_enc, _, _ = to_torch_batches(om, xnum, codes, mask)
    with torch.no_grad():
        lg, mu, lv = model._encoder(x_enc)
        pm, plv, pp = SUAVE._mixture_posterior_statistics(lg, mu, lv, temperature=None)
        pm_t, plv_t, pp_t = SUAVE._mixture_posterior_statistics(lg, mu, lv, temperature=0.5)
    rec = {"columns": cols_json(cols),
           "meta": json.dumps(dict(kind=kind, K=K, D=D, hidden=list(hidden), B=B, seed=seed, L=L)),
           "xnum": xnum, "codes": codes, "mask": mask,
           "out/logits": lg.numpy(), "out/mu": mu.numpy(), "out/logvar": lv.numpy(),
           "out/post_mean": pm.numpy(), "out/post_logvar": plv.numpy(), "out/post_probs": pp.numpy(),
           "out/post_mean_t05": pm_t.numpy(), "out/post_logvar_t05": plv_t.numpy(),
           "out/post_probs_t05": pp_t.numpy()}
    for k_, v in ref_params(model, "supervised").items():
        rec["param/" + k_] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name)


def gen_predictive(name, kind, K, D, hidden, B, seed, L=6):
    """Monte-Carlo posterior predictive of every attribute (model.py:3304-3421): ``_posterior_predictive_params``
    (L component / latent draws per row, decoder trunk on L*B rows, ONE head) and ``_draw_predictive_samples``."""

    rng = np.random.default_rng(seed)
    cols = make_columns(kind)
    model, om = build_reference(cols, K, D, hidden, 0.0, "supervised", seed)
    model._encoder.eval()
    model._decoder.eval()
    xnum, codes, mask = make_batch(cols, B, 0.3, rng)
    x_enc, data_t, mask_t = to_torch_batches(om, xnum, codes, mask)
    rec = {"columns": cols_json(cols),
           "meta": json.dumps(dict(kind=kind, K=K, D=D, hidden=list(hidden), B=B, seed=seed, L=L)),
           "xnum": xnum, "codes": codes, "mask": mask}
    for k_, v in ref_params(model, "supervised").items():
        rec["param/" + k_] = v
    for c in cols:
        attr_data = data_t[c.type][c.name]
        attr_mask = torch.zeros_like(mask_t[c.type][c.name])  # the attribute is decoded as missing (model.py:3300-3301)
        with NoiseTape() as tape, torch.no_grad():
            res = model._posterior_predictive_params(x_enc, c.name, c.type, attr_data, attr_mask, L)
            samples = None
            if c.type in ("real", "pos", "count"):
                samples = model._draw_predictive_samples(c.type, res["params"], L, B, torch.device("cpu"))
        rec[f"noise/{c.name}/comp_exp"] = tape.multinomial_exp[0].numpy()   # [L,B,K]
        rec[f"noise/{c.name}/eps"] = tape.randn[0].numpy()                  # [L,B,D]
        for pk, pv in res["params"].items():
            rec[f"out/{c.name}/params/{pk}"] = pv.numpy()
        # order of the tape: posterior draw, eps, the head's own (discarded) sample draw, then the predictive draw
        if c.type in ("real", "pos"):
            rec[f"noise/{c.name}/normal"] = tape.randn[2].numpy().reshape(L, B)
        elif c.type == "count":
            rec[f"noise/{c.name}/uniform"] = tape.poisson_u[1].numpy().reshape(L, B)
            rec[f"noise/{c.name}/normal"] = tape.poisson_n[1].numpy().reshape(L, B)
        if samples is not None:
            rec[f"out/{c.name}/samples"] = samples.numpy()                   # [L,B]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name)


def gen_impute(name, seed, n=48, K=3, D=5, hidden=(16, 8)):
    """SUAVE.impute end to end (model.py:3512-3692) on a raw frame with NaNs, for the three assignment
    strategies: the (latent, assignments) handed to Decoder.forward (what the reference's own
    tests/test_impute.py:107-146 spies on), the injected draws, the reconstruction and the merged frame."""

    import pandas as pd
    from unittest.mock import patch
    from suave import data as ref_data

    rng = np.random.default_rng(seed)
    raw = {
        "r0": rng.normal(50, 12, n), "p0": rng.lognormal(1.0, 0.5, n), "n0": rng.poisson(3.0, n).astype(float),
        "c0": rng.integers(0, 3, n).astype(float), "o0": rng.integers(0, 4, n).astype(float),
        "r1": rng.normal(0, 1, n), "n1": (rng.poisson(2.0, n) + 1).astype(float), "c1": rng.integers(0, 5, n).astype(float),
    }
    for k in raw:
        raw[k][rng.random(n) < 0.25] = np.nan
    X = pd.DataFrame(raw)
    spec = {"r0": {"type": "real"}, "p0": {"type": "pos"}, "n0": {"type": "count"},
            "c0": {"type": "cat", "n_classes": 3}, "o0": {"type": "ordinal", "n_classes": 4},
            "r1": {"type": "real"}, "n1": {"type": "count"}, "c1": {"type": "cat", "n_classes": 5}}
    schema = Schema(spec)
    _, stats = ref_data.standardize(X, schema)
    cols = [orc.Column(nm, sp["type"], sp.get("n_classes"), None,
                       {k: float(v) for k, v in stats[nm].items() if isinstance(v, (int, float, np.floating, np.integer))})
            for nm, sp in spec.items()]
    rec = {"schema": json.dumps(schema.to_dict()),
           "stats": json.dumps({k: {kk: (vv if not isinstance(vv, (np.floating, np.integer)) else float(vv))
                                    for kk, vv in v.items()} for k, v in stats.items()}),
           "columns": cols_json(cols)}
    for k, v in raw.items():
        rec["raw/" + k] = v
    for behaviour in ("supervised", "unsupervised"):
        model, om = build_reference(cols, K, D, hidden, 0.0, behaviour, seed)
        model._norm_stats_per_col = stats
        model._is_fitted = True
        model._inference_tau = 0.7
        model.gumbel_temperature = 0.9
        for k_, v in ref_params(model, behaviour).items():
            rec[f"{behaviour}/param/{k_}"] = v
        for strategy in ("hard", "soft", "sample"):
            with NoiseTape() as tape, patch.object(model._decoder, "forward", wraps=model._decoder.forward) as spy:
                recon = model.impute(X, only_missing=False, assignment_strategy=strategy)
                latent_arg, assign_arg = spy.call_args[0][:2]
            pre = f"{behaviour}/{strategy}/"
            rec[pre + "latent"] = latent_arg.detach().numpy()
            rec[pre + "assignments"] = assign_arg.detach().numpy()
            if strategy == "sample":
                rec[pre + "noise/assign_exp"] = tape.exp[0].numpy()
            i_rn = i_mn = i_ps = 0
            for c in cols:
                if c.type in ("real", "pos"):
                    rec[pre + f"noise/{c.name}/normal"] = tape.randn[i_rn].numpy()
                    i_rn += 1
                elif c.type in ("cat", "ordinal"):
                    rec[pre + f"noise/{c.name}/exp"] = tape.multinomial_exp[i_mn].numpy()
                    i_mn += 1
                else:
                    rec[pre + f"noise/{c.name}/uniform"] = tape.poisson_u[i_ps].numpy()
                    rec[pre + f"noise/{c.name}/normal"] = tape.poisson_n[i_ps].numpy()
                    i_ps += 1
                v = recon[c.name]
                rec[pre + f"recon/{c.name}"] = (np.asarray(v.cat.codes, np.int64) if c.type in ("cat", "ordinal")
                                                else v.to_numpy())
            # merged frame (only_missing=True) follows from the reconstruction: observed cells keep the input
            with NoiseTape(), torch.no_grad():
                torch.manual_seed(seed)
                merged = model.impute(X, only_missing=True, assignment_strategy="hard")
            if strategy == "hard":
                for c in cols:
                    v = merged[c.name]
                    obs = ~X[c.name].isna().to_numpy()
                    got = np.asarray(v.cat.codes if c.type in ("cat", "ordinal") else v, np.float64)
                    rec[f"{behaviour}/merged_observed/{c.name}"] = got[obs]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name)


def gen_tensorise(name, seed, n=40):
    """standardize -> _apply_training_normalization -> _prepare_training_tensors on a frame with NaNs
    (data.py:133-266, model.py:3116-3213, 2784-2922): raw columns in, encoder_inputs + stats out."""

    import pandas as pd
    from suave import data as ref_data

    rng = np.random.default_rng(seed)
    raw = {
        "r0": rng.normal(50, 12, n), "p0": rng.lognormal(1.0, 0.5, n), "n0": rng.poisson(3.0, n).astype(float),
        "c0": rng.integers(0, 3, n).astype(float), "o0": rng.integers(0, 4, n).astype(float),
        "r1": rng.normal(0, 1, n), "n1": (rng.poisson(2.0, n) + 1).astype(float), "c1": rng.integers(0, 5, n).astype(float),
    }
    for k in raw:
        raw[k][rng.random(n) < 0.2] = np.nan
    X = pd.DataFrame(raw)
    schema = Schema({"r0": {"type": "real"}, "p0": {"type": "pos"}, "n0": {"type": "count"},
                     "c0": {"type": "cat", "n_classes": 3}, "o0": {"type": "ordinal", "n_classes": 4},
                     "r1": {"type": "real"}, "n1": {"type": "count"}, "c1": {"type": "cat", "n_classes": 5}})
    _, stats = ref_data.standardize(X, schema)
    model = SUAVE(schema, latent_dim=4, n_components=2)
    model._norm_stats_per_col = stats
    enc = model._prepare_inference_inputs(X)
    rec = {"enc_inputs": enc.numpy(), "schema": json.dumps(schema.to_dict()),
           "stats": json.dumps({k: {kk: (vv if not isinstance(vv, (np.floating, np.integer)) else float(vv))
                                    for kk, vv in v.items()} for k, v in stats.items()})}
    for k, v in raw.items():
        rec["raw/" + k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name)


def main():
    only = set(sys.argv[1:])  # optional fixture names: regenerate just those

    def want(name):
        return not only or name in only

    check_multinomial_equivalence()
    jobs = [
        (gen_tensorise, ("tensorise_mixed8", 51), {}),
        (gen_elbo, ("elbo_sup_mixed7_k3", "mixed7", 3, 6, (16, 8), 64, 0.3, 0.0, "supervised", 11), {}),
        (gen_elbo, ("elbo_sup_mixed7_k1", "mixed7", 1, 2, (8,), 4, 0.0, 0.0, "supervised", 12), {}),
        (gen_elbo, ("elbo_sup_mixed12_drop", "mixed12", 3, 6, (32, 16), 64, 0.3, 0.1, "supervised", 13), {"tau": 0.7}),
        (gen_elbo, ("elbo_sup_ydim", "ydim", 2, 4, (16, 8), 32, 0.2, 0.0, "supervised", 14), {}),
        (gen_elbo, ("elbo_unsup_mixed7_k3", "mixed7", 3, 6, (16, 8), 64, 0.3, 0.0, "unsupervised", 21), {"temperature": 0.8}),
        (gen_elbo, ("elbo_unsup_mixed12_notemp", "mixed12", 2, 4, (16, 8), 32, 0.3, 0.0, "unsupervised", 22),
         {"temperature": None}),
        (gen_elbo, ("elbo_unsup_mixed7_drop", "mixed7", 3, 6, (16, 8), 16, 0.3, 0.1, "unsupervised", 23), {"temperature": 0.9}),
        (gen_decode, ("decode_mixed12", "mixed12", 3, 6, (16, 8), 64, 31), {}),
        (gen_decode, ("decode_ydim", "ydim", 2, 4, (16, 8), 32, 32), {}),
        (gen_posterior, ("posterior_mixed7", "mixed7", 3, 6, (16, 8), 64, 41), {}),
        # wide categorical columns (BASELINE config #4 in miniature)
        (gen_elbo, ("elbo_sup_wide7", "wide7", 3, 6, (16, 8), 48, 0.3, 0.0, "supervised", 61), {}),
        (gen_elbo, ("elbo_unsup_wide7", "wide7", 2, 4, (16, 8), 24, 0.3, 0.0, "unsupervised", 62), {"temperature": 0.8}),
        (gen_decode, ("decode_wide7", "wide7", 3, 6, (16, 8), 32, 63), {}),
        (gen_joint, ("joint_sup_mixed7", "mixed7", 3, 6, (16, 8), 48, 0.3, 71), {}),
        (gen_predictive, ("predictive_mixed7", "mixed7", 3, 6, (16, 8), 24, 81), {}),
        (gen_impute, ("impute_mixed8", 91), {}),
    ]
    for fn, args, kw in jobs:
        if want(args[0]):
            fn(*args, **kw)


if __name__ == "__main__":
    main()
