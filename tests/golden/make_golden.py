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
        self.randint_u = []
        self.prior_u = []
        self.perms = []
        # Categorical.sample((n,)) from ONE distribution (sampling.py:49-50): "race" = n exponential races
        # (posterior draws), "cdf" = inversion of the prior's cumulative distribution on binary32 uniforms
        # (the build's suave_prior_latents map); `prior_logits` must then hold the distribution's logits
        self.multi_mode = "race"
        self.prior_logits = None
        self._orig = {}

    def __enter__(self):
        tape = self
        self._orig["randn_like"] = torch.randn_like
        self._orig["exponential_"] = torch.Tensor.exponential_
        self._orig["dropout"] = F.dropout
        self._orig["multinomial"] = torch.multinomial
        self._orig["poisson"] = torch.poisson
        self._orig["randint"] = torch.randint
        self._orig["randperm"] = torch.randperm

        def randint(low, high, size, *a, **k):
            # torch.randint(0, n, (1,)) of _draw_conditional_latents (model.py:3088): integer draws cannot be replayed
            # from a binary32 uniform, so the build defines pick = min(int(u * n), n - 1) and the same map replaces
            # the primitive here (like torch.poisson below)
            assert low == 0 and tuple(size) == (1,)
            u = torch.rand(1)
            tape.randint_u.append(u.clone())
            pick = min(int(np.float32(u.numpy()[0]) * np.float32(high)), int(high) - 1)
            return torch.tensor([pick], dtype=torch.long)

        def randperm(n, *a, **k):
            out = tape._orig["randperm"](n, *a, **k)
            tape.perms.append(out.detach().clone())
            return out

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
            if tape.multi_mode == "cdf":
                assert probs.shape[0] == 1 and tape.prior_logits is not None
                u = torch.rand(num_samples)
                tape.prior_u.append(u.clone())
                k = sorc.component_from_uniform(sorc.prior_cdf(tape.prior_logits), u.numpy().astype(np.float32))
                return torch.from_numpy(k).view(1, -1)
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
        torch.randint = randint
        torch.randperm = randperm
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig["randn_like"]
        torch.Tensor.exponential_ = self._orig["exponential_"]
        F.dropout = self._orig["dropout"]
        torch.nn.functional.dropout = self._orig["dropout"]
        torch.multinomial = self._orig["multinomial"]
        torch.poisson = self._orig["poisson"]
        torch.randint = self._orig["randint"]
        torch.randperm = self._orig["randperm"]
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
    if kind == "bigcount6":  # count heads whose rates cover every range of the build's Poisson map (gen_decode boosts them)
        return [
            orc.Column("r0", "real", stats={"mean": 3.0, "std": 2.0}),
            orc.Column("n0", "count", stats={"offset": 1.0}),
            orc.Column("n1", "count", stats={"offset": 1.0}),
            orc.Column("c0", "cat", n_classes=3),
            orc.Column("n2", "count", stats={"offset": 0.0}),
            orc.Column("n3", "count", stats={"offset": 1.0}),
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


def gen_decode(name, kind, K, D, hidden, B, seed, rate_boost=None):
    """Decoder forward on placeholder batches (sampling path, model.py:4397-4420):
    head params, log-likelihoods with both masks, injected draws and samples.
    ``rate_boost`` {count column: [K] values}: added to the component part of that head's rate_layer weight, so that
    rate = softplus(w_y . y + w_s[k]) lands in a chosen range per component (softplus(x) ~ x beyond 20)."""

    rng = np.random.default_rng(seed)
    cols = make_columns(kind)
    model, om = build_reference(cols, K, D, hidden, 0.0, "supervised", seed)
    model._decoder.eval()
    if rate_boost:
        with torch.no_grad():
            for cname, vals in rate_boost.items():
                wgt = model._decoder.heads[cname].rate_layer.weight
                wgt[0, -K:] += torch.tensor(vals, dtype=wgt.dtype)
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


def _head_noise(rec, pre, cols, tape, i_rn, i_mn, i_ps):
    """Per-column head-sampler draws of ONE decoder forward, in schema order, starting at the given tape positions."""

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
    return i_rn, i_mn, i_ps


def _frame_to_rec(rec, pre, cols, frame):
    for c in cols:
        v = frame[c.name]
        if c.type in ("cat", "ordinal"):
            rec[pre + f"frame/{c.name}"] = np.asarray(v.cat.codes, np.int64)
            rec[pre + f"categories/{c.name}"] = np.asarray([str(x) for x in v.cat.categories])
            rec[pre + f"ordered/{c.name}"] = np.asarray(bool(v.cat.ordered))
        else:
            rec[pre + f"frame/{c.name}"] = v.to_numpy()
        rec[pre + f"dtype/{c.name}"] = np.asarray(str(v.dtype))


def gen_generation(name, kind, K, D, hidden, n, seed, n_train=60):
    """The latent sources of generation and the composed SUAVE.sample (a18 / a19 / a23):
    sampling.sample_prepared_mixture_latents (sampling.py:38-56) for both behaviours,
    SUAVE._draw_conditional_latents (model.py:3042-3114) and SUAVE.sample(conditional=False | True)
    (model.py:4284-4431), each run UNTOUCHED under the noise tape."""

    import warnings

    rng = np.random.default_rng(seed)
    cols = make_columns(kind)
    rec = {"columns": cols_json(cols),
           "meta": json.dumps(dict(kind=kind, K=K, D=D, hidden=list(hidden), n=n, seed=seed, n_train=n_train))}
    cpu = torch.device("cpu")
    for behaviour in ("supervised", "unsupervised"):
        model, om = build_reference(cols, K, D, hidden, 0.0, behaviour, seed)
        model._is_fitted = True
        model._decoder.eval()
        model._select_device = lambda: cpu
        # category labels of two discrete columns (norm stats drive decoder_outputs_to_frame, sampling.py:156-176)
        stats = {c.name: dict(c.stats) for c in cols}
        first_cat = next(c for c in cols if c.type == "cat")
        stats[first_cat.name]["categories"] = [f"lv{i}" for i in range(first_cat.n_classes)]
        model._norm_stats_per_col = stats
        model._feature_layout = {t: {} for t in orc.TYPE_ORDER}
        for c in cols:
            model._feature_layout[c.type][c.name] = 1 if c.type in ("real", "pos", "count") else int(c.n_classes)
        pre = behaviour + "/"
        for k_, v in ref_params(model, behaviour).items():
            rec[pre + "param/" + k_] = v
        rec[pre + "stats"] = json.dumps(stats)
        logits = model._prior_component_logits_tensor().detach().numpy()
        # ---- a18: prior latents
        with NoiseTape() as tape, torch.no_grad():
            tape.multi_mode, tape.prior_logits = "cdf", logits
            z, assign = model._draw_latent_samples(n, conditional=False, targets=None, device=cpu)
        rec[pre + "prior/u"] = tape.prior_u[0].numpy()
        rec[pre + "prior/eps"] = tape.randn[0].numpy()
        rec[pre + "prior/z"] = z.numpy()
        rec[pre + "prior/component"] = assign.argmax(dim=1).numpy().astype(np.int64)
        rec[pre + "prior/assignments"] = assign.numpy()
        # ---- a23: SUAVE.sample(conditional=False), one chunk
        with NoiseTape() as tape, torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tape.multi_mode, tape.prior_logits = "cdf", logits
            frame = model.sample(n, conditional=False, batch_size=n)
        rec[pre + "sample/u"] = tape.prior_u[0].numpy()
        rec[pre + "sample/eps"] = tape.randn[0].numpy()
        _head_noise(rec, pre + "sample/", cols, tape, 1, 0, 0)
        _frame_to_rec(rec, pre + "sample/", cols, frame)
        if behaviour != "supervised":
            continue
        # ---- a19: conditional latents from cached training posteriors
        classes = np.asarray([10, 20, 30])
        train_targets = rng.integers(0, 3, n_train)
        model._classes = classes
        model._class_to_index = {int(c): i for i, c in enumerate(classes)}
        model._train_target_indices = train_targets.astype(np.int64)
        model._train_component_probs = torch.from_numpy(rng.dirichlet(np.ones(K), n_train).astype(np.float32))
        model._train_component_mu = torch.from_numpy(rng.standard_normal((n_train, K, D)).astype(np.float32))
        model._train_component_logvar = torch.from_numpy((0.3 * rng.standard_normal((n_train, K, D))).astype(np.float32))
        y = classes[rng.integers(0, 3, n)]
        rec["cond/classes"] = classes
        rec["cond/train_targets"] = train_targets.astype(np.int64)
        rec["cond/train_probs"] = model._train_component_probs.numpy()
        rec["cond/train_mu"] = model._train_component_mu.numpy()
        rec["cond/train_logvar"] = model._train_component_logvar.numpy()
        rec["cond/y"] = y
        with NoiseTape() as tape, torch.no_grad():
            z, assign = model._draw_conditional_latents(y, cpu)
        rec["cond/u_row"] = torch.cat(tape.randint_u).numpy()
        rec["cond/e_comp"] = torch.cat(tape.multinomial_exp).numpy()          # [n,K]
        rec["cond/eps"] = torch.cat(tape.randn).numpy()                       # [n,D]
        rec["cond/z"] = z.numpy()
        rec["cond/component"] = assign.argmax(dim=1).numpy().astype(np.int64)
        # the training row each draw used (candidate list of the label's class, model.py:3080-3091)
        rows = []
        for lab, u in zip(y, rec["cond/u_row"]):
            cand = np.flatnonzero(train_targets == model._class_to_index[int(lab)])
            rows.append(cand[min(int(np.float32(u) * np.float32(cand.size)), cand.size - 1)])
        rec["cond/rows"] = np.asarray(rows, np.int64)
        # ---- a23: SUAVE.sample(conditional=True, y), one chunk
        with NoiseTape() as tape, torch.no_grad():
            frame = model.sample(n, conditional=True, y=y, batch_size=n)
        rec["cond_sample/u_row"] = torch.cat(tape.randint_u).numpy()
        rec["cond_sample/e_comp"] = torch.cat(tape.multinomial_exp[:n]).numpy()
        rec["cond_sample/eps"] = torch.cat(tape.randn[:n]).numpy()
        _head_noise(rec, "cond_sample/", cols, tape, n, n, 0)
        _frame_to_rec(rec, "cond_sample/", cols, frame)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name)



def _tape_steps(tape, D):
    """(exp, eps) of every _forward_elbo_batch call on the tape, in call order: the Gumbel Exp(1) draws and the
    [B, D] reparameterisation draw (the heads' own discarded sample draws are [B, 1] and are skipped: quirk Q2)."""

    eps = [t for t in tape.randn if t.dim() == 2 and t.shape[1] == D]
    assert len(eps) == len(tape.exp)
    return [(e.numpy(), z.numpy()) for e, z in zip(tape.exp, eps)]


def gen_phases(name, kind, K, D, hidden, n_train, n_val, batch, seed, lr=1e-2, cls_weight=0.7, lam=0.5):
    """The rest of the supervised schedule, each phase run UNTOUCHED under the noise tape (dropout 0):
      * `_run_joint_finetune` (model.py:1866-2086): four learning-rate groups (:1894-1909), validation metrics,
        best-state restore;
      * `_refine_decoder_after_joint` (:2088-2445) in mode decoder_prior -- the L2 prior anchor (:2338-2346) --
        and in mode decoder_only (prior frozen), encoder and classifier frozen (:2251-2281);
      * `_collect_posterior_statistics` + `_closed_form_prior_update_from_stats` (:1669-1741, :2447-2495).
    Recorded: permutations, the draws of every ELBO call (training and validation), every parameter before and
    after each phase, the gradients of every optimisation step, the validation metrics."""

    from suave.modules.heads import ClassificationHead
    import torch.optim

    assert D > 1
    rng = np.random.default_rng(seed)
    cols = make_columns(kind)
    model, om = build_reference(cols, K, D, hidden, 0.0, "supervised", seed)
    model.learning_rate = lr
    model.decoder_refine_prior_lambda = lam
    torch.manual_seed(seed + 3)
    n_classes = 3
    model._classes = np.arange(n_classes)
    model._classifier = ClassificationHead(D, n_classes, class_weight=[1.0, 2.0, 0.5])
    xnum, codes, mask = make_batch(cols, n_train, 0.3, rng)
    vxnum, vcodes, vmask = make_batch(cols, n_val, 0.3, rng)
    x, data_t, mask_t = to_torch_batches(om, xnum, codes, mask)
    vx, vdata_t, vmask_t = to_torch_batches(om, vxnum, vcodes, vmask)
    y = torch.from_numpy(rng.integers(0, n_classes, n_train).astype(np.int64))
    vy = torch.from_numpy(rng.integers(0, n_classes, n_val).astype(np.int64))
    rec = {"columns": cols_json(cols),
           "meta": json.dumps(dict(kind=kind, K=K, D=D, hidden=list(hidden), n_train=n_train, n_val=n_val, batch=batch,
                                   seed=seed, lr=lr, cls_weight=cls_weight, lam=lam, beta=float(model.beta),
                                   n_classes=n_classes, behaviour="supervised", dropout=0.0)),
           "xnum": xnum, "codes": codes, "mask": mask, "val/xnum": vxnum, "val/codes": vcodes, "val/mask": vmask,
           "y": y.numpy(), "val/y": vy.numpy(), "clf/class_weight": model._classifier.class_weight.numpy()}

    def snapshot(pre):
        for k_, v in ref_params(model, "supervised").items():
            rec[pre + "param/" + k_] = v
        for k_, v in model._classifier.state_dict().items():
            if k_ != "class_weight":
                rec[pre + "clf/" + k_] = v.detach().numpy().copy()

    named = ref_named_parameters(model, "supervised") + [("clf." + k_, p_) for k_, p_ in model._classifier.named_parameters()]
    name_of = {id(p_): k_ for k_, p_ in named}
    step_grads = []
    orig_step = torch.optim.Adam.step

    def recording_step(self, *a, **k):
        g = {}
        for group in self.param_groups:
            for p_ in group["params"]:
                if p_.grad is not None:
                    g[name_of[id(p_)]] = p_.grad.detach().numpy().copy()
        step_grads.append(g)
        return orig_step(self, *a, **k)

    # Quirk Q10: `_capture_model_state` snapshots through `tensor.detach().cpu()` (model.py:2680-2694, 4437-4454).  On a
    # CUDA device that copies; on the CPU it returns the live tensor, so in a CPU run the "best state" aliases the
    # parameters and `_restore_model_state` restores nothing.  The fixture records the behaviour of the device the
    # reference is written for (and the build runs on): the snapshot is a copy.
    from collections import OrderedDict

    def state_dict_copy(module):
        if module is None:
            return None
        return OrderedDict((k_, v.detach().clone()) for k_, v in module.state_dict().items())

    SUAVE._state_dict_to_cpu = staticmethod(state_dict_copy)
    SUAVE._tensor_to_cpu = staticmethod(lambda t: None if t is None else t.detach().clone())

    def run_phase(pre, fn):
        snapshot(pre + "before/")
        step_grads.clear()
        torch.optim.Adam.step = recording_step
        try:
            with NoiseTape() as tape:
                out = fn()
        finally:
            torch.optim.Adam.step = orig_step
        snapshot(pre + "after/")
        rec[pre + "n_perms"] = np.asarray(len(tape.perms))
        for i, p_ in enumerate(tape.perms):
            rec[pre + f"perm/{i}"] = p_.numpy()
        steps = _tape_steps(tape, D)
        rec[pre + "n_steps"] = np.asarray(len(steps))
        for i, (e, z) in enumerate(steps):
            rec[pre + f"exp/{i}"] = e
            rec[pre + f"eps/{i}"] = z
        rec[pre + "n_opt_steps"] = np.asarray(len(step_grads))
        for i, g in enumerate(step_grads):
            for k_, v in g.items():
                rec[pre + f"grad/{i}/{k_}"] = v
        return out

    # ---- joint fine-tuning: 2 epochs
    done = run_phase("joint/", lambda: model._run_joint_finetune(
        2, x, data_t, mask_t, vx, vdata_t, vmask_t, batch_size=batch, lr_scale=0.1, early_stop_patience=5,
        y_train_tensor=y, y_val_tensor=vy, classification_loss_weight=cls_weight))
    rec["joint/epochs_done"] = np.asarray(done)
    rec["joint/val_metrics"] = json.dumps({k_: float(v) for k_, v in model._joint_val_metrics.items()})
    model.classification_loss_weight = cls_weight
    # ---- decoder refinement with the L2 prior anchor: 2 epochs
    model.decoder_refine_mode = "decoder_prior"
    done, _ = run_phase("refine_prior/", lambda: model._refine_decoder_after_joint(
        2, x, data_t, mask_t, vx, vdata_t, vmask_t, batch_size=batch, patience=5))
    rec["refine_prior/epochs_done"] = np.asarray(done)
    # ---- decoder only: 1 epoch (prior tensors frozen)
    model.decoder_refine_mode = "decoder_only"
    done, _ = run_phase("refine_decoder/", lambda: model._refine_decoder_after_joint(
        1, x, data_t, mask_t, vx, vdata_t, vmask_t, batch_size=batch, patience=5))
    rec["refine_decoder/epochs_done"] = np.asarray(done)
    # ---- closed-form EM refresh of the prior from posterior responsibilities
    snapshot("em/before/")
    stats = model._collect_posterior_statistics(x, batch_size=batch, temperature=None)
    model._closed_form_prior_update_from_stats(stats, device=torch.device("cpu"))
    snapshot("em/after/")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name)



_JSON_LEAVES = {"schema", "normalization", "feature_layout", "class_to_index", "joint_val_metrics", "auto_configured",
                "heuristic_overrides", "auto_hyperparameters", "temperature_scaler", "temperature_scaler_state",
                "warmup_val_history", "classes", "hidden_dims", "head_hidden_dims"}


def _flatten_payload(obj, prefix, arrays, leaves):
    """A torch archive as data: tensors / arrays -> npz entries, everything else -> one JSON map by path."""

    last = prefix.split("/")[-1]
    if isinstance(obj, dict) and last not in _JSON_LEAVES:
        if not obj:
            leaves[prefix] = {"kind": "dict", "value": {}}
        for k, v in obj.items():
            _flatten_payload(v, f"{prefix}/{k}", arrays, leaves)
    elif isinstance(obj, torch.Tensor):
        arrays[prefix] = obj.detach().numpy()
        leaves[prefix] = {"kind": "tensor"}
    elif isinstance(obj, np.ndarray):
        arrays[prefix] = obj
        leaves[prefix] = {"kind": "ndarray"}
    elif last == "class_to_index" and isinstance(obj, dict):
        leaves[prefix] = {"kind": "pairs", "value": [[k, v] for k, v in obj.items()]}
    else:
        leaves[prefix] = {"kind": type(obj).__name__, "value": obj}


def gen_persistence(name, kind, K, D, hidden, seed, n_train=20):
    """The archive `SUAVE.save` writes (model.py:4456-4579) for a supervised model (classifier, calibration state,
    training posterior caches) and an unsupervised one, stored as data: every tensor + the JSON of every other leaf
    (no pickle travels).  Also proves the other direction in this container: an archive assembled by the build's
    `suave_amd.persist.build_payload` from the same tensors is read back by the reference's own `SUAVE.load`
    (model.py:4581-4950) into identical modules."""

    import tempfile
    from suave.modules.heads import ClassificationHead
    from suave_amd import persist
    from suave_amd.types import Schema as OurSchema

    rng = np.random.default_rng(seed)
    cols = make_columns(kind)
    rec = {"columns": cols_json(cols), "meta": json.dumps(dict(kind=kind, K=K, D=D, hidden=list(hidden), seed=seed))}
    for behaviour in ("supervised", "unsupervised"):
        model, om = build_reference(cols, K, D, hidden, 0.1, behaviour, seed)
        stats = {c.name: dict(c.stats) for c in cols}
        first_cat = next(c for c in cols if c.type == "cat")
        stats[first_cat.name]["categories"] = [f"lv{i}" for i in range(first_cat.n_classes)]
        model._norm_stats_per_col = stats
        model._feature_layout = {t: {} for t in orc.TYPE_ORDER}
        for c in cols:
            model._feature_layout[c.type][c.name] = 1 if c.type in ("real", "pos", "count") else int(c.n_classes)
        model._is_fitted = True
        model._inference_tau = 0.85
        model._warmup_val_history = [{"nll": 12.5, "reconstruction": -11.0, "categorical_kl": 0.1, "gaussian_kl": 0.9}]
        t32 = lambda *shape: torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
        model._train_latent_mu, model._train_latent_logvar = t32(n_train, D), t32(n_train, D)
        model._train_component_logits = t32(n_train, K)
        model._train_component_mu, model._train_component_logvar = t32(n_train, K, D), t32(n_train, K, D)
        model._train_component_probs = torch.softmax(model._train_component_logits, dim=-1)
        if behaviour == "supervised":
            classes = np.asarray([10, 20, 30])
            model._classes = classes
            model.classes_ = classes
            model._class_to_index = {int(c): i for i, c in enumerate(classes)}
            model._train_target_indices = rng.integers(0, 3, n_train).astype(np.int64)
            torch.manual_seed(seed + 5)
            model._classifier = ClassificationHead(D, 3, class_weight=[1.0, 2.0, 0.5], dropout=0.1)
            model._temperature_scaler.temperature, model._temperature_scaler.fitted = 1.7, True
            model._is_calibrated = True
            model.classification_loss_weight = 0.6
            model._joint_val_metrics = {"nll": 12.0, "classification_loss": 0.9, "joint_objective": 12.54}
        pre = behaviour + "/"
        P = ref_params(model, behaviour)
        for k_, v in P.items():
            rec[pre + "param/" + k_] = v
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "ref.pt")
            model.save(path)
            payload = torch.load(path, map_location="cpu", weights_only=False)  # written by this script one line above
            arrays, leaves = {}, {}
            _flatten_payload(payload, pre + "payload", arrays, leaves)
            rec.update(arrays)
            rec[pre + "leaves"] = json.dumps(leaves)
            rec[pre + "tree"] = json.dumps(persist.tree(payload))
            # ---- the other direction: the build's payload through the reference's loader
            cache = {"mean": model._train_latent_mu, "post_logvar": model._train_latent_logvar,
                     "logits": model._train_component_logits, "mu": model._train_component_mu,
                     "logvar": model._train_component_logvar, "probs": model._train_component_probs}
            clf = model._classifier
            ours = persist.build_payload(
                metadata=dict(payload["metadata"]), state=P, schema=OurSchema(payload["metadata"]["schema"]),
                classifier_state=None if clf is None else {k_: v for k_, v in clf.state_dict().items()},
                class_weight=None if clf is None else clf.class_weight, use_weight=clf is not None,
                temperature_scaler=model._temperature_scaler.state_dict(), norm_stats=stats,
                classes=model._classes, class_to_index=model._class_to_index, is_calibrated=model._is_calibrated,
                train_cache=cache, train_target_indices=model._train_target_indices,
                warmup_val_history=model._warmup_val_history, joint_val_metrics=model._joint_val_metrics)
            path2 = os.path.join(tmp, "ours.pt")
            torch.save(ours, path2)
            torch.load(path2, map_location="cpu", weights_only=True)  # nothing but tensors and plain containers
            back = SUAVE.load(path2)
        for k_, v in ref_params(back, behaviour).items():
            assert np.array_equal(v, P[k_]), k_
        assert back._norm_stats_per_col == stats and back._feature_layout == model._feature_layout
        assert back._inference_tau == 0.85 and back.hidden_dims == tuple(hidden) and back.behaviour == behaviour
        assert torch.equal(back._train_component_mu, model._train_component_mu)
        if behaviour == "supervised":
            assert list(back._classes) == [10, 20, 30] and back._class_to_index == model._class_to_index
            assert np.array_equal(back._train_target_indices, model._train_target_indices)
            for k_, v in model._classifier.state_dict().items():
                assert torch.equal(back._classifier.state_dict()[k_], v), k_
            assert back._is_calibrated and abs(back._temperature_scaler.temperature - 1.7) < 1e-12
        ours_tree, ref_tree = persist.tree(ours), persist.tree(payload)
        assert set(ours_tree) == set(ref_tree), set(ours_tree) ^ set(ref_tree)
        rec[pre + "ours_loaded_by_reference"] = np.asarray(True)
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
        # count rates around the branch point 64, in the mode-walk range (64 .. 65536) and beyond it
        (gen_decode, ("decode_bigcount6", "bigcount6", 3, 6, (16, 8), 96, 64),
         {"rate_boost": {"n0": [60.0, 64.0, 70.0], "n1": [100.0, 400.0, 1000.0], "n2": [5000.0, 30000.0, 65530.0],
                         "n3": [66000.0, 200000.0, 3.0]}}),
        (gen_joint, ("joint_sup_mixed7", "mixed7", 3, 6, (16, 8), 48, 0.3, 71), {}),
        (gen_predictive, ("predictive_mixed7", "mixed7", 3, 6, (16, 8), 24, 81), {}),
        (gen_impute, ("impute_mixed8", 91), {}),
        (gen_persistence, ("persistence_mixed7", "mixed7", 3, 6, (16, 8), 121), {}),
        (gen_phases, ("phases_mixed7", "mixed7", 3, 6, (16, 8), 40, 16, 24, 111), {}),
        (gen_generation, ("generation_mixed7", "mixed7", 4, 6, (16, 8), 96, 101), {}),
    ]
    for fn, args, kw in jobs:
        if want(args[0]):
            fn(*args, **kw)


if __name__ == "__main__":
    main()
