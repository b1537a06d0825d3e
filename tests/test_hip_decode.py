"""GPU parity of the generative decode (C-ABI) against the reference's golden vectors and the CPU oracle.

Sampled categorical / ordinal / count values and mixture components must be bit-exact on identical
inputs and injected noise; real / positive samples within 1e-4 (they are bit-exact too, given y_full).
"""

import numpy as np
import pytest

from oracle import sampling_oracle as sorc
from oracle import suave_oracle as orc
from tests.helpers import fixture_model, layout_from_columns, load_golden, rel_err, s64_columns

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _engine(om, P):
    from suave_amd.engine import Engine

    eng = Engine(layout_from_columns(om.columns, K=om.K, D=om.D, hidden=om.hidden, dropout=om.dropout,
                                     behaviour=om.behaviour), "cuda:0")
    eng.load_state(P)
    return eng


def _pack_noise(om, layout, per_col, B):
    nn = np.zeros((B, len(om.columns)), np.float32)
    nu = np.zeros((B, len(om.columns)), np.float32)
    ne = np.ones((B, max(layout.total_classes, 1)), np.float32)
    for j, c in enumerate(om.columns):
        nz = per_col[c.name]
        if c.type in ("real", "pos", "count"):
            nn[:, j] = np.asarray(nz["normal"]).reshape(-1)
        if c.type == "count":
            nu[:, j] = np.asarray(nz["uniform"]).reshape(-1)
        if c.type in ("cat", "ordinal"):
            off = layout.class_offsets[c.name]
            ne[:, off:off + c.n_classes] = nz["exp"]
    return nn, nu, ne


def _unpack(om, layout, onum, ocodes):
    out = {}
    i_num = i_disc = 0
    for c in om.columns:
        if c.type in ("real", "pos", "count"):
            out[c.name] = onum[:, i_num]
            i_num += 1
        else:
            out[c.name] = ocodes[:, i_disc].astype(np.int64)
            i_disc += 1
    return out


@pytest.mark.parametrize("name", ["decode_mixed12", "decode_ydim", "decode_wide7", "decode_bigcount6"])
def test_golden_decode(name):
    import torch

    fx = load_golden(name)
    om, P, meta = fixture_model(fx)
    eng = _engine(om, P)
    B = meta["B"]
    per_col = {c.name: {k.split("/")[-1]: v for k, v in fx.items() if k.startswith(f"noise/{c.name}/")} for c in om.columns}
    nn, nu, ne = _pack_noise(om, eng.layout, per_col, B)
    dev = eng.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    onum, ocodes = eng.decode_sample(t(fx["z"]), component=t(fx["k"].astype(np.int32)), noise_normal=t(nn),
                                     noise_uniform=t(nu), noise_exp=t(ne))
    got = _unpack(om, eng.layout, onum.cpu().numpy(), ocodes.cpu().numpy())
    for c in om.columns:
        ref = fx[f"out/{c.name}/frame"]
        if c.type in ("real", "pos"):
            assert rel_err(got[c.name], ref) < TOL, c.name
        elif c.type == "count":
            # a count is a step function of its rate: at rates >= 2048 the trunk's ~1e-6 relative difference from
            # torch's y moves the steps (90 % of the rows must still agree; test_poisson_map_on_chosen_rates pins the map)
            same = np.rint(got[c.name]).astype(np.int64) == ref.astype(np.int64)
            big = fx[f"out/{c.name}/params/rate"].reshape(-1) >= 2048
            assert same[~big].all() and (not big.any() or same[big].mean() >= 0.9), c.name
        else:  # reference's sampled values, bit for bit
            assert np.array_equal(np.rint(got[c.name]).astype(np.int64), ref.astype(np.int64)), c.name


def _rate_model(rates):
    """One count column whose rate depends on the component alone: rate_k = softplus(w_s[k]) + eps, zero y weights."""

    cols = [orc.Column("n0", "count", stats={"offset": 1.0}), orc.Column("r0", "real", stats={"mean": 0.0, "std": 1.0})]
    K = len(rates)
    om = orc.OracleModel(cols, n_components=K, latent_dim=32, hidden=(256, 128))
    P = om.init_params(seed=7)
    w = P["decoder.heads.n0.rate_layer.weight"]
    w[:] = 0.0
    w[0, -K:] = np.asarray(rates, np.float32)
    return cols, om, P


@pytest.mark.parametrize("fused", [True, False])
def test_poisson_map_on_chosen_rates(fused):
    """Every range of the (rate, u, n) -> count map (sample_col.h poisson_from_noise) against the oracle's, bit for bit,
    on rates that do not depend on the trunk: below 64, around the branch points, the mode walk up to 65 535, beyond."""

    import torch

    rates = [0.5, 20.0, 63.4, 63.6, 64.0, 64.5, 100.0, 100.7, 1000.0, 4097.25, 30000.0, 65535.5, 65536.0, 70000.0, 1.0e6, 3.0e6]
    cols, om, P = _rate_model(rates)
    K = len(rates)
    n = 1 << 16
    rng = np.random.default_rng(3)
    comp = rng.integers(0, K, n).astype(np.int32)
    u = rng.random(n).astype(np.float32)
    u[:64] = np.float32(1.0) - np.float32(2.0) ** -24  # the far tail: the walk runs until both tails are spent
    u[64:128] = np.float32(2.0) ** -24
    nz = {"n0": {"normal": rng.standard_normal(n).astype(np.float32), "uniform": u},
          "r0": {"normal": rng.standard_normal(n).astype(np.float32)}}
    eng = _engine(om, P)
    nn, nu, ne = _pack_noise(om, eng.layout, nz, n)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    # component indices take decode.hip's fused kernel, dense (here one-hot) assignments the per-layer path + sample.hip
    how = {"component": t(comp)} if fused else {"assignments": t(np.eye(K, dtype=np.float32)[comp])}
    onum, _ = eng.decode_sample(t(rng.standard_normal((n, 32)).astype(np.float32)), noise_normal=t(nn),
                                noise_uniform=t(nu), noise_exp=t(ne), **how)
    got = _unpack(om, eng.layout, onum.cpu().numpy(), np.zeros((n, 0), np.int16))["n0"]
    rate = sorc.det_softplus(np.asarray(rates, np.float32)[comp]) + sorc.EPS
    ref = sorc.poisson_from_noise(rate.astype(np.float32), u, nz["n0"]["normal"])
    assert np.array_equal(got.astype(np.int64), np.maximum(np.rint(ref), 0).astype(np.int64))


def test_poisson_draws_of_the_kernel_generator_are_poisson():
    """In-kernel draws (Philox -> u01 -> the map) at rate 100 and 1 000: first three moments and the 97.5 % quantile
    against scipy.stats.poisson, 2^20 rows per rate."""

    import torch
    from scipy import stats

    rates = [100.0, 1000.0, 20.0, 70000.0]
    cols, om, P = _rate_model(rates)
    eng = _engine(om, P)
    n = 1 << 20
    for k, r in enumerate(rates):
        comp = torch.full((n,), k, dtype=torch.int32, device=eng.device)
        onum, _ = eng.decode_sample(torch.zeros((n, 32), device=eng.device), component=comp, seed=11, row_offset=k * n)
        c = _unpack(om, eng.layout, onum.cpu().numpy(), np.zeros((n, 0), np.int16))["n0"].astype(np.float64)
        m, v = c.mean(), c.var()
        sk = ((c - m) ** 3).mean() / v ** 1.5
        assert abs(m - r) < 4.5 * np.sqrt(r / n), (r, m)
        assert abs(v - r) < 4.5 * r * np.sqrt(2.0 / n) + (0.1 if r > 65536 else 0.0), (r, v)
        assert abs(sk - r ** -0.5) < 4.5 * np.sqrt(6.0 / n), (r, sk)
        q = stats.poisson.ppf(0.975, r)
        want = stats.poisson.cdf(q, r)
        assert abs((c <= q).mean() - want) < 4.5 * np.sqrt(want * (1 - want) / n) + (2e-4 if r > 65536 else 0.0), r


@pytest.mark.parametrize("dense_s", [False, True])
def test_s64_decode_bit_exact_given_y(dense_s):
    _decode_given_y(s64_columns(64, real_variant="scaled"), (256, 128), 5000, dense_s)


@pytest.mark.parametrize("dense_s", [False, True])
def test_w1024_miniature_decode_bit_exact_given_y(dense_s):
    """BASELINE config #4 in miniature: the y row no longer fits LDS, so the unstaged sample kernel runs, and the
    wide categorical columns read their padded weight layout."""

    from tests.helpers import w1024_columns

    _decode_given_y(w1024_columns(25, 12, 6, 14, 7), (48, 32), 700, dense_s)


def _decode_given_y(cols, hidden, n, dense_s, ties=False):
    import torch

    om = orc.OracleModel(cols, n_components=10, latent_dim=32, hidden=hidden)
    P = om.init_params(seed=7)
    rng = np.random.default_rng(8)
    for k in P:
        if ".heads." in k:
            P[k] = (0.5 * rng.standard_normal(P[k].shape)).astype(np.float32)
        if ties and k.endswith("logits_layer.weight") and P[k].shape[0] >= 2:
            P[k][1] = P[k][0]  # classes 1 and 2 get the same logit in every row
    eng = _engine(om, P)
    z = rng.standard_normal((n, 32)).astype(np.float32)
    comp = rng.integers(0, 10, n).astype(np.int32)
    if dense_s:
        s = rng.dirichlet(np.ones(10), n).astype(np.float32)
    else:
        s = np.eye(10, dtype=np.float32)[comp]
    per_col = {}
    for c in cols:
        if c.type in ("real", "pos"):
            per_col[c.name] = {"normal": rng.standard_normal(n).astype(np.float32)}
        elif c.type == "count":
            per_col[c.name] = {"normal": rng.standard_normal(n).astype(np.float32), "uniform": rng.random(n).astype(np.float32)}
        else:
            e = rng.exponential(size=(n, c.n_classes)).astype(np.float32)
            if ties and c.type == "cat" and c.n_classes >= 3:
                # ... and the same, or a one-ulp different, exponential draw: exact ties and races decided in the last
                # bit, where only the reference's own arithmetic gives the reference's winner
                e[:, 1] = np.minimum(e[:, 1], 0.01)  # small draw = large Gumbel: the pair often leads the race (also among 256)
                ulp = rng.integers(-1, 2, n)
                e[:, 2] = np.where(ulp == 0, e[:, 1], np.nextafter(e[:, 1], np.where(ulp > 0, np.float32(np.inf), np.float32(0))))
            per_col[c.name] = {"exp": e}
    nn, nu, ne = _pack_noise(om, eng.layout, per_col, n)
    dev = eng.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    onum, ocodes, y = eng.decode_sample(t(z), component=None if dense_s else t(comp), assignments=t(s) if dense_s else None,
                                        noise_normal=t(nn), noise_uniform=t(nu), noise_exp=t(ne), want_y=True)
    y = y.cpu().numpy()
    y_ref, _ = orc.decoder_trunk_forward(om, P, z)
    assert rel_err(y, y_ref) < TOL
    ref = sorc.sample_heads(om, P, y, s, per_col)  # the oracle's maps on the GPU's own y_full
    got = _unpack(om, eng.layout, onum.cpu().numpy(), ocodes.cpu().numpy())
    for c in cols:
        if c.type in ("real", "pos"):  # values: hardware exp / log / sqrt against the oracle's (tolerance 2e-5, owed 1e-4)
            assert rel_err(got[c.name], ref[c.name]) < 2e-5, c.name
        else:  # decisions: bit for bit
            assert np.array_equal(np.rint(got[c.name]).astype(np.int64), ref[c.name]), c.name
    if ties:
        return got


def test_categorical_races_decided_in_the_last_bit_follow_the_reference_arithmetic():
    """The sample kernel runs a race with the hardware transcendentals first and repeats it with detmath only when
    the winner's lead is inside the error bound.  Here classes 1 and 2 of every categorical column have identical
    logits and exponential draws that are equal or one ulp apart, and mostly lead the race: every winner must still
    be the oracle's, and both classes of the pair must actually win somewhere."""

    cols = s64_columns(64, real_variant="scaled")
    got = _decode_given_y(cols, (256, 128), 4000, False, ties=True)
    wins = np.concatenate([got[c.name] for c in cols if c.type == "cat" and c.n_classes >= 3])
    # (class 2 only wins when its draw is the one-ulp smaller one AND the rounded logarithms differ: a fraction of a percent)
    assert (wins == 1).mean() > 0.2 and (wins == 2).sum() >= 1
    # the wide categorical columns race on MFMA logits first (k_wide_sample): same requirement
    from tests.helpers import w1024_columns

    cols = w1024_columns(25, 12, 6, 14, 7)
    got = _decode_given_y(cols, (48, 32), 3000, False, ties=True)
    wins = np.concatenate([got[c.name] for c in cols if c.type == "cat" and c.n_classes >= 32])
    assert (wins == 1).mean() > 0.02  # the tied pair leads a fair share of the races among up to 256 classes


def test_prior_and_conditional_latents():
    import torch

    cols = s64_columns(10)
    K, D = 5, 6
    om = orc.OracleModel(cols, n_components=K, latent_dim=D, hidden=(16,))
    P = om.init_params(seed=2)
    eng = _engine(om, P)
    dev = eng.device
    rng = np.random.default_rng(3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    n = 3000
    u = rng.random(n).astype(np.float32)
    eps = rng.standard_normal((n, D)).astype(np.float32)
    cdf = sorc.prior_cdf(P["prior.logits"])
    z, comp = eng.prior_latents(n, t(cdf), u=t(u), eps=t(eps))
    z_ref, k_ref = sorc.prior_latents(P["prior.logits"], P["prior.mu"], P["prior.logvar"], u, eps)
    assert np.array_equal(comp.cpu().numpy().astype(np.int64), k_ref)
    assert rel_err(z.cpu().numpy(), z_ref) < 1e-5
    # conditional: cached training posteriors of N rows, labels over 3 classes
    N = 400
    train_probs = rng.dirichlet(np.ones(K), N).astype(np.float32)
    train_mu = rng.standard_normal((N, K, D)).astype(np.float32)
    train_lv = (0.3 * rng.standard_normal((N, K, D))).astype(np.float32)
    targets = rng.integers(0, 3, N)
    class_rows = np.concatenate([np.flatnonzero(targets == c) for c in range(3)]).astype(np.int64)
    class_off = np.concatenate([[0], np.cumsum([(targets == c).sum() for c in range(3)])]).astype(np.int64)
    labels = rng.integers(0, 3, n).astype(np.int32)
    u_row = rng.random(n).astype(np.float32)
    e_comp = rng.exponential(size=(n, K)).astype(np.float32)
    z, comp = eng.conditional_latents(t(labels), t(class_off), t(class_rows), t(train_probs), t(train_mu), t(train_lv),
                                      u_row=t(u_row), e_comp=t(e_comp), eps=t(eps))
    z_ref, k_ref, _ = sorc.conditional_latents(class_off, class_rows, labels, train_probs, train_mu, train_lv, u_row,
                                               e_comp, eps)
    assert np.array_equal(comp.cpu().numpy().astype(np.int64), k_ref)
    assert rel_err(z.cpu().numpy(), z_ref) < 1e-5
    # generated noise: every row gets a component in range and finite latents; same seed -> same draws
    z1, c1 = eng.prior_latents(n, t(cdf), seed=5, row_offset=100)
    z2, c2 = eng.prior_latents(n, t(cdf), seed=5, row_offset=100)
    assert torch.equal(z1, z2) and torch.equal(c1, c2)
    assert int(c1.min()) >= 0 and int(c1.max()) < K and bool(torch.isfinite(z1).all())
    # row-sharded generation is independent of the shard boundaries (counter = global row index)
    za, ca = eng.prior_latents(1000, t(cdf), seed=5, row_offset=100)
    zb, cb = eng.prior_latents(2000, t(cdf), seed=5, row_offset=1100)
    assert torch.equal(torch.cat([za, zb]), z1) and torch.equal(torch.cat([ca, cb]), c1)


def test_encode_matches_reference():
    import torch
    from suave_amd.engine import PackedTable

    fx = load_golden("posterior_mixed7")
    om, P, meta = fixture_model(fx)
    eng = _engine(om, P)
    table = PackedTable(fx["xnum"], fx["codes"], fx["mask"], eng.device)
    B, K, D = meta["B"], om.K, om.D
    out = eng.encode(table, 0, B)
    enc = out["enc_out"].cpu().numpy()
    assert rel_err(enc[:, :K], fx["out/logits"]) < TOL
    assert rel_err(enc[:, K:K + K * D], fx["out/mu"].reshape(B, -1)) < TOL
    assert rel_err(enc[:, K + K * D:], fx["out/logvar"].reshape(B, -1)) < TOL
    assert rel_err(out["mean"].cpu().numpy(), fx["out/post_mean"]) < TOL
    assert rel_err(out["logvar"].cpu().numpy(), fx["out/post_logvar"]) < TOL
    assert rel_err(out["probs"].cpu().numpy(), fx["out/post_probs"]) < TOL
    out = eng.encode(table, 0, B, temperature=0.5)
    assert rel_err(out["mean"].cpu().numpy(), fx["out/post_mean_t05"]) < TOL
    assert rel_err(out["logvar"].cpu().numpy(), fx["out/post_logvar_t05"]) < TOL
    assert rel_err(out["probs"].cpu().numpy(), fx["out/post_probs_t05"]) < TOL


def test_golden_posterior_predictive():
    """suave_posterior_draws + suave_decode_params + suave_decode_sample against the reference's
    _posterior_predictive_params / _draw_predictive_samples (model.py:3304-3421), every attribute, L draws."""

    import torch
    from suave_amd.engine import PackedTable

    fx = load_golden("predictive_mixed7")
    om, P, meta = fixture_model(fx)
    eng = _engine(om, P)
    L, B, K, D = meta["L"], meta["B"], om.K, om.D
    dev = eng.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    table = PackedTable(fx["xnum"], fx["codes"], fx["mask"], dev)
    enc = eng.encode(table, 0, B)["enc_out"]
    lg, mu, lv, _ = orc.encoder_forward(om, P, orc.dense_from_packed(om, fx["xnum"], fx["codes"], fx["mask"])[0])
    names = [c.name for c in om.columns]
    num_names = [c.name for c in om.columns if c.type in ("real", "pos", "count")]
    for j, c in enumerate(om.columns):
        e, eps = fx[f"noise/{c.name}/comp_exp"], fx[f"noise/{c.name}/eps"]
        z, comp = eng.posterior_draws(enc, L, e=t(e), eps=t(eps))
        k_ref, z_ref = sorc.posterior_draws(lg, mu, lv, e, eps)
        assert np.array_equal(comp.cpu().numpy().astype(np.int64), k_ref.reshape(-1)), c.name  # bit-exact components
        assert rel_err(z.cpu().numpy(), z_ref.reshape(L * B, D)) < TOL
        prm = eng.decode_params(z, j, component=comp).cpu().numpy()
        ref = {kk.split("/")[-1]: v for kk, v in fx.items() if kk.startswith(f"out/{c.name}/params/")}
        order = {"real": ["mean", "var"], "pos": ["mean_log", "var_log", "mean"], "count": ["rate"],
                 "cat": ["probs"], "ordinal": ["probs"]}[c.type]
        if c.type in ("cat", "ordinal"):
            assert rel_err(prm, ref["probs"]) < TOL, c.name
        else:
            for i, kk in enumerate(order):
                assert rel_err(prm[:, i], ref[kk].reshape(-1)) < TOL, (c.name, kk)
        if c.type in ("real", "pos", "count"):
            nn = np.zeros((L * B, len(names)), np.float32)
            nu = np.full((L * B, len(names)), 0.5, np.float32)
            nn[:, j] = fx[f"noise/{c.name}/normal"].reshape(-1)
            if c.type == "count":
                nu[:, j] = fx[f"noise/{c.name}/uniform"].reshape(-1)
            onum, _ = eng.decode_sample(z, component=comp, noise_normal=t(nn), noise_uniform=t(nu))
            got = onum.cpu().numpy()[:, num_names.index(c.name)].reshape(L, B)
            if c.type == "count":
                assert np.array_equal(np.rint(got), np.rint(fx[f"out/{c.name}/samples"])), c.name
            else:
                assert rel_err(got, fx[f"out/{c.name}/samples"]) < TOL, c.name


@pytest.mark.parametrize("behaviour", ["supervised", "unsupervised"])
@pytest.mark.parametrize("strategy", ["hard", "soft", "sample"])
def test_golden_impute_end_to_end(behaviour, strategy):
    """SUAVE.impute on a raw frame with NaNs against the reference's own impute (model.py:3512-3692): the
    (latent, assignments) handed to the decoder (tests/test_impute.py:107-146), the reconstruction under
    injected draws (discrete cells bit-exact) and the preserved observed cells."""

    import json

    import pandas as pd
    import torch
    from suave_amd.engine import Engine
    from suave_amd.model import SUAVE
    from suave_amd.plan import ModelLayout
    from suave_amd.types import Schema
    from tests.helpers import columns_from_json

    fx = load_golden("impute_mixed8")
    cols = columns_from_json(fx["columns"])
    schema = Schema(json.loads(str(fx["schema"])))
    stats = json.loads(str(fx["stats"]))
    P = {k[len(behaviour) + 7:]: v for k, v in fx.items() if k.startswith(behaviour + "/param/")}
    K, D, hidden = 3, 5, (16, 8)
    m = SUAVE(schema, behaviour=behaviour, latent_dim=D, n_components=K, hidden_dims=hidden, dropout=0.0,
              gumbel_temperature=0.9)
    m._norm_stats_per_col = stats
    m._layout = ModelLayout(schema, K, D, hidden, 0.0, behaviour, stats)
    m._engine = Engine(m._layout, "cuda:0")
    m._engine.load_state(P)
    m._inference_tau = 0.7
    m._is_fitted = True
    X = pd.DataFrame({k[4:]: v for k, v in fx.items() if k.startswith("raw/")})
    n = len(X)
    pre = f"{behaviour}/{strategy}/"
    om = orc.OracleModel(cols, n_components=K, latent_dim=D, hidden=hidden, behaviour=behaviour)
    per_col = {c.name: {k.split("/")[-1]: v for k, v in fx.items() if k.startswith(pre + f"noise/{c.name}/")} for c in cols}
    nn, nu, ne = _pack_noise(om, m._layout, per_col, n)
    dev = m._engine.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    noise = {"normal": t(nn), "uniform": t(nu), "exp": t(ne)}
    if strategy == "sample":
        noise["assign_exp"] = t(fx[pre + "noise/assign_exp"])
    routing = {}
    recon = m.impute(X, only_missing=False, assignment_strategy=strategy, _noise=noise, _routing=routing)
    assert rel_err(routing["latent"].cpu().numpy(), fx[pre + "latent"]) < TOL
    assert rel_err(routing["assignments"].cpu().numpy(), fx[pre + "assignments"]) < TOL
    for c in cols:
        ref = fx[pre + f"recon/{c.name}"]
        if c.type in ("real", "pos"):
            assert rel_err(recon[c.name].to_numpy(), ref) < TOL, c.name
        elif c.type == "count":
            assert np.array_equal(recon[c.name].to_numpy().astype(np.int64), ref.astype(np.int64)), c.name
        else:
            assert np.array_equal(np.asarray(recon[c.name].cat.codes, np.int64), ref), c.name
    merged = m.impute(X, only_missing=True, assignment_strategy=strategy, _noise=noise)
    assert not merged.isna().any().any()
    for c in cols:
        obs = ~X[c.name].isna().to_numpy()
        v = merged[c.name]
        got = np.asarray(v.cat.codes if c.type in ("cat", "ordinal") else v, np.float64)[obs]
        if strategy == "hard":
            assert np.allclose(got, fx[f"{behaviour}/merged_observed/{c.name}"], rtol=1e-6), c.name
        assert np.allclose(got, X[c.name].to_numpy()[obs], rtol=1e-6), c.name


def test_encode_with_embedded_wide_columns_matches_oracle():
    """suave_encode on a schema with wide categorical columns: the first encoder layer gathers embedding rows
    (embed.hip) instead of multiplying one-hot blocks; rows include missing cells (code -1) and a gathered,
    ragged batch."""

    import torch
    from suave_amd.engine import PackedTable
    from tests.helpers import synth_packed, w1024_columns

    cols = w1024_columns(25, 12, 6, 14, 7)
    om = orc.OracleModel(cols, n_components=4, latent_dim=8, hidden=(48, 32))
    P = om.init_params(seed=3)
    eng = _engine(om, P)
    assert any(c.enc_embed >= 0 for c in eng.layout.cols)
    n = 777
    xnum, codes, mask = synth_packed(cols, n, 0.2, seed=9)
    table = PackedTable(xnum, codes, mask, eng.device)
    idx = np.random.default_rng(1).permutation(n)[:515].astype(np.int32)
    out = eng.encode(table, 0, len(idx), row_index=torch.from_numpy(idx).to(eng.device))
    x_enc, _, _ = orc.dense_from_packed(om, xnum[idx], codes[idx], mask[idx])
    lg, mu, lv, _ = orc.encoder_forward(om, P, x_enc)
    enc = out["enc_out"].cpu().numpy()
    K, D = om.K, om.D
    assert rel_err(enc[:, :K], lg) < TOL
    assert rel_err(enc[:, K:K + K * D], mu.reshape(len(idx), -1)) < TOL
    assert rel_err(enc[:, K + K * D:], lv.reshape(len(idx), -1)) < TOL


def _generation_model(behaviour):
    """A fitted-by-hand suave_amd.SUAVE on the generation fixture's parameters."""

    import json

    from suave_amd.engine import Engine
    from suave_amd.model import SUAVE
    from suave_amd.plan import ModelLayout
    from suave_amd.types import Schema
    from tests.helpers import columns_from_json

    fx = load_golden("generation_mixed7")
    meta = json.loads(str(fx["meta"]))
    stats = json.loads(str(fx[behaviour + "/stats"]))
    cols = columns_from_json(fx["columns"])
    schema = Schema({c.name: {"type": c.type, **({"n_classes": c.n_classes} if c.n_classes else {})} for c in cols})
    K, D, hidden = meta["K"], meta["D"], tuple(meta["hidden"])
    P = {k[len(behaviour) + 7:]: v for k, v in fx.items() if k.startswith(behaviour + "/param/")}
    m = SUAVE(schema, behaviour=behaviour, latent_dim=D, n_components=K, hidden_dims=hidden, dropout=0.0)
    m._norm_stats_per_col = stats
    m._layout = ModelLayout(schema, K, D, hidden, 0.0, behaviour, stats)
    m._engine = Engine(m._layout, "cuda:0")
    m._engine.load_state(P)
    m._is_fitted = True
    om = orc.OracleModel(cols, n_components=K, latent_dim=D, hidden=hidden, behaviour=behaviour)
    return fx, meta, cols, om, m


def _check_sample_frame(frame, cols, fx, pre):
    for c in cols:
        ref = fx[pre + f"frame/{c.name}"]
        v = frame[c.name]
        assert str(v.dtype) == str(fx[pre + f"dtype/{c.name}"]), c.name  # float32 / int64 / category
        if c.type in ("real", "pos"):
            assert rel_err(v.to_numpy(), ref) < TOL, c.name
        elif c.type == "count":
            assert np.array_equal(v.to_numpy(), ref), c.name
        else:
            assert np.array_equal(np.asarray(v.cat.codes, np.int64), ref), c.name
            assert [str(x) for x in v.cat.categories] == list(fx[pre + f"categories/{c.name}"]), c.name
            assert bool(v.cat.ordered) == bool(fx[pre + f"ordered/{c.name}"]), c.name


@pytest.mark.parametrize("behaviour", ["supervised", "unsupervised"])
def test_golden_prior_latents_and_sample(behaviour):
    """suave_prior_latents and SUAVE.sample(conditional=False) against the reference's
    sampling.sample_prepared_mixture_latents (sampling.py:38-56) and SUAVE.sample (model.py:4284-4431) on the
    recorded draws: components bit-exact, z to 1e-4, the frame's discrete values / dtypes / categories identical."""

    import torch

    fx, meta, cols, om, m = _generation_model(behaviour)
    eng, n = m._engine, meta["n"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    pre = behaviour + "/prior/"
    z, comp = eng.prior_latents(n, m._prior_cdf(), u=t(fx[pre + "u"]), eps=t(fx[pre + "eps"]))
    assert np.array_equal(comp.cpu().numpy().astype(np.int64), fx[pre + "component"])
    assert rel_err(z.cpu().numpy(), fx[pre + "z"]) < TOL
    pre = behaviour + "/sample/"
    per_col = {c.name: {k.split("/")[-1]: v for k, v in fx.items() if k.startswith(pre + f"noise/{c.name}/")} for c in cols}
    nn, nu, ne = _pack_noise(om, m._layout, per_col, n)
    noise = {"u": t(fx[pre + "u"]), "eps": t(fx[pre + "eps"]), "normal": t(nn), "uniform": t(nu), "exp": t(ne)}
    _check_sample_frame(m.sample(n, _noise=noise), cols, fx, pre)
    _check_sample_frame(m.sample(n, batch_size=17, _noise=noise), cols, fx, pre)  # chunking does not change rows


def test_golden_conditional_latents_and_sample():
    """suave_conditional_latents and SUAVE.sample(conditional=True, y) against SUAVE._draw_conditional_latents
    (model.py:3042-3114): chosen component bit-exact (it is drawn from the chosen training row's cached
    probabilities, so a wrong row shows), z to 1e-4, frame identical for discrete columns."""

    import torch

    fx, meta, cols, om, m = _generation_model("supervised")
    eng, n = m._engine, meta["n"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    classes = fx["cond/classes"]
    m._classes = classes
    m._class_to_index = {int(c): i for i, c in enumerate(classes)}
    m._train_target_indices = fx["cond/train_targets"]
    m._train_cache = {"probs": t(fx["cond/train_probs"]), "mu": t(fx["cond/train_mu"]), "logvar": t(fx["cond/train_logvar"])}
    tt = fx["cond/train_targets"]
    class_rows = np.concatenate([np.flatnonzero(tt == c) for c in range(len(classes))]).astype(np.int64)
    class_off = np.concatenate([[0], np.cumsum([(tt == c).sum() for c in range(len(classes))])]).astype(np.int64)
    labels = np.searchsorted(classes, fx["cond/y"]).astype(np.int32)
    z, comp = eng.conditional_latents(t(labels), t(class_off), t(class_rows), m._train_cache["probs"], m._train_cache["mu"],
                                      m._train_cache["logvar"], u_row=t(fx["cond/u_row"]), e_comp=t(fx["cond/e_comp"]),
                                      eps=t(fx["cond/eps"]))
    assert np.array_equal(comp.cpu().numpy().astype(np.int64), fx["cond/component"])
    assert rel_err(z.cpu().numpy(), fx["cond/z"]) < TOL
    # z = mu[row, k] + eps * exp(0.5 logvar[row, k]) with the reference's rows: pins the chosen training rows too
    rows, k = fx["cond/rows"], fx["cond/component"]
    z_rows = fx["cond/train_mu"][rows, k] + fx["cond/eps"] * np.exp(0.5 * fx["cond/train_logvar"][rows, k])
    assert rel_err(z.cpu().numpy(), z_rows) < 1e-5
    pre = "cond_sample/"
    per_col = {c.name: {kk.split("/")[-1]: v for kk, v in fx.items() if kk.startswith(pre + f"noise/{c.name}/")} for c in cols}
    nn, nu, ne = _pack_noise(om, m._layout, per_col, n)
    noise = {"u_row": t(fx[pre + "u_row"]), "e_comp": t(fx[pre + "e_comp"]), "eps": t(fx[pre + "eps"]),
             "normal": t(nn), "uniform": t(nu), "exp": t(ne)}
    _check_sample_frame(m.sample(n, conditional=True, y=fx["cond/y"], _noise=noise), cols, fx, pre)
    _check_sample_frame(m.sample(n, conditional=True, y=fx["cond/y"], batch_size=40, _noise=noise), cols, fx, pre)


@pytest.mark.parametrize("name", ["decode_mixed12", "decode_ydim", "decode_wide7"])
def test_golden_per_column_log_px(name):
    """suave_decode_log_px against the reference decoder's per-head "log_px" / "log_px_missing" (decoder.py:83-94,
    128-142, 169-176, 211-219, 255-262) at the fixture's latents and components, every column, tolerance 1e-4; and
    the same per-column terms out of the ELBO step (suave_elbo_out_t) add up to its reconstruction."""

    import torch
    from suave_amd.engine import PackedTable

    fx = load_golden(name)
    om, P, meta = fixture_model(fx)
    eng = _engine(om, P)
    B = meta["B"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    table = PackedTable(fx["xnum"], fx["codes"], fx["mask"], eng.device)
    lp, lpm = eng.decode_log_px(t(fx["z"]), t(fx["k"].astype(np.int32)), table)
    lp, lpm = lp.cpu().numpy(), lpm.cpu().numpy()
    for j, c in enumerate(om.columns):
        assert rel_err(lp[:, j], fx[f"out/{c.name}/log_px"]) < TOL, c.name
        assert rel_err(lpm[:, j], fx[f"out/{c.name}/log_px_missing"]) < TOL, c.name
    res = eng.elbo(table, 0, B, beta_scale=1.0, tau=1.0, grad=False, training=False, seed=3,
                   want=("reconstruction", "log_px", "log_px_missing"))
    rec, lp2 = res["reconstruction"].cpu().numpy(), res["log_px"].cpu().numpy()
    assert rel_err(lp2.astype(np.float64).sum(axis=1), rec) < 1e-5
    mask = fx["mask"].astype(bool)
    assert not lp2[~mask].any() and not res["log_px_missing"].cpu().numpy()[mask].any()


def test_fused_decode_agrees_with_the_per_layer_path_on_random_shapes(monkeypatch):
    """scripts/decode_fuzz.py, twelve random (schema, K, D, hidden, n) cases: the same rows through component indices
    (k_decode_fused where the shape allows) and through one-hot dense assignments (per-layer GEMMs + k_sample_heads)
    with the same generator keys -- discrete cells and counts bit for bit, numeric values to 2e-4, y_full to 1e-4."""

    import importlib.util
    import os
    import sys

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "decode_fuzz.py")
    spec = importlib.util.spec_from_file_location("decode_fuzz", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["decode_fuzz.py", "12", "3"])
    assert mod.main() == 0
