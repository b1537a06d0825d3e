"""Known-answer checks of the oracle's likelihood heads and KL terms.

Same cases the reference pins in ``tests/test_third_party_alignment.py:146-458``
(fixed head weights, ``np.random.default_rng`` seeds 42/7/21/314/11/17),
``tests/test_vae_components.py:122-198`` (closed-form NLLs) and
``tests/test_elbo_sampling_alignment.py:24-96`` (sampled == expectation ELBO in
the zero-temperature limit).  Expected values come from ``scipy.stats`` /
closed forms in float64 -- an implementation independent of the oracle's.
"""

import math

import numpy as np
from scipy import stats

from oracle import suave_oracle as orc

ATOL, RTOL = 1e-5, 1e-4  # the reference's own tolerances


def _onehot(idx, n):
    return np.eye(n, dtype=np.float32)[idx]


def _softplus64(x):
    return np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0.0)


def test_real_head_known_answer():
    rng = np.random.default_rng(42)
    B, K = 6, 2
    y = rng.normal(size=(B, 1)).astype(np.float32)
    s = _onehot(rng.integers(0, K, B), K)
    x = rng.normal(size=(B, 1)).astype(np.float32)
    mask = (rng.random((B, 1)) > 0.2).astype(np.float32)
    P = {"decoder.heads.c.mean_layer.weight": np.array([[0.7, -0.3, 0.5]], np.float32),
         "decoder.heads.c.var_layer.weight": np.array([[0.4, -0.2]], np.float32)}
    col = orc.Column("c", "real", stats={"mean": 0.0, "std": 1.0})
    out = orc.head_forward(col, K, P, y, s, x, mask)
    f = np.concatenate([y, s], 1).astype(np.float64)
    mean = f @ P["decoder.heads.c.mean_layer.weight"][0].astype(np.float64)
    var = _softplus64(s.astype(np.float64) @ P["decoder.heads.c.var_layer.weight"][0]) + 1e-6
    expect = stats.norm.logpdf(x[:, 0], mean, np.sqrt(var))
    np.testing.assert_allclose(out["log_px"], expect * mask[:, 0], atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(out["log_px_missing"], expect * (1 - mask[:, 0]), atol=ATOL, rtol=RTOL)


def test_pos_head_known_answer():
    rng = np.random.default_rng(21)
    B, K = 5, 2
    mean_log = float(rng.normal(0.2, 0.5))
    std_log = float(rng.uniform(0.5, 1.5))
    y = rng.normal(size=(B, 2)).astype(np.float32)
    s = _onehot(rng.integers(0, K, B), K)
    x_raw = rng.uniform(0.0, 5.0, (B, 1)).astype(np.float32)
    mask = (rng.random((B, 1)) > 0.25).astype(np.float32)
    x = ((np.log1p(x_raw) - mean_log) / std_log).astype(np.float32)
    P = {"decoder.heads.c.mean_layer.weight": np.array([[0.5, -0.2, 0.3, -0.4]], np.float32),
         "decoder.heads.c.var_layer.weight": np.array([[0.25, -0.35]], np.float32)}
    col = orc.Column("c", "pos", y_dim=2, stats={"mean_log": mean_log, "std_log": std_log})
    out = orc.head_forward(col, K, P, y, s, x, mask)
    f = np.concatenate([y, s], 1).astype(np.float64)
    m = (f @ P["decoder.heads.c.mean_layer.weight"][0].astype(np.float64)) * std_log + mean_log
    v = (_softplus64(s.astype(np.float64) @ P["decoder.heads.c.var_layer.weight"][0]) + 1e-6) * std_log**2
    # log-normal density of (1 + x_raw): Normal logpdf of log1p(x) minus log1p(x)
    lx = np.log1p(x_raw[:, 0].astype(np.float64))
    expect = stats.norm.logpdf(lx, m, np.sqrt(v)) - lx
    np.testing.assert_allclose(out["log_px"], expect * mask[:, 0], atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(out["log_px_missing"], expect * (1 - mask[:, 0]), atol=ATOL, rtol=RTOL)


def test_count_head_known_answer():
    rng = np.random.default_rng(314)
    B, K = 6, 3
    counts = rng.integers(0, 5, (B, 1)).astype(np.float32)
    offset = 1.0 if (counts <= 0).any() else 0.0
    x = np.log(counts + offset).astype(np.float32)
    mask = (rng.random((B, 1)) > 0.35).astype(np.float32)
    y = rng.normal(size=(B, 2)).astype(np.float32)
    s = _onehot(rng.integers(0, K, B), K)
    P = {"decoder.heads.c.rate_layer.weight": np.array([[0.3, -0.2, 0.4, -0.1, 0.5]], np.float32)}
    col = orc.Column("c", "count", y_dim=2, stats={"offset": offset})
    out = orc.head_forward(col, K, P, y, s, x, mask)
    f = np.concatenate([y, s], 1).astype(np.float64)
    rate = _softplus64(f @ P["decoder.heads.c.rate_layer.weight"][0].astype(np.float64)) + 1e-6
    expect = stats.poisson.logpmf(counts[:, 0], rate)
    np.testing.assert_allclose(out["log_px"], expect * mask[:, 0], atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(out["log_px_missing"], expect * (1 - mask[:, 0]), atol=ATOL, rtol=RTOL)


def test_cat_head_known_answer():
    rng = np.random.default_rng(7)
    B, K, C = 5, 2, 4
    y = rng.normal(size=(B, 3)).astype(np.float32)
    s = _onehot(rng.integers(0, K, B), K)
    idx = rng.integers(0, C, B)
    x = _onehot(idx, C)
    mask = (rng.random((B, 1)) > 0.3).astype(np.float32)
    W = np.array([[0.2, -0.3, 0.5, -0.1, 0.4], [-0.5, 0.1, 0.2, 0.3, -0.2], [0.1, 0.6, -0.4, 0.2, 0.5]], np.float32)
    P = {"decoder.heads.c.logits_layer.weight": W}
    col = orc.Column("c", "cat", n_classes=C, y_dim=3)
    out = orc.head_forward(col, K, P, y, s, x, mask)
    f = np.concatenate([y, s], 1).astype(np.float64)
    logits = np.concatenate([np.zeros((B, 1)), f @ W.T.astype(np.float64)], 1)
    logp = logits - np.log(np.exp(logits).sum(1, keepdims=True))
    expect = logp[np.arange(B), idx]
    np.testing.assert_allclose(out["log_px"], expect * mask[:, 0], atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(out["log_px_missing"], expect * (1 - mask[:, 0]), atol=ATOL, rtol=RTOL)


def test_ordinal_head_known_answer():
    rng = np.random.default_rng(11)
    B, K, C = 4, 2, 4
    idx = rng.integers(0, C, B)
    thermo = (np.arange(C)[None, :] <= idx[:, None]).astype(np.float32)
    mask = (rng.random((B, 1)) > 0.4).astype(np.float32)
    y = rng.normal(size=(B, 1)).astype(np.float32)
    s = _onehot(rng.integers(0, K, B), K)
    Wp = np.array([[0.6, -0.2], [0.1, 0.4], [-0.3, 0.5]], np.float32)
    Wm = np.array([[0.2, -0.5, 0.3]], np.float32)
    P = {"decoder.heads.c.partition_layer.weight": Wp, "decoder.heads.c.mean_layer.weight": Wm}
    col = orc.Column("c", "ordinal", n_classes=C, y_dim=1)
    out = orc.head_forward(col, K, P, y, s, thermo, mask)
    thr = np.cumsum(_softplus64(s.astype(np.float64) @ Wp.T) + 1e-6, axis=1)
    mean = np.concatenate([y, s], 1).astype(np.float64) @ Wm[0]
    cdf = np.concatenate([np.zeros((B, 1)), stats.logistic.cdf(thr - mean[:, None]), np.ones((B, 1))], 1)
    probs = np.clip(np.diff(cdf, axis=1), 1e-6, 1.0)
    expect = np.log(probs[np.arange(B), idx])
    np.testing.assert_allclose(out["log_px"], expect * mask[:, 0], atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(out["log_px_missing"], expect * (1 - mask[:, 0]), atol=ATOL, rtol=RTOL)


def test_categorical_kl_known_answer():
    rng = np.random.default_rng(11)
    lq = rng.normal(size=(6, 4)).astype(np.float32)
    kl = orc.kl_categorical(lq, np.zeros(4, np.float32))
    p = np.exp(lq - lq.max(1, keepdims=True)).astype(np.float64)
    p /= p.sum(1, keepdims=True)
    expect = math.log(4.0) - stats.entropy(p, axis=1)
    np.testing.assert_allclose(kl, expect, atol=1e-5, rtol=1e-5)


def test_gaussian_mixture_kl_known_answer():
    rng = np.random.default_rng(17)
    B, K, D = 5, 3, 4
    logits = rng.normal(size=(B, K)).astype(np.float32)
    w = orc.softmax(logits)
    mu_q = rng.normal(size=(B, K, D)).astype(np.float32)
    lv_q = rng.normal(size=(B, K, D)).astype(np.float32)
    mu_p = rng.normal(scale=0.05, size=(K, D)).astype(np.float32)
    kl = orc.kl_normal_mixture(mu_q, lv_q, mu_p, np.zeros((K, D), np.float32), w)
    comp = 0.5 * (np.exp(lv_q.astype(np.float64)) + (mu_q - mu_p[None]) ** 2 - lv_q - 1.0).sum(-1)
    np.testing.assert_allclose(kl, (w * comp).sum(-1), atol=1e-5, rtol=1e-5)


def test_closed_form_nlls():
    # test_vae_components.py:122-198
    one = np.ones((1, 1), np.float32)
    col = orc.Column("c", "real", stats={"mean": 0.0, "std": 1.0})
    P = {"decoder.heads.c.mean_layer.weight": np.array([[0.0, 1.5]], np.float32),
         "decoder.heads.c.var_layer.weight": np.array([[math.log(math.e - 1.0)]], np.float32)}  # softplus -> 1
    out = orc.head_forward(col, 1, P, np.zeros((1, 1), np.float32), one, 0.5 * one, one)
    assert abs(float(out["log_px"][0]) - stats.norm.logpdf(0.5, 1.5, math.sqrt(1 + 1e-6))) < 1e-5


def test_sampled_elbo_equals_expectation_at_zero_temperature():
    # test_elbo_sampling_alignment.py:24-96
    col = orc.Column("feature", "real", y_dim=1, stats={"mean": 0.0, "std": 1.0})
    om = orc.OracleModel([col], n_components=2, latent_dim=2, hidden=(), dropout=0.0)
    P = {"decoder.y_projection.weight": np.array([[0.8, -0.2]], np.float32),
         "decoder.y_projection.bias": np.zeros(1, np.float32),
         "decoder.heads.feature.mean_layer.weight": np.array([[0.5, -0.3, 0.4]], np.float32),
         "decoder.heads.feature.var_layer.weight": np.array([[0.2, -0.1]], np.float32)}
    x = np.array([[0.5], [-1.0]], np.float32)
    data = {"feature": x}
    masks = {"feature": np.ones((2, 1), np.float32)}
    logits = np.array([[40.0, -40.0], [-40.0, 40.0]], np.float32)
    mu = np.array([[[0.1, -0.2], [0.3, -0.1]], [[-0.4, 0.2], [0.6, -0.3]]], np.float32)
    lv = np.full((2, 2, 2), -1e6, np.float32)
    probs = orc.softmax(logits)
    k = logits.argmax(1)
    zero2 = np.zeros((2, 2), np.float32)
    s = np.eye(2, dtype=np.float32)[k]
    z = mu[np.arange(2), k]  # exp(0.5 * -1e6) == 0
    y, _ = orc.decoder_trunk_forward(om, P, z)
    recon = orc.heads_forward(om, P, y, s, data, masks)["recon"]
    cat_kl = orc.kl_categorical(logits, np.zeros(2, np.float32))
    g_new = orc.kl_normal_vs_normal(z, lv[np.arange(2), k], zero2[k], zero2[k])
    elbo_new = recon - (cat_kl + g_new)
    lp = []
    for j in range(2):
        sj = np.zeros((2, 2), np.float32)
        sj[:, j] = 1
        yj, _ = orc.decoder_trunk_forward(om, P, mu[:, j])
        lp.append(orc.heads_forward(om, P, yj, sj, data, masks)["recon"])
    recon_exp = (probs * np.stack(lp, 1)).sum(1)
    g_exp = orc.kl_normal_mixture(mu, lv, zero2, zero2, probs)
    elbo_exp = recon_exp - (cat_kl + g_exp)
    np.testing.assert_allclose(elbo_new, elbo_exp, atol=1e-5, rtol=1e-4)


def test_poisson_map_is_poisson_in_every_range():
    """The build-defined (rate, u, n) -> count map (oracle/sampling_oracle.poisson_from_noise == sample_col.h
    poisson_from_noise; torch.poisson cannot be replayed) against scipy.stats.poisson.
      * rate < 64 (sequential inversion from 0) and 64 <= rate < 65536 (inversion from the mode): on the uniform grid
        u = (i + 1/2) / 2^20 the frequency of EVERY value equals its pmf to the grid's resolution -- the map is an exact
        Poisson sampler up to binary32 CDF resolution; first three moments and the 97.5 % quantile follow.
      * rate >= 65536: normal draw with the first Cornish-Fisher term; mean, variance and skewness 1/sqrt(rate)."""

    from scipy import stats

    from oracle import sampling_oracle as sorc

    N = 1 << 20
    u = ((np.arange(N) + 0.5) / N).astype(np.float32)
    zero = np.zeros(N, np.float32)
    for rate in (3.0, 20.0, 63.9, 64.0, 100.0, 100.7, 1000.0, 1000.999):
        k = sorc.poisson_from_noise(np.full(N, rate, np.float32), u, zero).astype(np.int64)
        r = float(np.float32(rate))
        lo = int(k.min())
        freq = np.bincount(k - lo) / N
        pmf = stats.poisson.pmf(np.arange(lo, lo + len(freq)), r)
        assert np.abs(freq - pmf).max() < 3e-6, rate
        m, v = k.mean(), k.var()
        sk = ((k - m) ** 3).mean() / v ** 1.5
        assert abs(m - r) < 1e-3 * max(1.0, np.sqrt(r)) and abs(v - r) < 2e-3 * r and abs(sk - r ** -0.5) < 2e-3, rate
        q = stats.poisson.ppf(0.975, r)
        assert abs((k <= q).mean() - stats.poisson.cdf(q, r)) < 3e-6 * (q - lo + 1), rate
        assert np.quantile(k, 0.975, method="inverted_cdf") == q, rate
    rng = np.random.default_rng(12)
    z = rng.standard_normal(N).astype(np.float32)
    for rate in (65536.0, 1.0e6):
        k = sorc.poisson_from_noise(np.full(N, rate, np.float32), u, z).astype(np.float64)
        m, v = k.mean(), k.var()
        sk = ((k - m) ** 3).mean() / v ** 1.5
        assert abs(m - rate) < 4 * np.sqrt(rate / N) + 0.1 and abs(v - rate) < 6e-3 * rate
        assert abs(sk - rate ** -0.5) < 4 * np.sqrt(6.0 / N), (rate, sk)


def test_poisson_map_on_the_reference_rates_of_the_bigcount_fixture():
    """decode_bigcount6: the reference's Decoder.forward on count heads whose rates span 2.5 .. 200 000 (every range
    of the map), with the map in place of torch.poisson (make_golden.NoiseTape).  The oracle's map on the
    REFERENCE's rates and the recorded draws must give the reference's samples bit for bit, in every range."""

    from oracle import sampling_oracle as sorc
    from tests.helpers import load_golden

    fx = load_golden("decode_bigcount6")
    seen = set()
    for c in ("n0", "n1", "n2", "n3"):
        rate = fx[f"out/{c}/params/rate"].reshape(-1)
        got = sorc.poisson_from_noise(rate, fx[f"noise/{c}/uniform"].reshape(-1), fx[f"noise/{c}/normal"].reshape(-1))
        assert np.array_equal(np.rint(got).astype(np.int64), fx[f"out/{c}/sample"].reshape(-1).astype(np.int64)), c
        seen |= {0 if r < 64 else 1 if r < 65536 else 2 for r in rate}
    assert seen == {0, 1, 2}
