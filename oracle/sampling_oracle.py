"""CPU restatement (numpy) of SUAVE's generative decode: latent draws, head
samplers and frame conversion.

TEST INFRASTRUCTURE ONLY -- see the header of ``suave_oracle.py``; the same
rules apply (never imported by ``suave_amd/``).

Bit-exactness contract (BASELINE.json north_star: "bit-exact for sampled
categorical/ordinal/count values and component assignments" on identical
inputs and injected noise).  A discrete decision is an argmax / threshold
compare, so it is reproducible across CPU and GPU only if every value that
feeds it is computed by the same sequence of correctly-rounded IEEE-754
binary32 operations.  ``exp``/``log`` of libm, numpy and the GPU's hardware
approximations differ in the last bits, therefore the decision path uses the
*deterministic* ``det_exp`` / ``det_log`` below: fdlibm-style polynomial
kernels written with individually rounded ``+ - * /`` only (no fused
multiply-add, no library call).  The HIP kernels evaluate the identical
operation sequence (``suave_amd/csrc/detmath.h``), so given the same
``y_full`` and noise the sampled integers agree bit for bit at any size.
Against the reference (libm/SLEEF transcendentals inside torch) the decisions
agree except at floating-point near-ties (probability ~1e-7 per cell), which is
what ``tests/golden/decode_*.npz`` pin.

Reference call sites restated here: ``distributions.py:67-80,112-117,130-134,
167-174`` (head samplers), ``decoder.py:90-91,139,172,214-216,258`` (where the
heads call them), ``sampling.py:38-56`` (prior latents), ``model.py:3042-3114``
(conditional latents), ``model.py:3304-3336`` (posterior draws) and
``sampling.py:138-179`` (frame conversion).
"""

from __future__ import annotations

from typing import Dict, Mapping

import numpy as np

F32 = np.float32
EPS = F32(1e-6)


# ---------------------------------------------------------------------------
# deterministic binary32 math (mirrors suave_amd/csrc/detmath.h op for op)
# ---------------------------------------------------------------------------
_LN2_HI = F32(6.9313812256e-01)
_LN2_LO = F32(9.0580006145e-06)
_INV_LN2 = F32(1.4426950216e+00)
_EXP_P1 = F32(1.6666625440e-01)
_EXP_P2 = F32(-2.7667332906e-03)
_LG1 = F32(0.66666662693)
_LG2 = F32(0.40000972152)
_LG3 = F32(0.28498786688)
_LG4 = F32(0.24279078841)
_SQRT2 = F32(1.41421353817)
_FLT_MIN = F32(1.17549435e-38)


def det_exp(x):
    """exp(x) in binary32 from + - * / only (fdlibm expf kernel)."""

    x = np.asarray(x, F32)
    xc = np.minimum(np.maximum(x, F32(-87.0)), F32(88.0))
    n = np.rint(xc * _INV_LN2).astype(F32)
    hi = xc - n * _LN2_HI
    lo = n * _LN2_LO
    r = hi - lo
    t = r * r
    c = r - t * (_EXP_P1 + t * _EXP_P2)
    y = F32(1.0) - ((lo - (r * c) / (F32(2.0) - c)) - hi)
    bits = y.view(np.int32) + (n.astype(np.int32) << 23)
    out = bits.view(F32)
    return np.where(x < F32(-87.0), F32(0.0), out).astype(F32)


def det_log(x):
    """log(x) for positive normal binary32 x from + - * / only (fdlibm logf kernel)."""

    x = np.maximum(np.asarray(x, F32), _FLT_MIN)
    ix = x.view(np.int32)
    e = (ix >> 23) - 127
    m = ((ix & 0x007FFFFF) | 0x3F800000).view(F32)
    big = m > _SQRT2
    m = np.where(big, m * F32(0.5), m).astype(F32)
    e = (e + big.astype(np.int32)).astype(F32)
    f = m - F32(1.0)
    s = f / (F32(2.0) + f)
    z = s * s
    w = z * z
    t1 = w * (_LG2 + w * _LG4)
    t2 = z * (_LG1 + w * _LG3)
    R = t2 + t1
    hfsq = (F32(0.5) * f) * f
    lm = f - (hfsq - s * (hfsq + R))
    return (e * _LN2_HI + (lm + e * _LN2_LO)).astype(F32)


def det_softplus(x):
    """softplus with torch's threshold: x if x > 20 else log(1 + exp(x))."""

    x = np.asarray(x, F32)
    sp = det_log(F32(1.0) + det_exp(np.minimum(x, F32(20.0))))
    return np.where(x > F32(20.0), x, sp).astype(F32)


def det_sigmoid(x):
    x = np.asarray(x, F32)
    return (F32(1.0) / (F32(1.0) + det_exp(-x))).astype(F32)


def det_dot(w_y, w_s, y, s):
    """Sequential, individually rounded dot of [y ; s] with [w_y ; w_s]."""

    acc = np.zeros(y.shape[0], F32)
    for i in range(y.shape[1]):
        acc = acc + F32(w_y[i]) * y[:, i].astype(F32)
    for j in range(s.shape[1]):
        acc = acc + F32(w_s[j]) * s[:, j].astype(F32)
    return acc.astype(F32)


# ---------------------------------------------------------------------------
# build-defined noise -> value maps
# ---------------------------------------------------------------------------
POISSON_SWITCH = F32(64.0)
POISSON_HUGE = F32(65536.0)
POISSON_MAX_ITERS = 512
POISSON_WALK_MAX = 4096
POISSON_WALK_STOP = F32(1e-9)


def _poisson_mode_pmf(rate):
    """p(k0; rate) at k0 = floor(rate) >= 64, binary32 with individually rounded operations:
    p = exp(k0 log1p(d / k0) - d - c) / sqrt(2 pi k0),  d = rate - k0 in [0, 1) (exact),
    k0 log1p(x) - d = d x (-1/2 + x/3 - x^2/4 + x^3/5) for x = d / k0 <= 1/64 (next term < 2e-10), and the
    Stirling correction c = 1/(12 k0) - 1/(360 k0^3) (next term < 1e-12).  No large terms cancel."""

    rate = np.asarray(rate, F32)
    k0 = np.floor(rate).astype(F32)
    d = (rate - k0).astype(F32)
    x = (d / k0).astype(F32)
    poly = (F32(0.2) * x).astype(F32)
    poly = (poly + F32(-0.25)).astype(F32)
    poly = (poly * x).astype(F32)
    poly = (poly + F32(0.33333334)).astype(F32)
    poly = (poly * x).astype(F32)
    poly = (poly + F32(-0.5)).astype(F32)
    e = ((d * x).astype(F32) * poly).astype(F32)
    ik = (F32(1.0) / k0).astype(F32)
    ik2 = (ik * ik).astype(F32)
    c = (ik * (F32(0.083333336) - (ik2 * F32(0.0027777778)).astype(F32)).astype(F32)).astype(F32)
    den = np.sqrt((F32(6.2831855) * k0).astype(F32)).astype(F32)
    return k0, (det_exp((e - c).astype(F32)) / den).astype(F32)


def poisson_mode_walk(rate, u):
    """Exact Poisson draw from ONE uniform for 64 <= rate < 65536: inversion over the values in the order
    k0, k0+1, k0-1, k0+2, k0-2, ... (k0 = floor(rate), the mode): value k owns the next p(k) of [0, 1) in that order,
    so P(k) = p(k) whatever the order, and starting at the mode the walk ends after ~1.6 sqrt(rate) values.
    p(k0) from `_poisson_mode_pmf`, the neighbours by the recurrences p(k+1) = p(k) rate / (k+1) and
    p(k-1) = p(k) k / rate.  The walk stops when both tails fall below 1e-9 (they no longer move a binary32 sum
    near 1); a u beyond the accumulated mass (rounding: < 1e-5) takes the mode."""

    rate = np.asarray(rate, F32)
    u = np.asarray(u, F32)
    k0, p0 = _poisson_mode_pmf(rate)
    acc = p0.copy()
    out = k0.copy()
    go = u > acc
    pu, pd = p0.copy(), p0.copy()
    for i in range(1, POISSON_WALK_MAX + 1):
        if not go.any():
            break
        fi = F32(i)
        ku = (k0 + fi).astype(F32)
        pu = ((pu * rate).astype(F32) / ku).astype(F32)
        acc_u = (acc + pu).astype(F32)
        hit_u = go & ~(u > acc_u)
        out = np.where(hit_u, ku, out).astype(F32)
        go = go & ~hit_u
        kd_prev = (k0 - (fi - F32(1.0))).astype(F32)   # the value whose probability pd holds
        pd = ((pd * kd_prev).astype(F32) / rate).astype(F32)
        acc_d = (acc_u + pd).astype(F32)
        hit_d = go & ~(u > acc_d)
        out = np.where(hit_d, (kd_prev - F32(1.0)).astype(F32), out).astype(F32)
        go = go & ~hit_d
        acc = acc_d
        go = go & ((pu > POISSON_WALK_STOP) | (pd > POISSON_WALK_STOP))
    return out.astype(F32)


def poisson_from_noise(rate, u, n):
    """(rate, uniform u, normal n) -> Poisson count, as binary32.

    ``torch.poisson`` consumes a data-dependent number of uniforms and cannot be
    replayed, so the build defines its own map and the oracle
    evaluates the same map on the reference's ``rate``.  Three ranges:
      * ``rate < 64``: CDF inversion by sequential search from 0 (exact Poisson up to binary32 CDF resolution);
      * ``64 <= rate < 65536``: inversion from the mode (`poisson_mode_walk`), exact in the same sense;
      * ``rate >= 65536``: the normal draw with its first Cornish-Fisher (skewness) term,
        floor(rate + sqrt(rate) n + (n^2 - 1)/6 + 1/2): the error of this expansion is O(1/rate) < 2e-5 in
        distribution, and binary32 counts are not all representable much above this range anyway.
    """

    rate = np.asarray(rate, F32)
    u = np.asarray(u, F32)
    n = np.asarray(n, F32)
    small = rate < POISSON_SWITCH
    r = np.where(small, rate, F32(1.0)).astype(F32)
    p = det_exp(-r)
    cdf = p.copy()
    k = np.zeros(rate.shape, F32)
    go = small & (u > cdf)
    for i in range(1, POISSON_MAX_ITERS + 1):
        if not go.any():
            break
        fi = F32(i)
        p_new = (p * r) / fi
        cdf_new = cdf + p_new
        k = np.where(go, fi, k).astype(F32)
        p = np.where(go, p_new, p).astype(F32)
        cdf = np.where(go, cdf_new, cdf).astype(F32)
        go = go & (u > cdf) & ((p > F32(1e-30)) | (fi <= r))
    mid = ~small & (rate < POISSON_HUGE)
    walk = np.zeros(rate.shape, F32)
    if mid.any():
        walk[mid] = poisson_mode_walk(rate[mid], u[mid])
    skew = (((n * n).astype(F32) - F32(1.0)).astype(F32) * F32(0.16666667)).astype(F32)
    big = (rate + (np.sqrt(rate).astype(F32) * n).astype(F32)).astype(F32)
    big = np.floor(((big + skew).astype(F32) + F32(0.5)).astype(F32))
    big = np.maximum(big, F32(0.0))
    return np.where(small, k, np.where(mid, walk, big)).astype(F32)


def argmax_first(score):
    return np.argmax(score, axis=-1).astype(np.int64)


def cat_from_noise(logits, e):
    """Categorical(logits).sample() with one draw per row == argmax(p / e),
    e ~ Exp(1) (torch's exponential race); evaluated as the equivalent
    Gumbel-max ``argmax(logits - log e)`` with ``det_log``."""

    return argmax_first(np.asarray(logits, F32) - det_log(e))


def probs_race(probs, e):
    """Categorical(probs=probs).sample(): argmax(p / e) (the per-row
    normalisation is a positive factor and drops out)."""

    return argmax_first(np.asarray(probs, F32) / np.asarray(e, F32))


def component_from_uniform(cdf, u):
    """Prior component draw: k = #{j < K-1 : u >= cdf_j} for cumulative
    probabilities ``cdf`` of softmax(prior_logits) (sampling.py:49-50)."""

    cdf = np.asarray(cdf, F32)
    u = np.asarray(u, F32)
    return (u[:, None] >= cdf[None, :-1]).sum(axis=1).astype(np.int64)


def prior_cdf(prior_logits):
    """Host-side cumulative distribution of the mixture prior (float64 softmax,
    rounded once to binary32)."""

    lg = np.asarray(prior_logits, np.float64)
    p = np.exp(lg - lg.max())
    p /= p.sum()
    return np.cumsum(p).astype(F32)


# ---------------------------------------------------------------------------
# deterministic head evaluation + sampling  (decoder.py:64-265)
# ---------------------------------------------------------------------------
def sample_heads(model, P, y_full, s, noise: Mapping[str, Mapping[str, np.ndarray]]):
    """Per-column sample draws given ``y_full`` and assignments ``s``.

    ``noise[col]`` holds ``normal`` (real/pos/count), ``uniform`` (count) or
    ``exp`` ``[B,C]`` (cat/ordinal).  Returns ``{col: values}`` in the
    reference's *frame* convention (sampling.py:138-179): float32 for real/pos,
    int64 counts (``round().clamp(min=0)``), int64 codes for cat, int64 level
    for ordinal; plus the head parameters under ``"_params"``.
    """

    y_full = np.asarray(y_full, F32)
    s = np.asarray(s, F32)
    out: Dict[str, object] = {}
    params: Dict[str, Dict[str, np.ndarray]] = {}
    K = model.K
    for col in model.columns:
        sl = model.y_slices[col.name]
        y = y_full[:, sl]
        yd = y.shape[1]
        pfx = f"decoder.heads.{col.name}."
        st = col.stats or {}
        nz = noise[col.name]
        if col.type in ("real", "pos"):
            Wm = np.asarray(P[pfx + "mean_layer.weight"], F32)[0]
            Wv = np.asarray(P[pfx + "var_layer.weight"], F32)[0]
            a = det_dot(Wm[:yd], Wm[yd:], y, s)
            bv = det_dot(Wv[:0], Wv, y[:, :0], s)
            v = det_softplus(bv) + EPS
            eps = np.asarray(nz["normal"], F32).reshape(-1)
            if col.type == "real":  # decoder.py:90-91, distributions.py:67-72
                shift = F32(st.get("mean", 0.0))
                scale = F32(st.get("std", 1.0))
                std = np.sqrt(np.maximum(v, EPS))
                out[col.name] = ((a + eps * std) * scale + shift).astype(F32)
                params[col.name] = {"mean": a * scale + shift, "var": v * (scale * scale)}
            else:  # decoder.py:139, distributions.py:112-117
                shift = F32(st.get("mean_log", 0.0))
                scale = F32(st.get("std_log", 1.0))
                mean = a * scale + shift
                vs = v * (scale * scale)
                std = np.sqrt(np.maximum(vs, EPS))
                out[col.name] = (det_exp(mean + eps * std) - F32(1.0)).astype(F32)
                params[col.name] = {"mean_log": mean, "var_log": vs}
        elif col.type == "count":  # decoder.py:172, distributions.py:130-134
            Wr = np.asarray(P[pfx + "rate_layer.weight"], F32)[0]
            rho = det_dot(Wr[:yd], Wr[yd:], y, s)
            rate = det_softplus(rho) + EPS
            cnt = poisson_from_noise(rate, np.asarray(nz["uniform"], F32).reshape(-1),
                                     np.asarray(nz["normal"], F32).reshape(-1))
            out[col.name] = np.maximum(np.rint(cnt), 0).astype(np.int64)  # sampling.py:154-155
            params[col.name] = {"rate": rate}
        elif col.type == "cat":  # decoder.py:214-216, distributions.py:75-80
            W = np.asarray(P[pfx + "logits_layer.weight"], F32)
            C = int(col.n_classes)
            logits = np.zeros((y.shape[0], C), F32)
            for c in range(1, C):
                logits[:, c] = det_dot(W[c - 1, :yd], W[c - 1, yd:], y, s)
            out[col.name] = cat_from_noise(logits, np.asarray(nz["exp"], F32))
            params[col.name] = {"logits": logits}
        elif col.type == "ordinal":  # decoder.py:258, distributions.py:137-151,167-174
            Wp = np.asarray(P[pfx + "partition_layer.weight"], F32)
            Wm = np.asarray(P[pfx + "mean_layer.weight"], F32)[0]
            C = int(col.n_classes)
            mean_o = det_dot(Wm[:yd], Wm[yd:], y, s)
            probs = np.zeros((y.shape[0], C), F32)
            thr = np.zeros(y.shape[0], F32)
            prev = np.zeros(y.shape[0], F32)
            for c in range(C - 1):
                part = det_dot(Wp[c, :0], Wp[c], y[:, :0], s)
                thr = thr + (det_softplus(part) + EPS)
                sg = det_sigmoid(thr - mean_o)
                probs[:, c] = sg - prev
                prev = sg
            probs[:, C - 1] = F32(1.0) - prev
            probs = np.minimum(np.maximum(probs, EPS), F32(1.0))
            # thermometer -> (thermo > 0.5).sum - 1 clamped at 0 == sampled index (sampling.py:166-168)
            out[col.name] = probs_race(probs, np.asarray(nz["exp"], F32))
            params[col.name] = {"probs": probs, "mean": mean_o}
        else:
            raise ValueError(col.type)
    out["_params"] = params
    return out


# ---------------------------------------------------------------------------
# latent draws
# ---------------------------------------------------------------------------
def prior_latents(prior_logits, prior_mu, prior_logvar, u, eps):
    """sampling.sample_prepared_mixture_latents -- sampling.py:38-56."""

    k = component_from_uniform(prior_cdf(prior_logits), u)
    mu = np.asarray(prior_mu, F32)[k]
    lv = np.asarray(prior_logvar, F32)[k]
    z = mu + np.asarray(eps, F32) * np.exp(F32(0.5) * lv)
    return z.astype(F32), k


def conditional_latents(class_rows_offsets, class_rows, labels, train_probs, train_mu, train_logvar,
                        u_row, e_comp, eps):
    """SUAVE._draw_conditional_latents -- model.py:3042-3114.

    ``class_rows[class_rows_offsets[c]:class_rows_offsets[c+1]]`` lists the
    training rows of class ``c`` (``np.flatnonzero(train_targets == c)``,
    model.py:3080-3082).  Per requested label: row = candidates[floor(u * n)]
    (``torch.randint``, model.py:3088), component = exponential race on the
    cached posterior probabilities (model.py:3092-3094), reparameterised draw
    from the cached component (model.py:3095-3105).
    """

    labels = np.asarray(labels, np.int64)
    start = np.asarray(class_rows_offsets, np.int64)[labels]
    count = np.asarray(class_rows_offsets, np.int64)[labels + 1] - start
    pick = np.minimum((np.asarray(u_row, F32) * count.astype(F32)).astype(np.int64), count - 1)
    row = np.asarray(class_rows, np.int64)[start + pick]
    k = probs_race(np.asarray(train_probs, F32)[row], e_comp)
    r = np.arange(row.shape[0])
    mu = np.asarray(train_mu, F32)[row, k]
    lv = np.asarray(train_logvar, F32)[row, k]
    z = mu + np.asarray(eps, F32) * np.exp(F32(0.5) * lv)
    del r
    return z.astype(F32), k, row


def posterior_draws(logits, mu, logvar, e, eps):
    """SUAVE._posterior_draws -- model.py:3304-3336, with the L component draws
    per row taken by the exponential race on ``e [L,B,K]`` (Gumbel-max on the
    logits) and ``eps [L,B,D]``."""

    L = e.shape[0]
    B = logits.shape[0]
    k = np.stack([cat_from_noise(logits, e[l]) for l in range(L)])  # [L,B]
    rows = np.arange(B)
    mu_s = np.asarray(mu, F32)[rows[None, :], k]
    lv_s = np.asarray(logvar, F32)[rows[None, :], k]
    z = mu_s + np.asarray(eps, F32) * np.exp(F32(0.5) * lv_s)
    return k, z.astype(F32)
