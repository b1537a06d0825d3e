"""End-to-end checks of the SUAVE-compatible class on the GPU (API semantics of the reference's
tests/test_minimal.py, test_impute.py:149-236 and test_posterior_predictive.py:58-129)."""

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def _frame(n, seed=0, with_nan=True):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 2, n)
    df = pd.DataFrame({
        "age": rng.normal(50 + 10 * g, 8, n),
        "lab": rng.lognormal(1.0 + 0.5 * g, 0.4, n),
        "visits": rng.poisson(2 + 3 * g, n).astype(float),
        "sex": rng.integers(0, 2, n).astype(float),
        "grade": np.clip(rng.integers(0, 3, n) + g, 0, 3).astype(float),
    })
    if with_nan:
        for c in df.columns:
            df.loc[rng.random(n) < 0.1, c] = np.nan
    return df, g


def _schema():
    from suave_amd.types import Schema

    return Schema({"age": {"type": "real"}, "lab": {"type": "pos"}, "visits": {"type": "count"},
                   "sex": {"type": "cat", "n_classes": 2}, "grade": {"type": "ordinal", "n_classes": 4}})


@pytest.mark.parametrize("behaviour", ["unsupervised", "supervised"])
def test_fit_encode_sample_impute_predict(behaviour):
    import torch
    from suave_amd.model import SUAVE

    df, g = _frame(1200, seed=1)
    model = SUAVE(_schema(), behaviour=behaviour, latent_dim=8, n_components=3, hidden_dims=(32, 16), dropout=0.1,
                  batch_size=128, learning_rate=3e-3, random_state=0)
    model.fit(df, y=g if behaviour == "supervised" else None, epochs=12, kl_warmup_epochs=4, head_epochs=4,
              finetune_epochs=4, decoder_refine_epochs=2)
    hist = model._train_history
    assert len(hist) == 12 and len(model._warmup_val_history) == 12
    assert hist[-1]["reconstruction"] > hist[0]["reconstruction"] + 0.5  # training improves the log-likelihood
    assert all(np.isfinite(list(h.values())).all() for h in hist + model._warmup_val_history)
    # encode
    z = model.encode(df)
    assert z.shape == (len(df), 8) and np.isfinite(z).all()
    comps = model.encode(df.iloc[:10], return_components=True)
    assert comps["assignments"].shape == (10, 3) and np.allclose(comps["assignments"].sum(1), 1.0)
    assert model.encode(df.iloc[:0]).shape == (0, 8)
    # sample: dtypes of sampling.decoder_outputs_to_frame, chunking invariance
    s1 = model.sample(500, batch_size=500, seed=7)
    s2 = model.sample(500, batch_size=64, seed=7)
    pd.testing.assert_frame_equal(s1, s2)
    assert s1["age"].dtype == np.float32 and s1["visits"].dtype == np.int64 and (s1["visits"] >= 0).all()
    assert isinstance(s1["sex"].dtype, pd.CategoricalDtype) and s1["grade"].cat.ordered
    assert abs(s1["age"].mean() - df["age"].mean()) < 15 and s1["lab"].min() > -1.0
    if behaviour == "supervised":
        sc = model.sample(300, conditional=True, y=np.array([0, 1] * 150), seed=3)
        assert len(sc) == 300
        with pytest.raises(ValueError):
            model.sample(2, conditional=True, y=np.array([0, 5]))
    # impute: observed cells preserved, no NaN left (test_impute.py:149-186)
    sub = df.iloc[:200]
    imp = model.impute(sub)
    assert imp.isna().sum().sum() == 0 and list(imp.columns) == list(sub.columns)
    obs = sub["age"].notna()
    assert np.allclose(imp.loc[obs, "age"].astype(float), sub.loc[obs, "age"].astype(float), rtol=1e-6)
    for strat in ("soft", "hard", "sample"):
        assert model.impute(sub, assignment_strategy=strat).isna().sum().sum() == 0
    with pytest.raises(ValueError):
        model.impute(sub, assignment_strategy="bogus")
    full = model.impute(sub, only_missing=False)
    assert full.shape == sub.shape
    # posterior predictive probabilities (test_posterior_predictive.py:58-72)
    p = model.predict_proba(sub, attr="grade", L=16)
    assert p.shape == (200, 4) and torch.allclose(p.sum(dim=1), torch.ones(200), atol=1e-5) and bool((p > 0).all())
    p2 = model.predict_proba(sub, attr=3, L=16)
    assert p2.shape == (200, 2)
    with pytest.raises(ValueError):
        model.predict_proba(sub, attr="age")
    with pytest.raises(ValueError):
        model.predict_proba(sub, attr="grade", L=0)


@pytest.mark.parametrize("refine_mode", ["decoder_prior", "prior_em_decoder", "prior_em_only", "decoder_only"])
def test_supervised_schedule_classifier_and_persistence(refine_mode, tmp_path):
    """Head phase, joint fine-tuning, decoder refine (all four modes), classifier prediction, calibration,
    confidence intervals, MC imputation and save/load (reference tests: test_training_schedule.py,
    test_calibration.py, test_posterior_predictive.py:74-129, test_model_serialization.py)."""

    import torch
    from suave_amd.model import SUAVE

    df, g = _frame(1500, seed=2)
    labels = np.where(g == 1, "pos", "neg")
    model = SUAVE(_schema(), latent_dim=8, n_components=2, hidden_dims=(32, 16), dropout=0.05, batch_size=128,
                  learning_rate=3e-3, random_state=1, decoder_refine_mode=refine_mode)
    model.fit(df, y=labels, warmup_epochs=8, kl_warmup_epochs=3, head_epochs=6, finetune_epochs=5, decoder_refine_epochs=3)
    assert len(model._train_history) == 8 and 1 <= len(model._joint_history) <= 5
    assert model._joint_val_metrics is not None and np.isfinite(model._joint_val_metrics["joint_objective"])
    assert 0.1 <= model.classification_loss_weight <= 1000.0
    assert len(model._refine_history) == (0 if refine_mode == "prior_em_only" else 3) or len(model._refine_history) >= 1
    proba = model.predict_proba(df)
    assert proba.shape == (len(df), 2) and np.allclose(proba.sum(axis=1), 1.0, atol=1e-5)
    pred = model.predict(df)
    assert set(np.unique(pred)) <= {"neg", "pos"}
    assert (pred == labels).mean() > 0.7  # the label shifts age / lab / visits / grade
    assert model._joint_val_metrics["auroc"] > 0.75
    model.calibrate(df.iloc[:400], labels[:400])
    assert model._is_calibrated and model._temperature_scaler.temperature > 0
    assert np.allclose(model.predict_proba(df).sum(axis=1), 1.0, atol=1e-5)
    with pytest.raises(ValueError):
        model.calibrate(df.iloc[:10], labels[:5])
    # posterior predictive statistics (test_posterior_predictive.py:74-129)
    sub = df.iloc[:120]
    ci = model.predict_confidence_interval(sub, "visits", L=64, return_samples=True)
    assert ci["samples"].shape == (120, 64) and bool((ci["lower"] <= ci["upper"]).all())
    assert torch.allclose(ci["point"], ci["samples"].mean(dim=1), atol=1e-4)  # count -> mean
    ci_pos = model.predict_confidence_interval(sub, "lab", L=64, return_samples=True)
    assert torch.allclose(ci_pos["point"], ci_pos["samples"].median(dim=1).values, atol=1e-5)  # pos -> median
    assert torch.equal(model.predict(sub, attr="visits", L=64), ci["point"])  # same seed -> same draws
    with pytest.raises(ValueError):
        model.predict_confidence_interval(sub, "sex")
    masked = sub.copy()
    # the attribute is always decoded as missing: its value must not matter (1e6 would be beyond the fp16 planes after
    # normalisation and switch the arithmetic: test_values_beyond_fp16_switch_the_arithmetic_instead_of_returning_nan)
    masked["age"] = 1e5
    a = model.predict_confidence_interval(sub, "age", L=32)["point"]
    b = model.predict_confidence_interval(masked, "age", L=32)["point"]
    assert torch.equal(a, b)
    imp = model.impute_mc(sub, L=16)
    assert imp.isna().sum().sum() == 0
    obs = sub["lab"].notna()
    assert np.allclose(imp.loc[obs, "lab"].astype(float), sub.loc[obs, "lab"].astype(float), rtol=1e-6)
    # persistence
    path = tmp_path / "model.pt"
    model.save(path)
    clone = SUAVE.load(path)
    assert np.allclose(clone.predict_proba(df), model.predict_proba(df), atol=1e-6)
    pd.testing.assert_frame_equal(clone.sample(200, seed=5), model.sample(200, seed=5))
    pd.testing.assert_frame_equal(clone.sample(100, conditional=True, y=np.array(["pos", "neg"] * 50), seed=5),
                                  model.sample(100, conditional=True, y=np.array(["pos", "neg"] * 50), seed=5))
    assert np.allclose(clone.encode(sub), model.encode(sub), atol=1e-6)


def test_unfitted_and_validation_errors():
    from suave_amd.model import SUAVE

    df, _ = _frame(10)
    m = SUAVE(_schema(), behaviour="unsupervised")
    with pytest.raises(RuntimeError):
        m.encode(df)
    with pytest.raises(RuntimeError):
        m.sample(3)
    with pytest.raises(RuntimeError):
        m.impute(df)
    with pytest.raises(ValueError):
        SUAVE(_schema(), behaviour="supervised").fit(df)  # targets required
    with pytest.raises(ValueError):
        SUAVE(_schema(), n_components=0)


def test_m20_shaped_fit_with_reference_defaults():
    """BASELINE.json config #1 in shape: a MIMIC-mortality-demo-like table -- 10 000 rows x 20 mixed columns (6 real,
    4 positive, 3 count, 4 categorical, 3 ordinal, ~8 % missing), a binary outcome that depends on the features --
    through the whole supervised ``fit`` schedule with the reference's architectural defaults (latent 32,
    hidden (256, 128), batch 128, dropout 0.1; suave/defaults.py) and shortened phases.  A smoke test of the product
    at the reference's own scale: every phase runs, the ELBO improves, the classifier separates the outcome on
    held-out rows, generation / imputation / encoding return well-formed frames."""

    from suave_amd import supervised as sup
    from suave_amd.model import SUAVE
    from suave_amd.types import Schema

    rng = np.random.default_rng(20)
    n = 10_000
    risk = rng.normal(size=n)
    cols, spec = {}, {}
    for i in range(6):
        cols[f"vital{i}"] = rng.normal(60 + 5 * i + 6 * risk * (i % 2), 10 + i, n); spec[f"vital{i}"] = {"type": "real"}
    for i in range(4):
        cols[f"lab{i}"] = rng.lognormal(0.5 + 0.3 * i + 0.25 * risk, 0.5, n); spec[f"lab{i}"] = {"type": "pos"}
    for i in range(3):
        cols[f"count{i}"] = rng.poisson(np.exp(0.8 + 0.3 * risk)).astype(float); spec[f"count{i}"] = {"type": "count"}
    for i, c in enumerate((2, 3, 5, 7)):
        cols[f"cat{i}"] = rng.integers(0, c, n).astype(float); spec[f"cat{i}"] = {"type": "cat", "n_classes": c}
    for i, c in enumerate((3, 4, 5)):
        cols[f"ord{i}"] = np.clip(np.rint(rng.normal(c / 2 + 0.6 * risk, 1.0, n)), 0, c - 1); spec[f"ord{i}"] = {"type": "ordinal", "n_classes": c}
    df = pd.DataFrame(cols)
    for c in df.columns:
        df.loc[rng.random(n) < 0.08, c] = np.nan
    y = (risk + 0.5 * rng.normal(size=n) > 0.8).astype(int)  # ~25 % positives
    train, test = df.iloc[:8000], df.iloc[8000:]
    model = SUAVE(Schema(spec), n_components=2, random_state=0)  # everything else: the reference's defaults
    assert (model.latent_dim, model.hidden_dims, model.batch_size, model.dropout) == (32, (256, 128), 128, 0.1)
    model.fit(train, y[:8000], warmup_epochs=4, kl_warmup_epochs=2, head_epochs=2, finetune_epochs=3,
              decoder_refine_epochs=2)
    assert len(model._train_history) == 4 and 1 <= len(model._joint_history) <= 3 and len(model._refine_history) >= 1
    assert model._train_history[-1]["reconstruction"] > model._train_history[0]["reconstruction"]
    assert np.isfinite([h["nll"] for h in model._warmup_val_history]).all()
    proba = model.predict_proba(test)
    assert proba.shape == (2000, 2) and np.allclose(proba.sum(1), 1.0, atol=1e-5)
    assert sup.compute_auroc(proba, y[8000:]) > 0.8
    z = model.encode(test)
    assert z.shape == (2000, 32) and np.isfinite(z).all()
    gen = model.sample(3000, seed=1)
    assert list(gen.columns) == list(df.columns) and len(gen) == 3000
    assert (gen["count0"] >= 0).all() and gen["lab0"].min() > -1.0 and int(gen["cat3"].cat.codes.max()) <= 6
    cond = model.sample(500, conditional=True, y=np.ones(500, int), seed=2)
    assert len(cond) == 500
    filled = model.impute(test)
    assert not filled.isna().any().any()
    keep = test.notna()
    assert np.allclose(filled["vital0"][keep["vital0"]], test["vital0"][keep["vital0"]])  # observed cells are kept
    # the inference methods walk the table in row chunks; the chunk size must not show in the results
    import suave_amd.model as sm

    imp = model.impute(test, seed=5)
    pp = model.predict_proba(test.iloc[:600], attr="cat2", L=20, seed=4)
    old = sm._ROW_CHUNK
    sm._ROW_CHUNK = 300
    try:
        assert np.array_equal(model.encode(test), z)
        pd.testing.assert_frame_equal(model.impute(test, seed=5), imp)
    finally:
        sm._ROW_CHUNK = old
    assert pp.shape == (600, 5) and np.allclose(pp.sum(1), 1.0, atol=1e-5)


@pytest.fixture
def restore_arithmetic():
    import ctypes as C

    from suave_amd import lib as slib

    L = slib.load()
    before = C.c_int(0)
    slib.check(L.suave_gemm_arithmetic_get(C.byref(before)), "suave_gemm_arithmetic_get")
    yield L
    slib.check(L.suave_gemm_arithmetic_set(before.value), "suave_gemm_arithmetic_set")


def test_values_beyond_fp16_switch_the_arithmetic_instead_of_returning_nan(restore_arithmetic):
    """The default matrix arithmetic splits every operand into two fp16 planes (range 65504).  A cell whose normalised
    value is beyond that (new data far outside the training range) would come back as NaN where the reference's fp32
    returns numbers: the packer counts such cells and the model switches the process to the three-bf16-plane
    arithmetic (fp32's exponent range), once, with a warning."""

    from suave_amd.model import SUAVE

    df, _ = _frame(600, seed=4)
    model = SUAVE(_schema(), behaviour="unsupervised", latent_dim=8, n_components=3, hidden_dims=(32, 16), batch_size=128,
                  random_state=0).fit(df, epochs=2)
    assert SUAVE._arithmetic() == 4
    far = df.iloc[:50].copy()
    far.loc[far.index[3], "age"] = 5.0e6   # ~6e5 standard deviations
    with pytest.warns(RuntimeWarning, match="three bf16 planes"):
        z = model.encode(far)
    assert SUAVE._arithmetic() == 3
    assert np.isfinite(z).all()
    ref = model.encode(df.iloc[:50])
    keep = np.arange(50) != 3
    assert np.allclose(z[keep], ref[keep], rtol=1e-4, atol=1e-5)  # the other rows: same numbers on the other arithmetic
    imp = model.impute(far)
    assert imp.isna().sum().sum() == 0


def test_nonfinite_epoch_on_the_fp16_planes_reruns_the_schedule_on_the_wide_arithmetic(restore_arithmetic, monkeypatch):
    """fit: non-finite ELBO sums under the default arithmetic abort the schedule, switch the arithmetic and run it
    again from the same initial state and seeds -- the result is bit for bit the fit a caller gets who selected that
    arithmetic up front.  (The overflow itself is simulated: the epoch sums are poisoned while the fp16 planes are on.)"""

    import ctypes as C

    import torch
    from suave_amd import lib as slib
    from suave_amd.model import SUAVE

    df, g = _frame(500, seed=5)
    kw = dict(behaviour="supervised", latent_dim=8, n_components=2, hidden_dims=(32, 16), batch_size=128, random_state=3)
    fitkw = dict(epochs=2, head_epochs=1, finetune_epochs=1, decoder_refine_epochs=1)
    orig = SUAVE._all_reduce
    calls = {"poisoned": 0}

    def poisoned(self, t):
        out = orig(self, t)
        if SUAVE._arithmetic() == 4 and t is self._engine.sums:
            calls["poisoned"] += 1
            return out * float("nan")
        return out

    monkeypatch.setattr(SUAVE, "_all_reduce", poisoned)
    with pytest.warns(RuntimeWarning, match="non-finite ELBO sums: matrix products switched|switched from two fp16"):
        m1 = SUAVE(_schema(), **kw).fit(df, y=g, **fitkw)
    assert calls["poisoned"] == 1 and SUAVE._arithmetic() == 3
    assert all(np.isfinite(list(h.values())).all() for h in m1._train_history)
    monkeypatch.setattr(SUAVE, "_all_reduce", orig)
    m2 = SUAVE(_schema(), **kw).fit(df, y=g, **fitkw)  # arithmetic 3 from the start
    assert torch.equal(m1._engine.params, m2._engine.params)
    for a, b in zip(m1._classifier.parameters(), m2._classifier.parameters()):
        assert torch.equal(a, b)
    slib.check(restore_arithmetic.suave_gemm_arithmetic_set(4), "suave_gemm_arithmetic_set")
    assert SUAVE._arithmetic() == 4


def test_device_head_fit_agrees_with_the_torch_head_fit(monkeypatch):
    """The supervised schedule with the classifier head in the library (head phase: suave_head_step; joint phase:
    suave_elbo_joint_backward) against the same schedule with the torch module computing the head
    (SUAVE_TORCH_HEAD=1: forward, autograd, torch.optim.Adam, suave_elbo_backward(dlatent)) -- same seeds, same
    row-keyed dropout masks, so the two fits differ by fp32 summation order only."""

    import torch
    from suave_amd.model import SUAVE

    df, g = _frame(900, seed=6)
    kw = dict(behaviour="supervised", latent_dim=8, n_components=2, hidden_dims=(32, 16), batch_size=128, random_state=1,
              dropout=0.1, learning_rate=2e-3)
    fitkw = dict(epochs=3, head_epochs=3, finetune_epochs=3, decoder_refine_epochs=1)
    m_dev = SUAVE(_schema(), **kw).fit(df, y=g, **fitkw)
    assert getattr(m_dev, "_dev_head", None) is not None and m_dev._dev_head.t > 0
    monkeypatch.setenv("SUAVE_TORCH_HEAD", "1")
    m_ref = SUAVE(_schema(), **kw).fit(df, y=g, **fitkw)
    assert getattr(m_ref, "_dev_head", None) is None
    assert np.allclose(m_dev._head_history, m_ref._head_history, rtol=1e-4)
    for a, b in zip(m_dev._joint_history, m_ref._joint_history):
        assert abs(a["train_classification_loss"] - b["train_classification_loss"]) < 2e-3 * abs(b["train_classification_loss"])
        assert abs(a["train_elbo_loss"] - b["train_elbo_loss"]) < 2e-3 * abs(b["train_elbo_loss"])
    pa, pb = m_dev.predict_proba(df.iloc[:200]), m_ref.predict_proba(df.iloc[:200])
    assert np.abs(np.asarray(pa) - np.asarray(pb)).max() < 5e-3


@pytest.mark.gpu
def test_the_class_survives_random_schemas_and_settings(monkeypatch):
    """scripts/api_fuzz.py, sixteen random (schema, missingness, behaviour, K, D, hidden, batch, dropout, val_split)
    cases: fit -> finite history; encode; sample (chunking invariance, dtypes, ranges; conditional); impute under the
    three assignment strategies (no NaN left, observed numeric cells kept); impute_mc; predict_proba (label and
    attribute); calibrate; save -> load -> bit-identical encode."""

    import importlib.util
    import os
    import sys

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "api_fuzz.py")
    spec = importlib.util.spec_from_file_location("api_fuzz", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["api_fuzz.py", "16", "4"])
    assert mod.main() == 0
