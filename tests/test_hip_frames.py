"""Data formats either side of the path: the device packer
(``suave_pack_columns``) against the golden-pinned host writer ``data.pack_frame`` and the device frame assembly
(``suave_unpack_columns`` + ``FrameAssembler``) against ``data.frame_from_packed``, both through the C-ABI."""

import warnings

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def _schema_and_frame(n, seed=3):
    from suave_amd.types import Schema

    rng = np.random.default_rng(seed)
    spec = {
        "r64": {"type": "real"}, "r32": {"type": "real"}, "rint": {"type": "real"}, "robj": {"type": "real"},
        "p": {"type": "pos"}, "c": {"type": "count"}, "c0": {"type": "count"},
        "cat_s": {"type": "cat", "n_classes": 3}, "cat_i": {"type": "cat", "n_classes": 4},
        "cat_w": {"type": "cat", "n_classes": 200},
        "o_f": {"type": "ordinal", "n_classes": 5}, "o_i": {"type": "ordinal", "n_classes": 3},
    }
    df = pd.DataFrame({
        "r64": rng.normal(50, 12, n),
        "r32": rng.normal(0, 1, n).astype(np.float32),
        "rint": rng.integers(-5, 5, n),
        "robj": pd.Series([str(v) if i % 7 else "oops" for i, v in enumerate(rng.normal(0, 3, n))], dtype=object),
        "p": rng.lognormal(1.0, 0.5, n),
        "c": rng.poisson(3, n).astype(float) + 1.0,
        "c0": rng.poisson(1, n),
        "cat_s": rng.choice(["a", "b", "c"], n),
        "cat_i": rng.integers(0, 4, n),
        "cat_w": (np.arange(n) * 7) % 200,  # every class occurs once n >= 200
        "o_f": rng.integers(0, 5, n).astype(float),
        "o_i": rng.integers(0, 3, n),
    })
    for c in ("r64", "p", "c", "o_f"):
        df.loc[rng.random(n) < 0.1, c] = np.nan
    df.loc[rng.random(n) < 0.1, "cat_s"] = None
    df.loc[5, "p"] = -0.9999       # the smallest values log1p accepts
    df.loc[7, "c0"] = 0            # log(0 + offset)
    df.loc[6, "r64"] = np.inf      # non-finite: masked
    return Schema(spec), df


def _engine(schema, stats):
    from suave_amd.engine import Engine
    from suave_amd.plan import ModelLayout

    return Engine(ModelLayout(schema, 3, 4, (16,), 0.0, "supervised", stats), "cuda:0")


@pytest.mark.parametrize("n,chunk", [(1, 1 << 22), (777, 1 << 22), (5000, 1024)])
def test_device_packer_matches_the_host_writer(n, chunk):
    from suave_amd import data as du

    schema, df = _schema_and_frame(max(n, 8))
    df = df.iloc[:n].reset_index(drop=True)
    stats = du.fit_norm_stats(_schema_and_frame(4000)[1], schema)
    eng = _engine(schema, stats)
    xh, ch, mh = du.pack_frame(df, schema, stats)
    xd, cd, md = du.pack_frame_device(df, schema, stats, eng, chunk_rows=chunk)
    assert np.array_equal(md.cpu().numpy(), mh)
    assert np.array_equal(cd.cpu().numpy(), ch)
    x = xd.cpu().numpy()
    assert x.shape == xh.shape and x.dtype == np.float32
    # float64 arithmetic on both sides, one rounding to float32; log1p / log may differ in the last bit
    assert np.all(np.abs(x - xh) <= 2.0e-7 * np.maximum(1.0, np.abs(xh)))
    numeric = [c for c in schema.feature_names if schema[c].type in ("real", "pos", "count")]
    exact = [i for i, c in enumerate(numeric) if schema[c].type == "real"]
    assert np.array_equal(x[:, exact], xh[:, exact])  # (x - mean) / std: bit for bit


def test_device_packer_rejects_what_the_reference_rejects():
    from suave_amd import data as du

    schema, df = _schema_and_frame(300)
    stats = du.fit_norm_stats(df, schema)
    eng = _engine(schema, stats)
    bad = df.copy()
    bad.loc[3, "p"] = -2.0
    with pytest.raises(ValueError, match="must be >= -1"):
        du.pack_frame_device(bad, schema, stats, eng)
    bad = df.copy()
    bad.loc[3, "c0"] = -1
    with pytest.raises(ValueError, match="non-negative"):
        du.pack_frame_device(bad, schema, stats, eng)
    bad = df.copy()
    bad.loc[3, "o_i"] = 7
    bad.loc[4, "o_f"] = -1.0
    with pytest.warns(UserWarning, match="ordinal values outside"):
        xd, cd, md = du.pack_frame_device(bad, schema, stats, eng)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xh, ch, mh = du.pack_frame(bad, schema, stats)
    assert np.array_equal(md.cpu().numpy(), mh) and np.array_equal(cd.cpu().numpy(), ch)


@pytest.mark.parametrize("n,chunk", [(1, 64), (1000, 1000), (4097, 512)])
def test_device_frame_assembly_matches_the_host_path(n, chunk):
    import torch
    from suave_amd import data as du

    schema, df = _schema_and_frame(500)
    stats = du.fit_norm_stats(df, schema)
    eng = _engine(schema, stats)
    rng = np.random.default_rng(11)
    names = list(schema.feature_names)
    n_num = sum(schema[c].type in ("real", "pos", "count") for c in names)
    disc = [c for c in names if schema[c].type in ("cat", "ordinal")]
    out_num = (rng.normal(2.0, 3.0, (n, n_num)) * rng.choice([1.0, 0.5, 2.5], (n, n_num))).astype(np.float32)
    out_codes = np.stack([rng.integers(0, schema[c].n_classes, n) for c in disc], axis=1).astype(np.int16)
    ref = du.frame_from_packed(out_num, out_codes, schema, stats)
    asm = du.FrameAssembler(schema, stats, eng, n, chunk)
    on, oc = torch.from_numpy(out_num).to(eng.device), torch.from_numpy(out_codes).to(eng.device)
    for s0 in range(0, n, chunk):
        asm.add(s0, on[s0:s0 + chunk].contiguous(), oc[s0:s0 + chunk].contiguous())
    got = asm.frame()
    assert list(got.columns) == list(ref.columns)
    for c in names:
        assert got[c].dtype == ref[c].dtype, c
    pd.testing.assert_frame_equal(got, ref)


def test_device_packer_on_the_reference_tensorise_fixture():
    """Row f1 against REFERENCE data, not product code: the raw frame of tests/golden/tensorise_mixed8.npz through
    ``suave_pack_columns`` must give the encoder inputs the reference's standardize -> _apply_training_normalization ->
    _prepare_training_tensors produced (data.py:133-266, model.py:3116-3213, 2784-2922)."""

    import json

    from oracle import suave_oracle as orc
    from suave_amd import data as du
    from suave_amd.types import Schema
    from tests.helpers import load_golden

    fx = load_golden("tensorise_mixed8")
    schema = Schema(json.loads(str(fx["schema"])))
    X = pd.DataFrame({k[len("raw/"):]: fx[k] for k in fx if k.startswith("raw/")})[list(schema.feature_names)]
    stats = json.loads(str(fx["stats"]))  # the reference's own statistics
    eng = _engine(schema, stats)
    for chunk in (1 << 22, 16):
        xd, cd, md = du.pack_frame_device(X, schema, stats, eng, chunk_rows=chunk)
        cols = [orc.Column(n, schema[n].type, schema[n].n_classes, schema[n].y_dim, stats[n]) for n in schema.feature_names]
        om = orc.OracleModel(cols, n_components=2, latent_dim=4, hidden=(8,))
        x_enc, _, _ = orc.dense_from_packed(om, xd.cpu().numpy(), cd.cpu().numpy(), md.cpu().numpy())
        np.testing.assert_allclose(x_enc, fx["enc_inputs"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("pre", ["supervised/sample/", "unsupervised/sample/", "cond_sample/"])
def test_device_frame_assembly_on_the_reference_sample_frames(pre):
    """Row f2 against REFERENCE data: the frames SUAVE.sample returned in tests/golden/generation_mixed7.npz (values of
    sampling.decoder_outputs_to_frame, sampling.py:138-179), written back into the decode kernels' output layout and
    sent through ``suave_unpack_columns`` + FrameAssembler, must come back as the same frames: values, dtypes,
    category lists and orderedness."""

    import json

    import torch
    from suave_amd import data as du
    from suave_amd.types import Schema
    from tests.helpers import load_golden

    fx = load_golden("generation_mixed7")
    cols = json.loads(str(fx["columns"]))
    spec = {c["name"]: {"type": c["type"], **({"n_classes": c["n_classes"]} if c.get("n_classes") else {})} for c in cols}
    schema = Schema(spec)
    stats = json.loads(str(fx[("supervised/" if pre == "cond_sample/" else pre.split("/")[0] + "/") + "stats"]))
    eng = _engine(schema, stats)
    names = list(schema.feature_names)
    num = [c for c in names if schema[c].type in ("real", "pos", "count")]
    disc = [c for c in names if c not in num]
    n = len(fx[pre + f"frame/{names[0]}"])
    out_num = np.stack([fx[pre + f"frame/{c}"].astype(np.float32) for c in num], axis=1)
    out_codes = np.stack([fx[pre + f"frame/{c}"].astype(np.int16) for c in disc], axis=1)
    for chunk in (n, 17):
        asm = du.FrameAssembler(schema, stats, eng, n, chunk)
        on, oc = torch.from_numpy(out_num).to(eng.device), torch.from_numpy(out_codes).to(eng.device)
        for s0 in range(0, n, chunk):
            asm.add(s0, on[s0:s0 + chunk].contiguous(), oc[s0:s0 + chunk].contiguous())
        got = asm.frame()
        assert list(got.columns) == names
        for c in names:
            v, ref = got[c], fx[pre + f"frame/{c}"]
            assert str(v.dtype) == str(fx[pre + f"dtype/{c}"]), c
            if c in num:
                assert np.array_equal(v.to_numpy(), ref), c
            else:
                assert np.array_equal(np.asarray(v.cat.codes, np.int64), ref), c
                assert [str(x) for x in v.cat.categories] == list(fx[pre + f"categories/{c}"]), c
                assert bool(v.cat.ordered) == bool(fx[pre + f"ordered/{c}"]), c
