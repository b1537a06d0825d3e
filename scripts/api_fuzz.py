"""Fuzz of the SUAVE-compatible class end to end on random schemas and hyper-parameters: fit (both behaviours), encode,
sample (conditional too), impute (three strategies), impute_mc, predict_proba / predict, calibrate, save -> load.
Checks: finite histories, shapes / dtypes / value ranges, observed cells preserved, chunking invariance of sample,
bit-identical encode after a save / load round trip.  usage: python scripts/api_fuzz.py [n_cases] [seed]"""
import os, sys, tempfile, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
from suave_amd.model import SUAVE
from suave_amd.types import Schema


def random_frame(rng, n, n_cols):
    spec, cols = {}, {}
    g = rng.integers(0, 3, n)
    for j in range(n_cols):
        t = str(rng.choice(["real", "pos", "count", "cat", "ordinal"]))
        name = f"f{j}"
        if t == "real":
            v = rng.normal(rng.normal() * 20 + g, 0.5 + rng.random() * 10, n); spec[name] = {"type": t}
            edge = rng.random()
            if edge < 0.08:
                v[:] = 3.25  # constant column: the normaliser's std floor
            elif edge < 0.16:
                v *= 1e6  # large units
            elif edge < 0.24:
                v[rng.integers(0, n)] = 1e9  # one wild outlier
        elif t == "pos":
            v = np.exp(rng.normal(0.3 * g, 0.5, n)); spec[name] = {"type": t}
        elif t == "count":
            v = rng.poisson(1 + 2 * g + rng.integers(0, 30), n).astype(float); spec[name] = {"type": t}
        else:
            c = int(rng.integers(2, 24 if t == "cat" else 9))
            v = rng.integers(0, c, n).astype(float); spec[name] = {"type": t, "n_classes": c}
        v = v.astype(np.float64)
        v[rng.random(n) < rng.choice([0.0, 0.1, 0.4])] = np.nan
        if t in ("cat", "ordinal"):
            # every declared level observed: with fewer, the reference (suave/sampling.py:121-135) pads the category
            # list with integer placeholders that can collide with observed levels and pandas raises -- kept as is
            v[:c] = np.arange(c)
        cols[name] = v
    return Schema(spec), pd.DataFrame(cols), g


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    warnings.simplefilter("ignore")
    bad = 0
    for case in range(n_cases):
        n = int(rng.choice([40, 257, 1000, 3000]))
        n_cols = int(rng.integers(1, 25))
        schema, df, g = random_frame(rng, n, n_cols)
        behaviour = str(rng.choice(["supervised", "unsupervised"]))
        kw = dict(latent_dim=int(rng.choice([2, 8, 32])), n_components=int(rng.integers(1, 8)),
                  hidden_dims=tuple(int(rng.choice([16, 32, 64, 256])) for _ in range(int(rng.integers(1, 3)))),
                  batch_size=int(rng.choice([32, 128, 256, 4096])), dropout=float(rng.choice([0.0, 0.1])),
                  val_split=float(rng.choice([0.0, 0.2])), random_state=int(rng.integers(100)))
        if n < 100:
            kw["val_split"] = 0.0  # a 32-row training split would again miss levels (see random_frame)
        desc = f"case {case}: {behaviour[:5]} n={n} cols={n_cols} {kw}"
        print(desc + " ...", flush=True)
        try:
            m = SUAVE(schema, behaviour=behaviour, **kw)
            if behaviour == "supervised":
                m.fit(df, y=g, epochs=2, head_epochs=1, finetune_epochs=1, decoder_refine_epochs=1)
            else:
                m.fit(df, epochs=2)
            hist = m._train_history
            assert len(hist) == 2 and all(np.isfinite(list(h.values())).all() for h in hist), hist
            D = kw["latent_dim"]
            z = m.encode(df)
            assert z.shape == (n, D) and np.isfinite(z).all()
            s1 = m.sample(300, batch_size=300, seed=5)
            s2 = m.sample(300, batch_size=77, seed=5)
            pd.testing.assert_frame_equal(s1, s2)
            for c in schema.feature_names:
                t = schema[c].type
                if t in ("real", "pos"):
                    assert s1[c].dtype == np.float32 and np.isfinite(s1[c]).all(), c
                    if t == "pos":
                        assert s1[c].min() > -1.0, c
                elif t == "count":
                    assert s1[c].dtype == np.int64 and (s1[c] >= 0).all(), c
                else:
                    assert isinstance(s1[c].dtype, pd.CategoricalDtype) and s1[c].cat.codes.min() >= 0, c
            if behaviour == "supervised":
                sc = m.sample(50, conditional=True, y=rng.choice(np.unique(g), 50), seed=2)
                assert len(sc) == 50
                p = np.asarray(m.predict_proba(df.iloc[:64]))
                assert p.shape[0] == min(64, n) and np.allclose(p.sum(1), 1.0, atol=1e-5)
                m.calibrate(df.iloc[: min(n, 200)], g[: min(n, 200)])
            sub = df.iloc[: min(n, 150)]
            for strat in ("hard", "soft", "sample"):
                imp = m.impute(sub, assignment_strategy=strat)
                assert imp.isna().sum().sum() == 0
                for c in schema.feature_names:
                    if schema[c].type in ("real", "pos", "count"):
                        obs = sub[c].notna().to_numpy()
                        assert np.allclose(imp[c].to_numpy(dtype=np.float64)[obs], sub[c].to_numpy()[obs], rtol=1e-5, atol=1e-6), (strat, c)
            mc = m.impute_mc(sub, L=4, seed=1)
            assert mc.isna().sum().sum() == 0
            disc = [c for c in schema.feature_names if schema[c].type in ("cat", "ordinal")]
            if disc:
                pp = m.predict_proba(sub, attr=disc[0], L=8)
                assert pp.shape == (len(sub), schema[disc[0]].n_classes) and bool(torch.isfinite(pp).all())
            with tempfile.TemporaryDirectory() as d:
                path = os.path.join(d, "m.pt")
                m.save(path)
                back = SUAVE.load(path, device="cuda:0")
                assert np.array_equal(back.encode(df), z)
            print(desc + " ok", flush=True)
        except Exception as exc:  # noqa: BLE001
            import traceback
            bad += 1
            print(desc + " FAILED: " + repr(exc)[:300], flush=True)
            traceback.print_exc()
    print("failures:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
