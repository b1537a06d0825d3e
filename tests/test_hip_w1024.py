"""BASELINE.json config #4 at its full shape on the GPU: 1024 mixed columns (400 real / 200 pos / 100 count / 224
categorical with up to 256 classes / 100 ordinal), latent 128, K = 10, hidden (384, 256); y_full has 15 605
columns (16 952 with the wide columns' padded slots), the encoder input 16 629.

* one 256-row training step (dropout masks injected) against the numpy oracle: bit-exact components, 1e-4 on
  the per-row ELBO terms, the latent and every gradient tensor -- the fused narrow-column kernel, the GEMM path of
  the 112 wide categorical heads and the embedding first layer all at their real sizes;
* the size-independent properties of tests/test_hip_fullsize.py on a 131 072-row table: per-row terms do not
  depend on the batching, the gradient is linear in the rows and reproducible bit for bit, generation does not
  depend on how rows are sharded, sampled cells are in range.
"""

import numpy as np
import pytest

from oracle import sampling_oracle as sorc
from oracle import suave_oracle as orc
from tests.helpers import rel_err, synth_packed, torch_noise, w1024_columns
from tests.test_hip_elbo import TOL, _engine, _grad_check

pytestmark = pytest.mark.gpu

K, D, HIDDEN = 10, 128, (384, 256)
ROWS = 131072


def test_w1024_full_shape_step_matches_oracle():
    import torch

    from suave_amd.engine import PackedTable

    cols = w1024_columns()
    B, drop = 256, 0.1
    om = orc.OracleModel(cols, n_components=K, latent_dim=D, hidden=HIDDEN, dropout=drop)
    assert len(cols) == 1024 and om.Y == 15605
    P = om.init_params(seed=3)
    rng = np.random.default_rng(5)
    for k in P:
        if ".heads." in k:
            P[k] = (0.3 * rng.standard_normal(P[k].shape)).astype(np.float32)
    xnum, codes, mask = synth_packed(cols, B, 0.1, seed=77)
    x_enc, data, masks = orc.dense_from_packed(om, xnum, codes, mask)
    assert x_enc.shape[1] == 16629
    noise = {"exp": rng.exponential(size=(B, K)).astype(np.float32), "eps": rng.standard_normal((B, D)).astype(np.float32),
             "enc_drop": [(rng.random((B, h)) >= drop).astype(np.uint8) for h in HIDDEN],
             "dec_drop": [(rng.random((B, h)) >= drop).astype(np.uint8) for h in HIDDEN]}
    ref = orc.elbo_supervised(om, P, x_enc, data, masks, noise, beta_scale=1.5, tau=1.0, training=True)
    eng = _engine(om, P)
    assert eng.layout.y_rows == 16952  # 15 605 + the wide columns' assignment slots and 16-byte padding
    table = PackedTable(xnum, codes, mask, eng.device)
    res = eng.elbo(table, 0, B, beta_scale=1.5, tau=1.0, grad=True, noise=torch_noise(noise, eng.device),
                   want=("reconstruction", "categorical_kl", "gaussian_kl", "latent", "component"), accumulate_sums=True)
    torch.cuda.synchronize()
    assert np.array_equal(res["component"].cpu().numpy().astype(np.int64), ref["component_indices"])
    for key in ("reconstruction", "categorical_kl", "gaussian_kl", "latent"):
        assert rel_err(res[key].cpu().numpy(), ref[key]) < TOL, key
    sums = eng.sums.cpu().numpy()
    assert abs(sums[0] - float(ref["reconstruction"].sum())) < TOL * abs(float(ref["reconstruction"].sum())) and sums[3] == B
    _grad_check(eng.grad_state(), ref["grads"])
    # decode at the same shape: the GPU's y_full against the oracle's trunk, decisions bit for bit given that y
    n = 200
    z = rng.standard_normal((n, D)).astype(np.float32)
    comp = rng.integers(0, K, n).astype(np.int32)
    per_col = {}
    for c in cols:
        if c.type in ("real", "pos"):
            per_col[c.name] = {"normal": rng.standard_normal(n).astype(np.float32)}
        elif c.type == "count":
            per_col[c.name] = {"normal": rng.standard_normal(n).astype(np.float32), "uniform": rng.random(n).astype(np.float32)}
        else:
            per_col[c.name] = {"exp": rng.exponential(size=(n, c.n_classes)).astype(np.float32)}
    from tests.test_hip_decode import _pack_noise, _unpack

    nn, nu, ne = _pack_noise(om, eng.layout, per_col, n)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    onum, ocodes, y = eng.decode_sample(t(z), component=t(comp), noise_normal=t(nn), noise_uniform=t(nu), noise_exp=t(ne),
                                        want_y=True)
    y = y.cpu().numpy()
    assert y.shape == (n, 15605)
    assert rel_err(y, orc.decoder_trunk_forward(om, P, z)[0]) < TOL
    want = sorc.sample_heads(om, P, y, np.eye(K, dtype=np.float32)[comp], per_col)
    got = _unpack(om, eng.layout, onum.cpu().numpy(), ocodes.cpu().numpy())
    for c in cols:
        if c.type in ("real", "pos"):
            assert rel_err(got[c.name], want[c.name]) < 2e-5, c.name
        else:
            assert np.array_equal(np.rint(got[c.name]).astype(np.int64), want[c.name]), c.name


@pytest.fixture(scope="module")
def w1024():
    import torch  # noqa: F401

    from bench import K_COMP, synth_table_device, w1024_schema
    from suave_amd.engine import Engine, PackedTable
    from suave_amd.plan import ModelLayout
    from suave_amd.types import Schema

    spec, stats = w1024_schema()
    layout = ModelLayout(Schema(spec), K_COMP, D, HIDDEN, 0.1, "supervised", stats)
    eng = Engine(layout, "cuda:0")
    P = layout.init_params(seed=0)
    rng = np.random.default_rng(5)
    for k in P:
        if ".heads." in k or k.startswith("prior."):
            P[k] = (0.3 * rng.standard_normal(P[k].shape)).astype(np.float32)
    eng.load_state(P)
    xnum, codes, mask = synth_table_device(layout, ROWS, eng.device, seed=20201021)
    table = PackedTable.__new__(PackedTable)
    table.xnum, table.codes, table.mask = xnum, codes, mask
    table.n_rows, table.ld_num, table.ld_disc, table.ld_mask = ROWS, xnum.shape[1], codes.shape[1], mask.shape[1]
    return eng, table, layout


def test_w1024_full_table_properties(w1024):
    import torch

    eng, table, layout = w1024

    def terms(chunk):
        outs = {k: [] for k in ("reconstruction", "categorical_kl", "gaussian_kl", "component")}
        for s0 in range(0, ROWS, chunk):
            o = eng.elbo(table, s0, min(chunk, ROWS - s0), beta_scale=1.0, tau=1.0, grad=False, training=True, seed=77,
                         step=3, want=tuple(outs))
            for k in outs:
                outs[k].append(o[k].clone())
        return {k: torch.cat(v) for k, v in outs.items()}

    a, b = terms(32768), terms(24000)  # ragged: last batch 11 072 rows
    assert torch.equal(a["component"], b["component"])
    for k in ("reconstruction", "categorical_kl", "gaussian_kl"):
        assert bool(torch.isfinite(a[k]).all()), k
        assert float((a[k] - b[k]).abs().max() / a[k].abs().max().clamp_min(1e-6)) < 1e-6, k
    assert float(a["categorical_kl"].min()) > -1e-5 and float(a["gaussian_kl"].min()) > -1e-5
    # gradient: linear in the rows (what data parallelism relies on), bit-identical run to run.  Whole and halves run
    # on the same kernels (the tile kernels at 32 768 / 16 384 rows, the small-batch kernels at 8 192 / 4 096): across
    # the two paths the last bit of a pre-activation may differ, and one ReLU that flips moves a weight row's
    # gradient by a row's whole contribution (1 / B), which is not a rounding error.
    for B in (32768, 8192):
        idx = torch.randperm(ROWS, device=eng.device, generator=torch.Generator(eng.device).manual_seed(1))[:B].to(torch.int32)
        kw = dict(beta_scale=0.8, tau=1.0, grad=True, seed=5, step=11, global_rows=B)
        eng.elbo(table, 0, B, row_index=idx, **kw)
        full = eng.grads.clone()
        eng.elbo(table, 0, B, row_index=idx, **kw)
        assert torch.equal(full, eng.grads)
        eng.elbo(table, 0, B // 2, row_index=idx[: B // 2].contiguous(), **kw)
        half = eng.grads.clone()
        eng.elbo(table, 0, B - B // 2, row_index=idx[B // 2:].contiguous(), **kw)
        both = half + eng.grads
        for name, (lo, hi) in layout.group_ranges.items():
            scale = float(full[lo:hi].abs().max().clamp_min(1e-12))
            assert float((both[lo:hi] - full[lo:hi]).abs().max()) <= 1e-5 * scale + 1e-9, (B, name)


def test_w1024_generation_does_not_depend_on_the_sharding(w1024):
    import torch

    eng, table, layout = w1024
    n = 40000
    P = eng.state()
    lg = np.asarray(P["prior.logits"], np.float64)
    p = np.exp(lg - lg.max()); p /= p.sum()
    cdf = torch.from_numpy(np.cumsum(p).astype(np.float32)).to(eng.device)

    def gen(chunk):
        num, code = [], []
        for s0 in range(0, n, chunk):
            nb = min(chunk, n - s0)
            z, k = eng.prior_latents(nb, cdf, seed=21, row_offset=s0)
            onum, ocodes = eng.decode_sample(z, component=k, seed=22, row_offset=s0)
            num.append(onum.clone()); code.append(ocodes.clone())
        return torch.cat(num), torch.cat(code)

    n1, c1 = gen(16384)
    n2, c2 = gen(9000)
    assert torch.equal(c1, c2) and torch.equal(n1, n2)
    assert bool(torch.isfinite(n1).all())
    i_disc = 0
    for name in layout.schema.feature_names:
        sp = layout.schema[name]
        if sp.type in ("cat", "ordinal"):
            col = c1[:, i_disc]
            assert int(col.min()) >= 0 and int(col.max()) < sp.n_classes, name
            i_disc += 1
