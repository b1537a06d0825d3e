"""Size-independent properties of the HIP path at BASELINE.json's full single-GPU size (config #2: 1M rows x 64
mixed columns, K=10, latent 32), where the numpy oracle would take minutes: chunking invariance of the per-row
ELBO terms, linearity of the gradient in the rows, determinism, row-sharding invariance of generation, value
ranges of the sampled cells, and empty inputs."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROWS = 1_000_000


@pytest.fixture(scope="module")
def s64():
    import torch
    from bench import HIDDEN, K_COMP, LATENT, s64_schema, synth_table_device
    from suave_amd.engine import Engine, PackedTable
    from suave_amd.plan import ModelLayout
    from suave_amd.types import Schema

    spec, stats = s64_schema()
    layout = ModelLayout(Schema(spec), K_COMP, LATENT, HIDDEN, 0.1, "supervised", stats)
    eng = Engine(layout, "cuda:0")
    P = layout.init_params(seed=0)
    rng = np.random.default_rng(5)
    for k in P:  # O(1) head weights and a non-trivial prior so that every term is exercised
        if ".heads." in k or k.startswith("prior."):
            P[k] = (0.3 * rng.standard_normal(P[k].shape)).astype(np.float32)
    eng.load_state(P)
    xnum, codes, mask = synth_table_device(layout, ROWS, eng.device, seed=20201021)
    table = PackedTable.__new__(PackedTable)
    table.xnum, table.codes, table.mask = xnum, codes, mask
    table.n_rows, table.ld_num, table.ld_disc, table.ld_mask = ROWS, xnum.shape[1], codes.shape[1], mask.shape[1]
    return eng, table, layout


def _terms(eng, table, chunk, training):
    import torch

    outs = {k: [] for k in ("reconstruction", "categorical_kl", "gaussian_kl", "component")}
    for s0 in range(0, ROWS, chunk):
        nb = min(chunk, ROWS - s0)
        o = eng.elbo(table, s0, nb, beta_scale=1.0, tau=1.0, grad=False, training=training, seed=77, step=3,
                     want=tuple(outs))
        for k in outs:
            outs[k].append(o[k].clone())
    return {k: torch.cat(v) for k, v in outs.items()}


@pytest.mark.parametrize("training", [False, True])
def test_full_table_elbo_terms_do_not_depend_on_the_batching(s64, training):
    """Rows are independent terms (model.py:1617) and the in-kernel noise / dropout is keyed by the table row, so
    the per-row ELBO terms of all 1M rows are the same whether the table is walked in 65 536-row batches or in
    ragged 40 000-row batches (last batch of 1M % 40 000 = 0 rows -> use 48 000: last batch 40 000)."""

    import torch

    eng, table, _ = s64
    a = _terms(eng, table, 65536, training)
    b = _terms(eng, table, 48000, training)
    assert torch.equal(a["component"], b["component"])  # bit-exact assignments
    for k in ("reconstruction", "categorical_kl", "gaussian_kl"):
        assert bool(torch.isfinite(a[k]).all()), k
        err = float((a[k] - b[k]).abs().max() / a[k].abs().max().clamp_min(1e-6))
        assert err < 1e-6, (k, err)  # tolerance 1e-6 relative; observed 0 (k-ordered fp32 FMA chains per row)
    # sanity of the values themselves: KL terms are non-negative, mean reconstruction is a finite negative number
    assert float(a["categorical_kl"].min()) > -1e-5 and float(a["gaussian_kl"].min()) > -1e-5
    assert float(a["reconstruction"].double().mean()) < 0.0


def test_gradient_is_linear_in_the_rows_and_deterministic(s64):
    """loss = -mean_b(elbo_b): the gradient of a 65 536-row batch equals the sum of the gradients of its two
    halves computed with global_rows = 65 536 (this is exactly what the data-parallel all-reduce relies on), to
    1e-5 of each tensor's largest entry; two runs of the same batch give bit-identical gradients."""

    import torch

    eng, table, layout = s64
    B = 65536
    idx = torch.randperm(ROWS, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(1),
                         dtype=torch.int64)[:B].to(torch.int32)
    kw = dict(beta_scale=0.8, tau=1.0, grad=True, global_rows=B, seed=9, step=2)
    eng.elbo(table, 0, B, row_index=idx, **kw)
    g_full = eng.grads.clone()
    eng.elbo(table, 0, B, row_index=idx, **kw)
    assert torch.equal(g_full, eng.grads)
    eng.elbo(table, 0, B // 2, row_index=idx[: B // 2].contiguous(), **kw)
    g_half = eng.grads.clone()
    eng.elbo(table, 0, B - B // 2, row_index=idx[B // 2:].contiguous(), **kw)
    g_sum = g_half + eng.grads
    assert bool(torch.isfinite(g_full).all())
    for name, (b, e) in layout.group_ranges.items():
        ref = g_full[b:e]
        err = float((g_sum[b:e] - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
        assert err < 1e-5, (name, err)


def test_generation_is_invariant_to_row_sharding_and_in_range(s64):
    """sample(): 1M rows decoded in one shard layout equal the same rows decoded in another (counter = global
    row index, no collective); codes lie in [0, C), counts are non-negative integers, numerics finite."""

    import torch
    from oracle import sampling_oracle as sorc

    eng, _, layout = s64
    P = eng.state()
    cdf = torch.from_numpy(sorc.prior_cdf(P["prior.logits"])).to(eng.device)

    def run(chunk):
        num, code, comp = [], [], []
        for s0 in range(0, ROWS, chunk):
            nb = min(chunk, ROWS - s0)
            z, k = eng.prior_latents(nb, cdf, seed=21, row_offset=s0)
            onum, ocodes = eng.decode_sample(z, component=k, seed=22, row_offset=s0)
            num.append(onum.clone()); code.append(ocodes.clone()); comp.append(k.clone())
        return torch.cat(num), torch.cat(code), torch.cat(comp)

    n1, c1, k1 = run(1 << 18)
    n2, c2, k2 = run(300_000)
    assert torch.equal(k1, k2) and torch.equal(c1, c2) and torch.equal(n1, n2)
    assert bool(torch.isfinite(n1).all())
    i_num = i_disc = 0
    for name in layout.schema.feature_names:
        sp = layout.schema[name]
        if sp.type in ("cat", "ordinal"):
            col = c1[:, i_disc]
            assert int(col.min()) >= 0 and int(col.max()) < int(sp.n_classes), name
            assert int(torch.unique(col).numel()) == int(sp.n_classes), name  # every level is reachable in 1M draws
            i_disc += 1
        else:
            col = n1[:, i_num]
            if sp.type == "count":
                assert float(col.min()) >= 0.0 and bool((col == col.round()).all()), name
            if sp.type == "pos":
                assert float(col.min()) > -1.0, name  # exp(.) - 1
            i_num += 1
    # the prior's component frequencies over 1M draws match softmax(prior_logits) (inverse-CDF draw)
    p = np.exp(P["prior.logits"] - P["prior.logits"].max()); p /= p.sum()
    freq = torch.bincount(k1.long(), minlength=len(p)).cpu().numpy() / ROWS
    assert np.abs(freq - p).max() < 5e-3


def test_conditional_generation_full_size_properties(s64):
    """sample(conditional=True) at 1M rows (suave_conditional_latents -> the fused decode): the rows do not depend on
    the chunking; every draw of label c comes from the cached posteriors of class c's training rows -- the mean of z
    over a class's draws is the mixture mean of that class (sum_rows sum_k p_k mu_k / n_c) within 6 standard errors --
    and the component frequencies per class are the class's mean responsibilities."""

    import torch

    eng, _, layout = s64
    K, D = layout.n_components, layout.latent_dim
    g = torch.Generator(device=eng.device).manual_seed(31)
    n_train, n_cls = 50_000, 3
    y_train = torch.randint(0, n_cls, (n_train,), device=eng.device, generator=g)
    probs = torch.softmax(2.0 * torch.randn((n_train, K), device=eng.device, generator=g), dim=1)
    mu = torch.randn((n_train, K, D), device=eng.device, generator=g) + y_train.view(-1, 1, 1).float()
    logvar = 0.3 * torch.randn((n_train, K, D), device=eng.device, generator=g) - 1.0
    order = torch.argsort(y_train, stable=True)
    class_rows = order.to(torch.int64)
    counts = torch.bincount(y_train, minlength=n_cls)
    class_offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=eng.device), counts.cumsum(0)]).to(torch.int64)
    labels = torch.randint(0, n_cls, (ROWS,), device=eng.device, generator=g).to(torch.int32)

    def run(chunk):
        zs, ks, nums, codes = [], [], [], []
        for s0 in range(0, ROWS, chunk):
            nb = min(chunk, ROWS - s0)
            z, k = eng.conditional_latents(labels[s0:s0 + nb].contiguous(), class_offsets, class_rows, probs, mu, logvar,
                                           seed=41, row_offset=s0)
            onum, ocodes = eng.decode_sample(z, component=k, seed=42, row_offset=s0)
            zs.append(z.clone()); ks.append(k.clone()); nums.append(onum.clone()); codes.append(ocodes.clone())
        return torch.cat(zs), torch.cat(ks), torch.cat(nums), torch.cat(codes)

    z1, k1, n1, c1 = run(1 << 18)
    z2, k2, n2, c2 = run(333_333)
    assert torch.equal(k1, k2) and torch.equal(z1, z2) and torch.equal(c1, c2) and torch.equal(n1, n2)
    assert bool(torch.isfinite(z1).all()) and bool(torch.isfinite(n1).all())
    for c in range(n_cls):
        rows_c = y_train == c
        sel = labels == c
        m = int(sel.sum())
        want_mean = (probs[rows_c].unsqueeze(-1) * mu[rows_c]).sum(dim=1).mean(dim=0)
        got_mean = z1[sel].mean(dim=0)
        sd = z1[sel].std(dim=0)
        assert bool(((got_mean - want_mean).abs() <= 6.0 * sd / m ** 0.5 + 1e-3).all()), c
        freq = torch.bincount(k1[sel].long(), minlength=K).float() / m
        assert float((freq - probs[rows_c].mean(dim=0)).abs().max()) < 6.0 * (0.25 / m) ** 0.5 + 1e-3, c


def test_mc32_posterior_predictive_full_size_properties(s64):
    """impute_mc's device path at the bench size (32 768 table rows x 32 draws = 1M decoded rows): encode ->
    suave_posterior_draws (row-major, keyed by (table row, draw)) -> the fused decode.  The decoded rows do not depend
    on how the table rows are chunked; the mean of a row's 32 z draws is its moment-matched posterior mean within 6
    standard errors (pooled over rows); the component frequencies follow the responsibilities."""

    import torch

    eng, table, layout = s64
    K, D, L, n = layout.n_components, layout.latent_dim, 32, 32768

    def run(chunk):
        zs, ks, nums, codes = [], [], [], []
        for s0 in range(0, n, chunk):
            nb = min(chunk, n - s0)
            enc = eng.encode(table, s0, nb)
            z, k = eng.posterior_draws(enc["enc_out"], L, seed=51, row_offset=s0, row_major=True)
            onum, ocodes = eng.decode_sample(z, component=k, seed=52, row_offset=s0 * L)
            zs.append(z.clone()); ks.append(k.clone()); nums.append(onum.clone()); codes.append(ocodes.clone())
        return torch.cat(zs), torch.cat(ks), torch.cat(nums), torch.cat(codes)

    z1, k1, n1, c1 = run(n)
    z2, k2, n2, c2 = run(5000)
    assert torch.equal(k1, k2) and torch.equal(z1, z2) and torch.equal(c1, c2) and torch.equal(n1, n2)
    assert bool(torch.isfinite(n1).all())
    enc = eng.encode(table, 0, n)
    zbar = z1.view(n, L, D).mean(dim=1)
    # per-coordinate standard error of a row's mean from the moment-matched variance of the mixture
    se = (enc["logvar"].exp() / L).sqrt()
    dev = ((zbar - enc["mean"]) / se)
    assert abs(float(dev.mean())) < 0.02 and abs(float(dev.std()) - 1.0) < 0.05  # standardised deviations ~ N(0, 1)
    freq = torch.zeros((n, K), device=eng.device).scatter_add_(1, k1.view(n, L).long(), torch.ones((n, L), device=eng.device)) / L
    assert float((freq.mean(dim=0) - enc["probs"].mean(dim=0)).abs().max()) < 2e-3


def test_empty_inputs_are_accepted(s64):
    """Zero-row batches: every entry point returns success and launches no model kernel (reference: batch == 0
    guards, model.py:3361-3362, 3398-3399, 4238); the backward's gradient of nothing is zero."""

    import torch

    eng, table, layout = s64
    eng.grads.fill_(1.0)
    s0 = eng.sums.clone()
    eng.elbo(table, 0, 0, beta_scale=1.0, grad=True, global_rows=1, accumulate_sums=True)
    assert not eng.grads.any() and torch.equal(s0, eng.sums)
    ev = torch.cuda.Event()
    ev.record()
    eng.grads.fill_(1.0)
    eng.elbo(table, 0, 0, beta_scale=1.0, grad=True, global_rows=1, ready_event=ev)  # an empty shard still records
    ev.synchronize()
    assert not eng.grads.any()
    assert eng.encode(table, 0, 0)["enc_out"].shape[0] == 0
    z = torch.empty((0, layout.latent_dim), device=eng.device)
    onum, ocodes = eng.decode_sample(z, component=torch.empty((0,), dtype=torch.int32, device=eng.device))
    assert onum.shape[0] == 0 and ocodes.shape[0] == 0
    K, D = layout.n_components, layout.latent_dim
    z0, k0 = eng.posterior_draws(torch.empty((0, K + 2 * K * D), device=eng.device), 4)
    assert z0.shape[0] == 0 and k0.shape[0] == 0
    from suave_amd.lib import SuaveHipError
    with pytest.raises(SuaveHipError, match="positive integer"):  # model.py:3313-3314
        eng.posterior_draws(torch.empty((0, K + 2 * K * D), device=eng.device), 0)
    assert eng.decode_params(z, 0, component=torch.empty((0,), dtype=torch.int32, device=eng.device)).shape[0] == 0


def test_overlap_entry_point_matches_and_its_event_marks_a_final_decoder_bucket(s64):
    """suave_elbo_backward_overlap (include/suave_hip.h): same gradient bits as suave_elbo_backward, and what a
    communication stream reads from the decoder + heads + prior range once the event has fired -- while the
    encoder backward is still running on the compute stream -- is already that range's final value."""

    import torch

    eng, table, layout = s64
    B = 65536
    idx = torch.randperm(ROWS, device=eng.device, generator=torch.Generator(eng.device).manual_seed(3))[:B].to(torch.int32)
    dl = (0.01 * torch.randn((B, layout.latent_dim), device=eng.device, generator=torch.Generator(eng.device).manual_seed(4)))
    eng.elbo(table, 0, B, row_index=idx, beta_scale=0.7, grad=True, seed=5, step=9, dlatent=dl, global_rows=2 * B)
    ref = eng.grads.clone()
    lo, hi = layout.group_ranges["decoder"][0], layout.group_ranges["prior"][1]
    side = torch.cuda.Stream(eng.device)
    ev = torch.cuda.Event()
    ev.record()
    early = torch.empty(hi - lo, device=eng.device)
    for _ in range(3):
        eng.grads.zero_()  # (alignment padding between tensors is never written: keep it equal to ref's)
        eng.elbo(table, 0, B, row_index=idx, beta_scale=0.7, grad=True, seed=5, step=9, dlatent=dl, global_rows=2 * B,
                 ready_event=ev)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            early.copy_(eng.grads[lo:hi])
        torch.cuda.synchronize()
        assert torch.equal(eng.grads, ref)
        assert torch.equal(early, ref[lo:hi])


def test_overlap_entry_point_in_the_unsupervised_branch():
    """The same for behaviour="unsupervised": the decoder + heads + PriorMean range is final at the event (the PriorMean
    gradient comes out of the mixture finalisation, before the encoder backward), and the gradient bits are those of
    suave_elbo_backward."""

    import torch
    from suave_amd.engine import Engine, PackedTable
    from tests.helpers import layout_from_columns, s64_columns, synth_packed

    cols = s64_columns(64, real_variant="scaled")
    layout = layout_from_columns(cols, K=4, D=32, hidden=(256, 128), dropout=0.1, behaviour="unsupervised")
    eng = Engine(layout, "cuda:0")
    from oracle import suave_oracle as orc
    eng.load_state(orc.OracleModel(cols, n_components=4, latent_dim=32, hidden=(256, 128), dropout=0.1,
                                   behaviour="unsupervised").init_params(seed=2))
    n, B = 40000, 16384
    xnum, codes, mask = synth_packed(cols, n, 0.1, seed=9)
    table = PackedTable(xnum, codes, mask, eng.device)
    idx = torch.randperm(n, device=eng.device, generator=torch.Generator(eng.device).manual_seed(3))[:B].to(torch.int32)
    eng.grads.zero_()
    eng.elbo(table, 0, B, row_index=idx, beta_scale=0.7, tau=0.9, grad=True, seed=5, step=9, global_rows=2 * B)
    ref = eng.grads.clone()
    assert float(ref.abs().max()) > 0
    lo, hi = layout.group_ranges["decoder"][0], layout.group_ranges["prior"][1]
    side = torch.cuda.Stream(eng.device)
    ev = torch.cuda.Event()
    ev.record()
    early = torch.empty(hi - lo, device=eng.device)
    for _ in range(3):
        eng.grads.zero_()
        eng.elbo(table, 0, B, row_index=idx, beta_scale=0.7, tau=0.9, grad=True, seed=5, step=9, global_rows=2 * B,
                 ready_event=ev)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            early.copy_(eng.grads[lo:hi])
        torch.cuda.synchronize()
        assert torch.equal(eng.grads, ref)
        assert torch.equal(early, ref[lo:hi])


def test_posterior_draws_lane_group_kernel_equals_the_scalar_kernel(s64):
    """suave_posterior_draws has two kernels with the same generator keys and arithmetic: eight lanes per row
    (D % 4 == 0, K <= 32, 16-byte aligned z) and one thread per draw.  Same bits out of both (the scalar one is
    reached here by handing the entry point a z buffer that is only 4-byte aligned), ragged row count included."""

    import ctypes as C

    import torch

    from suave_amd import lib as slib

    eng, table, layout = s64
    B, L, D = 1003, 7, layout.latent_dim
    enc = eng.encode(table, 5000, B)["enc_out"]
    z_a, k_a = eng.posterior_draws(enc, L, seed=21, row_offset=12345)
    buf = torch.empty(L * B * D + 1, dtype=torch.float32, device=eng.device)
    z_b = buf[1:]
    k_b = torch.empty(L * B, dtype=torch.int32, device=eng.device)
    assert z_b.data_ptr() % 16 == 4
    rc = eng.lib.suave_posterior_draws(eng._plan, enc.data_ptr(), B, L, None, None, 21, 12345, eng._stream(), z_b.data_ptr(),
                                       k_b.data_ptr())
    slib.check(rc, "suave_posterior_draws")
    torch.cuda.synchronize()
    assert torch.equal(k_a, k_b)
    assert torch.equal(z_a.reshape(-1), z_b)
    assert int(k_a.min()) >= 0 and int(k_a.max()) < layout.n_components


def test_posterior_draws_by_row_is_the_transposed_draw_major_output(s64):
    """suave_posterior_draws_by_row: same keys and bits as suave_posterior_draws, output row b * L + l."""

    import torch

    eng, table, layout = s64
    K, D, L, n = layout.n_components, layout.latent_dim, 7, 3001
    enc = eng.encode(table, 500, n)["enc_out"]
    z0, k0 = eng.posterior_draws(enc, L, seed=13, row_offset=500)
    z1, k1 = eng.posterior_draws(enc, L, seed=13, row_offset=500, row_major=True)
    assert torch.equal(z1, z0.view(L, n, D).transpose(0, 1).reshape(n * L, D))
    assert torch.equal(k1, k0.view(L, n).transpose(0, 1).reshape(-1))
    g = torch.Generator(device=eng.device).manual_seed(2)
    e = torch.empty((L, n, K), device=eng.device).exponential_(generator=g)
    eps = torch.randn((L, n, D), device=eng.device, generator=g)
    z2, k2 = eng.posterior_draws(enc, L, e=e, eps=eps)
    z3, k3 = eng.posterior_draws(enc, L, e=e, eps=eps, row_major=True)  # injected draws stay draw-major
    assert torch.equal(z3, z2.view(L, n, D).transpose(0, 1).reshape(n * L, D))
    assert torch.equal(k3, k2.view(L, n).transpose(0, 1).reshape(-1))
