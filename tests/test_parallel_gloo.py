"""Data-parallel host logic on CPU with gloo, world_size 2.

The product's compute needs a GPU, so each rank's shard gradient comes from the CPU oracle here (test
infrastructure); what is under test is the sharding arithmetic of suave_amd.parallel: with every rank
scaling by 1/global_rows, the SUM all-reduce of shard gradients equals the single-batch gradient, also
for ragged shards, and epoch metric sums agree on all ranks.
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import suave_oracle as orc
from suave_amd import parallel
from tests.helpers import layout_from_columns, s64_columns, synth_packed


def test_shard_rows_cover_and_balance():
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_rows(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gradient_buckets_cover_everything():
    cols = s64_columns(64)
    layout = layout_from_columns(cols, K=10, D=32, hidden=(256, 128))
    assert parallel.gradient_buckets(layout) == [(0, layout.n_params)]
    small = parallel.gradient_buckets(layout, max_bucket_floats=50000)
    covered = np.zeros(layout.n_params, bool)
    for b, e in small:
        assert not covered[b:e].any()
        covered[b:e] = True
    assert covered.all()
    assert small[0][0] == layout.group_ranges["decoder"][0]  # decoder side first: ready first in backward


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cols = s64_columns(12)
    om = orc.OracleModel(cols, n_components=3, latent_dim=4, hidden=(16, 8))
    P = om.init_params(seed=1)
    layout = layout_from_columns(cols, K=3, D=4, hidden=(16, 8))
    xnum, codes, mask = synth_packed(cols, B, 0.2, seed=5)
    rng = np.random.default_rng(6)
    noise = {"exp": rng.exponential(size=(B, 3)).astype(np.float32), "eps": rng.standard_normal((B, 4)).astype(np.float32)}
    b, e = parallel.shard_rows(B, rank, world)
    x_enc, data, masks = orc.dense_from_packed(om, xnum[b:e], codes[b:e], mask[b:e])
    loc = orc.elbo_supervised(om, P, x_enc, data, masks, {k: v[b:e] for k, v in noise.items()}, beta_scale=1.0)
    n_glob = parallel.global_batch_rows(e - b)
    scale = (e - b) / n_glob  # the oracle normalises by the local batch; the HIP entry point takes global_rows
    flat = torch.from_numpy(layout.pack({k: np.asarray(v, np.float32).reshape(layout.slots[k].shape) * scale
                                         for k, v in loc["grads"].items()}))
    parallel.allreduce_gradients(flat, parallel.gradient_buckets(layout, 300))
    sums = torch.tensor([loc["reconstruction"].sum(), loc["categorical_kl"].sum(), loc["gaussian_kl"].sum(), e - b],
                        dtype=torch.float64)
    parallel.allreduce_sums(sums)
    out_q.put((rank, n_glob, flat.numpy(), sums.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [64, 51])
def test_two_rank_gradient_equals_single_batch(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cols = s64_columns(12)
    om = orc.OracleModel(cols, n_components=3, latent_dim=4, hidden=(16, 8))
    P = om.init_params(seed=1)
    layout = layout_from_columns(cols, K=3, D=4, hidden=(16, 8))
    xnum, codes, mask = synth_packed(cols, B, 0.2, seed=5)
    rng = np.random.default_rng(6)
    noise = {"exp": rng.exponential(size=(B, 3)).astype(np.float32), "eps": rng.standard_normal((B, 4)).astype(np.float32)}
    x_enc, data, masks = orc.dense_from_packed(om, xnum, codes, mask)
    full = orc.elbo_supervised(om, P, x_enc, data, masks, noise, beta_scale=1.0)
    ref = layout.pack({k: np.asarray(v, np.float32).reshape(layout.slots[k].shape) for k, v in full["grads"].items()})
    for rank, n_glob, flat, sums in results:
        assert n_glob == B
        assert np.allclose(flat, ref, rtol=1e-4, atol=1e-6)
        assert abs(sums[0] - float(full["reconstruction"].sum())) < 1e-3 and sums[3] == B
    assert np.array_equal(results[0][2], results[1][2])  # replicas stay bit-identical


def _dp_worker(rank, world, port, out_q):
    """DataParallel's host logic (what SUAVE.fit drives) on CPU tensors: sharding, the two-bucket reduction, the
    classifier-head gradient of a row-sharded weighted cross-entropy, broadcast, and the replicated-table check."""

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from suave_amd import supervised as sup

    layout = layout_from_columns(s64_columns(12), K=3, D=4, hidden=(16, 8))
    dp = parallel.DataParallel.from_environment(layout, "cpu")
    assert dp is not None and dp.host_staged and dp.ready is None
    rows = torch.arange(51)
    mine = dp.shard(rows)
    flat = torch.full((layout.n_params,), float(rank + 1))
    dp.reduce_gradients(flat)
    # classifier head: sharded loss_terms / global loss_weight == the whole batch's weighted mean
    torch.manual_seed(0)
    clf = sup.ClassificationHead(4, 3, class_weight=np.array([0.5, 1.0, 2.5], np.float32))
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(51, 4, generator=g)
    y = torch.randint(0, 3, (51,), generator=g)
    loss = clf.loss_terms(clf(lat[mine]), y[mine]) / clf.loss_weight(y)
    loss.backward()
    dp.reduce_module_gradients(clf)
    grad = torch.cat([p.grad.reshape(-1) for p in clf.parameters()])
    with torch.no_grad():
        for p in clf.parameters():
            p.add_(float(rank))  # make the replicas differ, then rank 0's bits must win
    dp.broadcast_module(clf)
    after = torch.cat([p.detach().reshape(-1) for p in clf.parameters()])
    dp.check_replicated(torch.ones(5, 2), torch.arange(10.0).view(5, 2))
    mismatch = False
    try:
        dp.check_replicated(torch.full((5, 2), float(rank)))
    except ValueError:
        mismatch = True
    out_q.put((rank, mine.numpy(), flat.numpy(), grad.numpy(), after.numpy(), mismatch))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_host_logic():
    from suave_amd import supervised as sup

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(np.concatenate([res[0][1], res[1][1]]), np.arange(51))  # shards tile the batch
    assert (res[0][2] == 3.0).all() and np.array_equal(res[0][2], res[1][2])       # both buckets cover every entry once
    torch.manual_seed(0)
    clf = sup.ClassificationHead(4, 3, class_weight=np.array([0.5, 1.0, 2.5], np.float32))
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(51, 4, generator=g)
    y = torch.randint(0, 3, (51,), generator=g)
    clf.loss(clf(lat), y).backward()
    want = torch.cat([p.grad.reshape(-1) for p in clf.parameters()]).numpy()
    assert np.array_equal(res[0][3], res[1][3]) and np.allclose(res[0][3], want, rtol=1e-5, atol=1e-7)
    assert np.array_equal(res[0][4], res[1][4])
    assert res[0][5] and res[1][5]


def _shard_worker(rank, world, port, out_q):
    """Row-sharded tables and generation spans (round 3): the host logic of fit(sharded=True) and of the rank-aware
    sample / impute family."""

    import pandas as pd

    from suave_amd import data as data_utils
    from suave_amd.types import Schema

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    # generation spans: same request on every rank -> disjoint shards keyed by the global row; explicit offsets pass through
    res["span"] = parallel.generation_span(1001, None)
    res["span_explicit"] = parallel.generation_span(10, 500)
    # all_gather_rows with different row counts per rank
    t = torch.arange(3 + rank, dtype=torch.float32).unsqueeze(1) + 100 * rank
    res["gathered"] = parallel.all_gather_rows(t).numpy()
    # normalisation statistics of a table whose rows are block-sharded over the ranks == those of the whole table
    rng = np.random.default_rng(0)
    n = 400
    X = pd.DataFrame({"a": rng.normal(5, 2, n), "b": np.exp(rng.normal(0, 1, n)), "c": rng.poisson(3, n).astype(float),
                      "d": rng.integers(0, 3, n), "e": rng.integers(0, 4, n)})
    X.loc[rng.random(n) < 0.1, "a"] = np.nan
    X.loc[:, "d"] = X["d"].where(X.index >= 150, 0)  # rank 0's block sees a single category
    sch = Schema({"a": {"type": "real"}, "b": {"type": "pos"}, "c": {"type": "count"},
                  "d": {"type": "cat", "n_classes": 3}, "e": {"type": "ordinal", "n_classes": 4}})
    cut = [0, 150, n]
    res["stats"] = data_utils.fit_norm_stats(X.iloc[cut[rank]:cut[rank + 1]], sch, sharded=True)
    res["stats_whole"] = data_utils.fit_norm_stats(X, sch)
    out_q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_host_logic():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0]["span"] == (0, 501) and got[1]["span"] == (501, 500)
    assert got[0]["span_explicit"] == got[1]["span_explicit"] == (500, 10)
    want = np.concatenate([np.arange(3.0), 100 + np.arange(4.0)])[:, None]
    assert np.array_equal(got[0]["gathered"], want) and np.array_equal(got[1]["gathered"], want)
    for r in range(world):
        whole, part = got[r]["stats_whole"], got[r]["stats"]
        for col in whole:
            for key, v in whole[col].items():
                if isinstance(v, float):
                    assert abs(part[col][key] - v) <= 1e-12 * max(1.0, abs(v)), (col, key)
                else:
                    assert part[col][key] == v, (col, key)


def test_device_classification_metrics_match_the_host_functions():
    """supervised.device_classification_metrics (torch reductions / sorts, run on the device inside fit) against the
    numpy statements of evaluate.py's Brier / ECE / AUROC it replaces on the validation path, ties included."""

    import numpy as np
    import torch

    from suave_amd import supervised as sup

    rng = np.random.default_rng(0)
    for C, n in ((2, 1000), (3, 777), (5, 64)):
        logits = rng.normal(size=(n, C)).astype(np.float32) * 2
        logits[::7] = logits[0]            # exact ties in every score column
        logits = np.round(logits, 1)       # and many more
        p = torch.softmax(torch.from_numpy(logits), dim=-1)
        t = rng.integers(0, C, n)
        got = sup.device_classification_metrics(p, torch.from_numpy(t)).numpy()
        pn = p.numpy()
        want = [sup.compute_brier(pn, t), sup.compute_ece(pn, t), sup.compute_auroc(pn, t)]
        assert np.allclose(got, want, rtol=1e-12, atol=1e-14), (C, got, want)
    # probabilities on bin edges, one class only (AUROC undefined -> NaN), empty input
    p = torch.tensor([[0.9, 0.1], [0.8, 0.2], [0.0, 1.0], [1.0, 0.0], [0.5, 0.5]], dtype=torch.float32)
    t = np.array([0, 0, 1, 0, 1])
    got = sup.device_classification_metrics(p, torch.from_numpy(t)).numpy()
    assert np.allclose(got, [sup.compute_brier(p.numpy(), t), sup.compute_ece(p.numpy(), t), sup.compute_auroc(p.numpy(), t)])
    one = sup.device_classification_metrics(p, torch.zeros(5, dtype=torch.int64)).numpy()
    assert np.isfinite(one[0]) and np.isfinite(one[1]) and np.isnan(one[2])
    assert np.isnan(sup.device_classification_metrics(p[:0], torch.zeros(0, dtype=torch.int64)).numpy()).all()
