"""Data parallelism inside ``SUAVE.fit`` on the GPU: two fresh child processes, both on cuda:0, gloo between them.

Each rank fits on its half of every global batch (tests/dp_child.py).  Asserted: the two ranks end with
bit-identical parameters, and those equal a one-process fit on the whole batches to 1e-5 -- the kernels key their
noise by (seed, step, table row), so the only difference is fp32 summation order.  The collectives run over gloo
through host memory here (one GPU); on an 8-GPU node the same code path reduces over RCCL with the decoder-side
bucket overlapped with the encoder backward (suave_amd/parallel.py).
"""

import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port() -> str:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return str(port)


def _run(case, world, tmp_path, backend=None):
    port = _free_port()
    outs = [str(tmp_path / f"{case}_w{world}_{backend}_r{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_child.py"), str(r), str(world), port, case, outs[r]]
                              + ([backend] if backend else []),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            log, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(log)
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-4000:]
    return [dict(np.load(o)) for o in outs]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["unsup_warmup", "sup_warmup", "sup_schedule", "sup_schedule_drop",
                                  "sup_schedule_sharded", "unsup_sharded", "s64_sharded"])
def test_two_rank_fit_matches_one_rank(case, tmp_path):
    two = _run(case, 2, tmp_path)
    one = _run(case, 1, tmp_path)[0]
    assert two[0]["step"] == two[1]["step"] == one["step"] and one["step"] >= 3
    for key in two[0]:
        assert np.array_equal(two[0][key], two[1][key]), f"{key}: the ranks' replicas diverged"
    assert np.isfinite(one["params"]).all()
    for key in ("params", "classifier", "val_nll", "train_recon"):
        if key in one:
            assert two[0][key].shape == one[key].shape
            err = np.max(np.abs(two[0][key].astype(np.float64) - one[key]))
            assert err <= 1e-5, f"{case}/{key}: 2-rank fit differs from the 1-rank fit by {err:.3g}"


@pytest.mark.gpu
def test_rccl_overlap_plumbing_on_one_rank(tmp_path):
    """The RCCL path of DataParallel (side stream behind the library's decoder-ready event, async all-reduces, the
    compute stream waiting for both) on a one-rank "nccl" group, the most one GPU can host: a one-rank sum is the
    identity, so the ELBO warm-up must reproduce the plain fit bit for bit, and the whole supervised schedule to
    1e-5 (its sharded cross-entropy is the same mean written as sum / weight: not the same bits)."""

    one = _run("sup_warmup", 1, tmp_path)[0]
    rccl = _run("sup_warmup", 1, tmp_path, backend="nccl")[0]
    for key in one:
        assert np.array_equal(one[key], rccl[key]), key
    one = _run("sup_schedule", 1, tmp_path)[0]
    rccl = _run("sup_schedule", 1, tmp_path, backend="nccl")[0]
    for key in one:
        assert np.max(np.abs(one[key].astype(np.float64) - rccl[key])) <= 1e-5, key


@pytest.mark.gpu
def test_two_rank_generation_is_the_one_rank_output_sharded_by_rows(tmp_path):
    """SUAVE.sample / impute / impute_mc / encode / predict_proba(attr) under torch.distributed: every rank computes
    its ``shard_rows`` share keyed by the GLOBAL row index, with no collective.  The ranks' shards are disjoint, cover
    every row, and their union is the one-rank output: bit for bit on the discrete columns (and on everything else
    the kernels produce: same kernels, same keys, same inputs)."""

    two = _run("generation", 2, tmp_path)
    one = _run("generation", 1, tmp_path)[0]
    assert np.array_equal(two[0]["params"], two[1]["params"])
    for key in [k for k in one if k.startswith("gen_")]:
        a, b, ref = two[0][key], two[1][key], one[key]
        if key == "gen_proba":  # gathered on request: every rank holds all rows
            assert np.array_equal(a, b) and np.array_equal(a, ref), key
            continue
        have_a, have_b = ~np.isnan(a).all(axis=1), ~np.isnan(b).all(axis=1)
        assert not (have_a & have_b).any() and (have_a | have_b).all(), f"{key}: shards overlap or leave rows out"
        union = np.where(have_a[:, None], a, b)
        assert np.array_equal(union, ref, equal_nan=True), key
