"""Child process of tests/test_hip_dp_fit.py (not collected by pytest): one rank of a data-parallel ``SUAVE.fit``.

    python tests/dp_child.py <rank> <world> <port> <case> <out.npz> [backend]

world == 1 runs the same fit without torch.distributed (the one-GPU run the ranks are compared with), unless a
backend is named: "nccl" with world 1 drives the RCCL stream / event plumbing on the box's one GPU.  Every
rank builds the same synthetic mixed-type frame from a fixed seed, fits on cuda:0 (ranks of a one-GPU box share
the card; the collectives go over gloo through host memory) and writes its parameters.
"""

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import pandas as pd
import torch
import torch.distributed as dist

from suave_amd.model import SUAVE
from suave_amd.types import Schema

CASES = {
    # 3 optimisation steps of the warm-up (144 train rows, batch 48 -> each of 2 ranks takes 24 rows a step)
    "unsup_warmup": dict(behaviour="unsupervised", n_components=3, dropout=0.1, fit=dict(epochs=1, batch_size=48)),
    "sup_warmup": dict(behaviour="supervised", n_components=3, dropout=0.1,
                       fit=dict(epochs=1, batch_size=48, head_epochs=0, finetune_epochs=0, decoder_refine_epochs=0)),
    # the whole supervised schedule incl. validation-driven best-state selection, ragged last batches (batch 50)
    "sup_schedule": dict(behaviour="supervised", n_components=2, dropout=0.0, val_split=0.2,
                         decoder_refine_mode="prior_em_decoder",
                         fit=dict(epochs=2, batch_size=50, head_epochs=2, finetune_epochs=2, decoder_refine_epochs=2)),
    # the same schedule with dropout in the classifier head too: its masks are keyed by the table row
    "sup_schedule_drop": dict(behaviour="supervised", n_components=2, dropout=0.1, val_split=0.2, head_hidden_dims=(8,),
                              fit=dict(epochs=1, batch_size=50, head_epochs=2, finetune_epochs=2, decoder_refine_epochs=1)),
    # ROW-SHARDED table: every rank passes (and packs) only its block of the rows (fit(sharded=True))
    "sup_schedule_sharded": dict(behaviour="supervised", n_components=2, dropout=0.1, val_split=0.2, sharded=True,
                                 decoder_refine_mode="prior_em_decoder",
                                 fit=dict(epochs=2, batch_size=50, head_epochs=2, finetune_epochs=2, decoder_refine_epochs=2)),
    "unsup_sharded": dict(behaviour="unsupervised", n_components=3, dropout=0.1, val_split=0.25, sharded=True,
                          fit=dict(epochs=2, batch_size=48)),
    # BASELINE config #2 / #3 shape: the 64-column schema, 65 536-row global batches, row-sharded table
    "s64_sharded": dict(behaviour="supervised", n_components=10, dropout=0.1, sharded=True, s64_rows=131072 + 4096,
                        latent_dim=32, hidden_dims=(256, 128),
                        fit=dict(epochs=1, batch_size=65536, head_epochs=0, finetune_epochs=0, decoder_refine_epochs=0)),
    # generation after a (replicated) fit: every rank makes the same calls and returns its shard of the rows
    "generation": dict(behaviour="supervised", n_components=3, dropout=0.0, generate=True,
                       fit=dict(epochs=2, batch_size=48, head_epochs=1, finetune_epochs=1, decoder_refine_epochs=0)),
}


def s64_frame(n, seed=5):
    """A raw frame of the 64-column benchmark schema (bench.s64_schema): 10 % of the cells missing."""

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    spec, _ = bench.s64_schema()
    rng = np.random.default_rng(seed)
    cols = {}
    for name, sp in spec.items():
        t = sp["type"]
        if t == "real":
            v = rng.normal(50.0, 12.0, n)
        elif t == "pos":
            v = np.exp(rng.normal(1.0, 0.5, n))
        elif t == "count":
            v = rng.poisson(3.0, n).astype(np.float64)
        else:
            v = rng.integers(0, sp["n_classes"], n).astype(np.float64)
        v[rng.random(n) < 0.1] = np.nan
        cols[name] = v
    return pd.DataFrame(cols), rng.integers(0, 2, n), Schema(spec)


def frame(n=144, seed=11):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, 2, n)
    df = pd.DataFrame({
        "r0": rng.normal(size=n) + y, "r1": rng.normal(size=n) * 2.0,
        "p0": np.exp(rng.normal(size=n) * 0.5), "c0": rng.poisson(3.0 + y, n).astype(float),
        "k0": rng.integers(0, 4, n).astype(float), "o0": rng.integers(0, 3, n).astype(float)})
    for c in df.columns:  # missing cells
        df.loc[rng.random(n) < 0.15, c] = np.nan
    schema = Schema({"r0": {"type": "real"}, "r1": {"type": "real"}, "p0": {"type": "pos"}, "c0": {"type": "count"},
                     "k0": {"type": "cat", "n_classes": 4}, "o0": {"type": "ordinal", "n_classes": 3}})
    return df, y, schema


def main():
    rank, world, port, case, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    backend = sys.argv[6] if len(sys.argv) > 6 else ("gloo" if world > 1 else None)

    def init_group():
        if backend is not None:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world),
                              SUAVE_DP_SINGLE_RANK="1")
            kw = dict(device_id=torch.device("cuda:0")) if backend == "nccl" else {}
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    # (the generation case fits BEFORE the group exists: every rank runs the same one-GPU fit -- bit-reproducible -- so
    # that the ranks' generation shards can be compared with the one-rank output bit for bit)
    if not CASES[case].get("generate"):
        init_group()
    cfg = dict(CASES[case])
    fit_kw = cfg.pop("fit")
    sharded, generate, s64_rows = cfg.pop("sharded", False), cfg.pop("generate", False), cfg.pop("s64_rows", 0)
    cfg.setdefault("val_split", 0.0)
    cfg.setdefault("latent_dim", 4)
    cfg.setdefault("hidden_dims", (16, 8))
    df, y, schema = s64_frame(s64_rows) if s64_rows else frame()
    if sharded and world > 1:  # this rank's block of the rows (uneven on purpose)
        cut = [0] + [int(len(df) * (r + 1) / world) + (7 if r + 1 < world else 0) for r in range(world)]
        df, y = df.iloc[cut[rank]:cut[rank + 1]], y[cut[rank]:cut[rank + 1]]
        fit_kw["sharded"] = True
    model = SUAVE(schema, random_state=3, device="cuda:0", **cfg)
    model.fit(df, y if cfg["behaviour"] == "supervised" else None, **fit_kw)
    res = {"params": model._engine.params.cpu().numpy(), "step": np.int64(model._step)}
    if generate:  # the ranks' shards (index = global rows) -> dense arrays with NaN where this rank computed nothing
        init_group()
        full, y_full, _ = frame()
        n_gen = 301
        lab = np.random.default_rng(2).integers(0, 2, n_gen)

        def dense(part, n):
            out = np.full((n, part.shape[1]), np.nan)
            out[np.asarray(part.index)] = part.astype(np.float64).to_numpy()
            return out

        res["gen_sample"] = dense(model.sample(n_gen, conditional=True, y=lab, seed=5), n_gen)
        res["gen_prior"] = dense(model.sample(n_gen, seed=6, batch_size=64), n_gen)
        imp = model.impute_mc(full, L=32, seed=7)
        res["gen_impute_mc"] = dense(imp, len(full))
        res["gen_impute"] = dense(model.impute(full, seed=8), len(full))
        res["gen_proba"] = dense(pd.DataFrame(model.predict_proba(full, attr="k0", L=16, seed=9, gather=True).numpy()), len(full))
        enc = model.encode(full)
        b, _n = __import__("suave_amd.parallel", fromlist=["x"]).generation_span(len(full), None)
        e_d = np.full((len(full), enc.shape[1]), np.nan)
        e_d[b:b + len(enc)] = enc
        res["gen_encode"] = e_d
    if model._classifier is not None:
        res["classifier"] = torch.cat([p.detach().reshape(-1) for p in model._classifier.parameters()]).cpu().numpy()
    if model._warmup_val_history:
        res["val_nll"] = np.asarray([h["nll"] for h in model._warmup_val_history])
    res["train_recon"] = np.asarray([h["reconstruction"] for h in model._train_history])
    np.savez(out, **res)
    if backend is not None:
        assert generate or (model._dp is not None and model._dp.host_staged == (backend != "nccl"))
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
