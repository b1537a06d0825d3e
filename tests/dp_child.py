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
}


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
    if backend is not None:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world),
                          SUAVE_DP_SINGLE_RANK="1")
        kw = dict(device_id=torch.device("cuda:0")) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    cfg = dict(CASES[case])
    fit_kw = cfg.pop("fit")
    cfg.setdefault("val_split", 0.0)
    df, y, schema = frame()
    model = SUAVE(schema, latent_dim=4, hidden_dims=(16, 8), random_state=3, device="cuda:0", **cfg)
    model.fit(df, y if cfg["behaviour"] == "supervised" else None, **fit_kw)
    res = {"params": model._engine.params.cpu().numpy(), "step": np.int64(model._step)}
    if model._classifier is not None:
        res["classifier"] = torch.cat([p.detach().reshape(-1) for p in model._classifier.parameters()]).cpu().numpy()
    if model._warmup_val_history:
        res["val_nll"] = np.asarray([h["nll"] for h in model._warmup_val_history])
    res["train_recon"] = np.asarray([h["reconstruction"] for h in model._train_history])
    np.savez(out, **res)
    if backend is not None:
        assert model._dp is not None and model._dp.host_staged == (backend != "nccl")
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
