"""Run the fused generative decode (decode.hip) alone: S64 model, n rows, `reps` calls.
Under the profiler:  rocprofv3 --kernel-trace --stats -- python3 scripts/decode_probe.py
                     rocprofv3 --pmc <counters> -- python3 scripts/decode_probe.py   (scripts/pmc_decode.sh)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from suave_amd.engine import Engine  # noqa: E402
from suave_amd.plan import ModelLayout  # noqa: E402
from suave_amd.types import Schema  # noqa: E402

n = int(os.environ.get("N", 1 << 20))
reps = int(os.environ.get("REPS", 4))
spec, stats = bench.s64_schema()
kind = os.environ.get("SCHEMA", "s64")  # s64 | real | pos | count | cat4 | cat8 | ord5: 64 columns of one type (cost per type)
if kind != "s64":
    t = kind.rstrip("0123456789")
    t = {"ord": "ordinal"}.get(t, t)
    c = int(kind[len(kind.rstrip("0123456789")):] or 0)
    spec, stats = {}, {}
    for j in range(64):
        spec[f"f{j}"] = {"type": t, "n_classes": c} if c else {"type": t}
        stats[f"f{j}"] = ({} if c else {"mean": 50.0, "std": 12.0} if t == "real" else
                          {"mean_log": 1.2, "std_log": 0.45} if t == "pos" else {"offset": 1.0})
layout = ModelLayout(Schema(spec), bench.K_COMP, bench.LATENT, bench.HIDDEN, 0.1, "supervised", stats)
eng = Engine(layout, "cuda:0")
eng.load_state(layout.init_params(seed=0))
g = torch.Generator(device="cuda:0")
g.manual_seed(3)
z = torch.randn((n, bench.LATENT), device="cuda:0", generator=g)
comp = torch.randint(0, bench.K_COMP, (n,), device="cuda:0", generator=g, dtype=torch.int32)
eng.decode_sample(z, component=comp, seed=8)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(reps):
    eng.decode_sample(z, component=comp, seed=8 + i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"[{kind}] decode_sample: {n} rows in {dt * 1e3:.3f} ms = {n / dt / 1e6:.1f} M rows/s")
