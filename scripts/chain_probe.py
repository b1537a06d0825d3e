#!/usr/bin/env python3
"""Tuning aid: the S64 decoder chain on N rows, a few launches (run under rocprofv3 --kernel-trace / --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import suave_oracle as orc
from suave_amd.engine import Engine
from tests.helpers import layout_from_columns, s64_columns

n = int(os.environ.get("ROWS", 1 << 20))
cols = s64_columns(64)
om = orc.OracleModel(cols, n_components=10, latent_dim=32, hidden=(256, 128))
eng = Engine(layout_from_columns(cols, K=10, D=32, hidden=(256, 128)), "cuda:0")
eng.load_state(om.init_params(seed=1))
z = torch.randn((n, 32), device="cuda:0")
comp = torch.randint(0, 10, (n,), device="cuda:0", dtype=torch.int32)
for _ in range(3):
    eng.decode_sample(z, component=comp, seed=1)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(5):
    eng.decode_sample(z, component=comp, seed=1)
ev1.record()
torch.cuda.synchronize()
print("decode_sample ms per call:", ev0.elapsed_time(ev1) / 5, "rows", n)
