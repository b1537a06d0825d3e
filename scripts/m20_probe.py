"""Tuning aid: the M20-shaped fits of bench.m20_subrecord (BASELINE config #1), for `rocprofv3 --kernel-trace --stats --
python3 scripts/m20_probe.py [unsupervised|supervised]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
import bench
from suave_amd.model import SUAVE
from suave_amd.types import Schema

which = sys.argv[1] if len(sys.argv) > 1 else "unsupervised"
spec = bench.m20_schema()
rng = np.random.default_rng(20201021)
n = 10_000
cols = {}
for name, sp in spec.items():
    t = sp["type"]
    v = (rng.normal(50.0, 12.0, n) if t == "real" else np.exp(rng.normal(1.0, 0.5, n)) if t == "pos" else
         rng.poisson(3.0, n).astype(np.float64) if t == "count" else rng.integers(0, sp["n_classes"], n).astype(np.float64))
    v[rng.random(n) < 0.1] = np.nan
    cols[name] = v
X, y = pd.DataFrame(cols), (rng.random(n) < 0.3).astype(np.int64)
kw = dict(latent_dim=32, n_components=10, hidden_dims=(256, 128), batch_size=256, val_split=0.2, random_state=0, device="cuda:0")
if which == "unsupervised":
    SUAVE(Schema(spec), behaviour="unsupervised", **kw).fit(X, epochs=5)
else:
    SUAVE(Schema(spec), behaviour="supervised", **kw).fit(X, y, warmup_epochs=3, head_epochs=1, finetune_epochs=1, decoder_refine_epochs=0)
torch.cuda.synchronize()
