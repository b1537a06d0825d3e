"""Where the wall time of SUAVE.fit goes on a 1 M x 64 table (cProfile, top entries by cumulative time).
usage (through gpurun, repo root): python scripts/fit_profile.py [rows] [epochs]"""
import cProfile, os, pstats, sys, time
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suave_amd.model import SUAVE
from suave_amd.types import Schema

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(0)
spec, cols = {}, {}
cyc, classes, seen = ["real", "pos", "count", "cat", "ordinal"], [2, 3, 4, 5, 8], {"cat": 0, "ordinal": 0}
for j in range(64):
    t, name = cyc[j % 5], f"f{j}"
    if t == "real": cols[name] = rng.standard_normal(n); spec[name] = {"type": t}
    elif t == "pos": cols[name] = rng.lognormal(size=n); spec[name] = {"type": t}
    elif t == "count": cols[name] = rng.poisson(3, size=n).astype(float); spec[name] = {"type": t}
    else:
        c = classes[seen[t] % 5]; seen[t] += 1
        cols[name] = rng.integers(0, c, size=n); spec[name] = {"type": t, "n_classes": c}
X = pd.DataFrame(cols)
y = pd.Series(rng.integers(0, 2, n))
behaviour = os.environ.get("BEHAVIOUR", "unsupervised")
model = SUAVE(Schema(spec), behaviour=behaviour, latent_dim=32, n_components=10, hidden_dims=(256, 128), dropout=0.1,
              batch_size=int(os.environ.get("BATCH", "65536")), random_state=0)
kw = dict(warmup_epochs=epochs) if behaviour == "supervised" else dict(epochs=epochs)
if behaviour == "supervised":
    kw.update(head_epochs=1, finetune_epochs=1)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
model.fit(X, y if behaviour == "supervised" else None, **kw)
torch.cuda.synchronize()
pr.disable()
print(f"fit({n} rows, {epochs} epochs, {behaviour}): {time.perf_counter() - t0:.2f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
t0 = time.perf_counter(); s = model.sample(n); print(f"sample({n}): {time.perf_counter() - t0:.2f} s")
t0 = time.perf_counter(); z = model.encode(X); print(f"encode({n}): {time.perf_counter() - t0:.2f} s")
