"""End-to-end cost of the data formats either side of the path (f1 / f2) on the S64 schema: DataFrame -> packed table
on the host (data.pack_frame + upload) against the device packer (data.pack_frame_device), and generated rows ->
DataFrame through data.frame_from_packed (per-chunk .cpu() + pd.concat) against data.FrameAssembler.
usage (through gpurun, repo root): python scripts/data_layer_timing.py [rows]"""
import os, sys, time
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suave_amd import data as du
from suave_amd.engine import Engine, PackedTable
from suave_amd.plan import ModelLayout
from suave_amd.types import Schema

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
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
for c in list(X.columns)[:10]:
    X.loc[rng.random(n) < 0.1, c] = np.nan
schema = Schema(spec)
t0 = time.perf_counter(); stats = du.fit_norm_stats(X, schema); t_stats = time.perf_counter() - t0
eng = Engine(ModelLayout(schema, 10, 32, (256, 128), 0.1, "supervised", stats), "cuda:0")
dev = eng.device
def sync(): torch.cuda.synchronize(dev)
du.pack_frame_device(X.iloc[:1000], schema, stats, eng); sync()
t0 = time.perf_counter(); xh, ch, mh = du.pack_frame(X, schema, stats); tab = PackedTable(xh, ch, mh, dev); sync(); t_host = time.perf_counter() - t0
t0 = time.perf_counter(); xd, cd, md = du.pack_frame_device(X, schema, stats, eng); tab2 = PackedTable(xd, cd, md, dev); sync(); t_dev = time.perf_counter() - t0
print(f"rows {n}: fit_norm_stats {t_stats:.2f} s; DataFrame -> packed table in HBM: host writer {t_host:.2f} s, device packer {t_dev:.2f} s ({t_host / t_dev:.1f}x)")
assert np.array_equal(md.cpu().numpy(), mh) and np.array_equal(cd.cpu().numpy(), ch)
# generation side: decoded rows (device) -> DataFrame
chunk = 262144
on = torch.from_numpy(xh).to(dev) * 3 + 1
oc = torch.from_numpy(np.maximum(ch, 0)).to(dev)
sync()
t0 = time.perf_counter()
frames = [du.frame_from_packed(on[s:s + chunk].cpu().numpy(), oc[s:s + chunk].cpu().numpy(), schema, stats) for s in range(0, n, chunk)]
ref = pd.concat(frames, ignore_index=True) if len(frames) > 1 else frames[0]
t_old = time.perf_counter() - t0
t0 = time.perf_counter()
asm = du.FrameAssembler(schema, stats, eng, n, chunk)
for s in range(0, n, chunk):
    asm.add(s, on[s:s + chunk], oc[s:s + chunk])
got = asm.frame()
t_new = time.perf_counter() - t0
print(f"rows {n}: decoded rows -> DataFrame: per-chunk host path {t_old:.2f} s, device assembly {t_new:.2f} s ({t_old / t_new:.1f}x)")
pd.testing.assert_frame_equal(got, ref)
