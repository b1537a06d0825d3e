"""Headline benchmark: ELBO train rows/sec (BASELINE.json), S64 schema.

A "step" is one pass of the hot path over one batch: packed-batch gather ->
encoder -> Gumbel assignment + reparameterisation -> decoder -> fused
likelihood + KL -> full backward -> (N>1: RCCL all-reduce of the flat gradient)
-> Adam.  Warm-up-phase objective of the reference (suave/model.py:1496-1506),
dropout 0.1, fp32 arithmetic, noise drawn in-kernel from the counter-based
generators, inputs resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

K_COMP, LATENT, HIDDEN = 10, 32, (256, 128)
PROF_CLASSES = ["gemm", "likelihood_kl", "latent", "expand", "reduce", "adam", "sample", "other"]
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 = dense fp16 (same guide); an upper bound here: the grouped forms do not use this pipe
# plane products per fp32 product and the MFMA they run on, by GEMM arithmetic mode (include/suave_hip.h)
ARITHMETIC = {4: (3, "two fp16 planes (hi + scaled residual), three plane products on v_mfma_f32_32x32x16_f16"),
              3: (6, "three bf16 planes, six plane products on v_mfma_f32_32x32x16_bf16"),
              0: (1, "exact fp32 on v_mfma_f32_32x32x2_f32")}
PEAK_HBM_GBS = 8000.0


def s64_schema(n_cols=64):
    """columns cycle [real,pos,count,cat,ordinal]; class counts cycle [2,3,4,5,8]."""

    cyc = ["real", "pos", "count", "cat", "ordinal"]
    classes = [2, 3, 4, 5, 8]
    seen = {"cat": 0, "ordinal": 0}
    spec, stats = {}, {}
    for j in range(n_cols):
        t = cyc[j % 5]
        name = f"f{j}"
        if t in ("cat", "ordinal"):
            spec[name] = {"type": t, "n_classes": classes[seen[t] % 5]}
            seen[t] += 1
            stats[name] = {}
        else:
            spec[name] = {"type": t}
            stats[name] = ({"mean": 50.0, "std": 12.0} if t == "real" else
                           {"mean_log": 1.2, "std_log": 0.45} if t == "pos" else {"offset": 1.0})
    return spec, stats


def w1024_schema():
    """400 real, 200 pos, 100 count, 224 cat (classes cycle 2..256), 100 ordinal
    (classes cycle 3,5,7,10), interleaved by a seeded shuffle; Y = 15 605, encoder input 16 629."""

    import numpy as _np

    cat_c, ord_c = [2, 4, 8, 16, 32, 64, 128, 256], [3, 5, 7, 10]
    kinds = [("real", None)] * 400 + [("pos", None)] * 200 + [("count", None)] * 100
    kinds += [("cat", cat_c[i % 8]) for i in range(224)] + [("ordinal", ord_c[i % 4]) for i in range(100)]
    order = _np.random.default_rng(4).permutation(len(kinds))
    spec, stats = {}, {}
    for j, i in enumerate(order):
        t, c = kinds[i]
        name = f"f{j}"
        if c is not None:
            spec[name] = {"type": t, "n_classes": c}
            stats[name] = {}
        else:
            spec[name] = {"type": t}
            stats[name] = ({"mean": 50.0, "std": 12.0} if t == "real" else
                           {"mean_log": 1.2, "std_log": 0.45} if t == "pos" else {"offset": 1.0})
    return spec, stats


def synth_table_device(layout, n_rows, device, seed):
    """Synthetic S64 table generated directly in HBM (torch is plumbing here, not the measured path)."""

    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    schema = layout.schema
    names = list(schema.feature_names)
    mask = (torch.rand((n_rows, len(names)), device=device, generator=g) >= 0.1)
    xnum = torch.empty((n_rows, layout.n_num), device=device, dtype=torch.float32)
    codes = torch.empty((n_rows, layout.n_disc), device=device, dtype=torch.int16)
    i_num = i_disc = 0
    for j, n in enumerate(names):
        spec = schema[n]
        if spec.type == "real":
            col = torch.randn(n_rows, device=device, generator=g)
        elif spec.type == "pos":
            raw = torch.exp(1.0 + 0.5 * torch.randn(n_rows, device=device, generator=g))
            col = (torch.log1p(raw) - 1.2) / 0.45
        elif spec.type == "count":
            cnt = torch.poisson(torch.full((n_rows,), 3.0, device=device), generator=g)
            col = torch.log(cnt + 1.0)
        if spec.type in ("real", "pos", "count"):
            xnum[:, i_num] = torch.where(mask[:, j], col, torch.zeros_like(col))
            i_num += 1
        else:
            c = torch.randint(0, spec.n_classes, (n_rows,), device=device, generator=g).to(torch.int16)
            codes[:, i_disc] = torch.where(mask[:, j], c, torch.full_like(c, -1))
            i_disc += 1
    return xnum, codes, mask.to(torch.uint8)


def gemm_flops_per_row(layout):
    """Algorithmic FLOPs of the dense layers, forward + backward (2*MAC, x3): SURVEY.md §8d."""

    # first layer, sparse-aware: the dense input columns plus ONE row of W1 per embedded
    # (wide categorical) column -- the one-hot blocks themselves are never multiplied
    n_embed = sum(1 for c in layout.cols if c.enc_embed >= 0)
    I, Y, K, D = layout.enc_in + n_embed, layout.total_y, layout.n_components, layout.latent_dim
    mac = 0
    prev = I
    for h in layout.hidden:
        mac += prev * h
        prev = h
    mac += prev * (K + 2 * K * D)
    prev = D
    for h in layout.hidden:
        mac += prev * h
        prev = h
    mac += prev * Y
    # wide categorical heads run as GEMMs too (logits_layer on [y_col ; s], decoder.py:201-206)
    from suave_amd.plan import WIDE_CAT_MIN_CLASSES
    for n in layout.schema.feature_names:
        sp = layout.schema[n]
        if sp.type == "cat" and int(sp.n_classes) >= WIDE_CAT_MIN_CLASSES:
            mac += (int(sp.n_classes) - 1) * (int(sp.n_classes) + K)
    return 6 * mac


def gemm_flops_executed_per_row(layout):
    """FLOPs the MFMA kernels actually execute per row: the supervised step evaluates only the logits and the
    selected component's mu / logvar rows of the encoder head (K + 2D of its K + 2KD columns)."""

    K, D = layout.n_components, layout.latent_dim
    h_last = layout.hidden[-1] if layout.hidden else layout.enc_in
    skipped = h_last * (2 * K * D - 2 * D) if layout.behaviour == "supervised" else 0
    return gemm_flops_per_row(layout) - 6 * skipped


def cpu_baseline(spec, stats, dropout, budget_s=15.0, batch=4096, latent=None, hidden=None, label="S64"):
    """The CPU oracle (numpy restatement of the reference path, oracle/) timed on this host's cores."""

    from oracle import suave_oracle as orc

    LATENT, HIDDEN = (latent or globals()["LATENT"]), tuple(hidden or globals()["HIDDEN"])
    cols = [orc.Column(n, s["type"], s.get("n_classes"), None, stats[n]) for n, s in spec.items()]
    om = orc.OracleModel(cols, n_components=K_COMP, latent_dim=LATENT, hidden=HIDDEN, dropout=dropout)
    P = om.init_params(seed=0)
    rng = np.random.default_rng(0)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests.helpers import synth_packed

    xnum, codes, mask = synth_packed(cols, batch, 0.1, seed=20201021)
    x_enc, data, masks = orc.dense_from_packed(om, xnum, codes, mask)
    state = {}
    times = []
    t_end = time.perf_counter() + budget_s
    step = 0
    while True:
        noise = {"exp": rng.exponential(size=(batch, K_COMP)).astype(np.float32),
                 "eps": rng.standard_normal((batch, LATENT)).astype(np.float32),
                 "enc_drop": [(rng.random((batch, h)) >= dropout) for h in HIDDEN],
                 "dec_drop": [(rng.random((batch, h)) >= dropout) for h in HIDDEN]}
        t0 = time.perf_counter()
        out = orc.elbo_supervised(om, P, x_enc, data, masks, noise, beta_scale=1.0, tau=1.0, training=True)
        step += 1
        orc.adam_step(P, out["grads"], state, lr=1e-3, step=step)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() > t_end and len(times) >= 4:
            break
    try:
        from threadpoolctl import threadpool_info
        cores = max([int(i.get("num_threads", 1)) for i in threadpool_info()] or [1])
    except Exception:
        cores = int(os.environ.get("OMP_NUM_THREADS", "1"))
    med = float(np.median(times[1:]))
    return {"value": batch / med, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": f"{len(times) - 1} ELBO steps (fwd+bwd+Adam) of the numpy oracle on {batch} {label} rows, median"}


def generation_rates(eng, table, rows, gen_rows, device, rank, world, dist, fence, prof, chunk=1 << 18, mc_chunk=1 << 14,
                     n_imp_max=1_000_000):
    """sample() / impute() rows/sec (BASELINE.json config #5 on this rank's shard): device-side throughput with the
    outputs left in HBM in the packed columnar layout (host DataFrame assembly is not part of the kernels' rate).
    Rows are sharded over ranks with no collective; every generator is keyed by the global row index."""

    import torch

    from suave_amd import lib as slib

    K, D = eng.layout.n_components, eng.layout.latent_dim
    # training-posterior cache of this rank's table (model.py:1669-1741), labels Bernoulli(0.3)
    n_cache = min(rows, 1_000_000)
    probs = torch.empty((n_cache, K), device=device)
    mu = torch.empty((n_cache, K, D), device=device)
    lv = torch.empty((n_cache, K, D), device=device)
    for s0 in range(0, n_cache, chunk):
        nb = min(chunk, n_cache - s0)
        o = eng.encode(table, table.row_base + s0, nb)
        probs[s0:s0 + nb] = o["probs"]
        mu[s0:s0 + nb] = o["enc_out"][:, K:K + K * D].view(nb, K, D)
        lv[s0:s0 + nb] = o["enc_out"][:, K + K * D:].view(nb, K, D)
    g = torch.Generator(device=device)
    g.manual_seed(99 + rank)
    train_y = (torch.rand(n_cache, device=device, generator=g) < 0.3).to(torch.int64)
    order = torch.argsort(train_y, stable=True)
    n0 = int((train_y == 0).sum().item())
    class_off = torch.tensor([0, n0, n_cache], device=device, dtype=torch.int64)
    labels = (torch.rand(chunk, device=device, generator=g) < 0.3).to(torch.int32)
    res = {}

    def timed(fn, n_rows_done, name):
        t_w = time.perf_counter()
        fn()  # warm-up pass (allocator, code objects)
        fence()
        if rank == 0:
            print(f"[bench] {name}: warm-up pass {time.perf_counter() - t_w:.2f} s", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        fn()
        fence()
        dt = time.perf_counter() - t0
        ms = (C.c_double * 8)()
        cnt = (C.c_int64 * 8)()
        if prof:  # kernel classes from a second, instrumented pass (the event brackets cost ~10 %: not in the timed one)
            slib.load().suave_profile_enable(1)
            fn()
            fence()
            slib.load().suave_profile_enable(0)
            slib.check(slib.load().suave_profile_read(ms, cnt, 8), "suave_profile_read")
        if dist is not None:
            t = torch.tensor([dt], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        res[name] = {"rows_per_s": world * n_rows_done / dt, "rows": world * n_rows_done, "seconds": dt}
        if prof and cnt[6] > 0:
            res[name]["sample_kernel_ms"] = ms[6]
            res[name]["gemm_ms"] = ms[0]
        return ms, cnt

    base = rank * gen_rows

    def run_sample():
        for s0 in range(0, gen_rows, chunk):
            nb = min(chunk, gen_rows - s0)
            z, comp = eng.conditional_latents(labels[:nb], class_off, order, probs, mu, lv, seed=7, row_offset=base + s0)
            eng.decode_sample(z, component=comp, seed=8, row_offset=base + s0)

    ms, cnt = timed(run_sample, gen_rows, "sample_conditional")
    if prof and cnt[6] > 0:
        L = eng.layout
        if ms[0] == 0.0 and len(L.hidden) == 2:
            # the fused decode kernel (decode.hip: decoder trunk -> y_projection -> head maps -> sample draw in one
            # launch, nothing but z / component in and the packed cells out): priced against the matrix pipe for the
            # trunk's algorithmic FLOPs (the head maps' extra MFMA layer and the sampler's VALU work are not counted)
            # and against HBM for its necessary bytes
            h1, h2 = L.hidden
            flops_row = 2 * (L.latent_dim * h1 + h1 * h2 + h2 * L.total_y)
            bytes_row = 4 * L.latent_dim + 4 + 4 * L.n_num + 2 * L.n_disc
            tf = flops_row * gen_rows / (ms[6] * 1e-3) / 1e12
            gbs = bytes_row * gen_rows / (ms[6] * 1e-3) / 1e9
            res["roofline_decode_fused"] = {"bound": "mfma", "kernel": "k_decode_fused", "achieved": tf,
                                            "peak": PEAK_BF16_MFMA_TFLOPS / 3, "unit": "TFLOP/s", "frac": tf / (PEAK_BF16_MFMA_TFLOPS / 3),
                                            "algorithmic_flops_per_row": flops_row, "hbm_GBps": gbs,
                                            "hbm_frac": gbs / PEAK_HBM_GBS, "algorithmic_bytes_per_row": bytes_row,
                                            "kernel_ms": ms[6]}
        else:
            # k_sample_heads against HBM: reads y_full (4 B x Y) and writes the packed columnar output
            bytes_row = 4 * L.total_y + 4 * L.n_num + 2 * L.n_disc
            gbs = bytes_row * gen_rows / (ms[6] * 1e-3) / 1e9
            res["roofline_sample_heads"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                            "frac": gbs / PEAK_HBM_GBS, "algorithmic_bytes_per_row": bytes_row}
    n_imp = min(rows, n_imp_max)

    def run_impute():  # reference-exact single pass at the posterior-mean latent, hard assignment (model.py:3627-3651)
        for s0 in range(0, n_imp, chunk):
            nb = min(chunk, n_imp - s0)
            z, comp = eng.encode_hard(table, table.row_base + s0, nb)
            eng.decode_sample(z, component=comp, seed=9, row_offset=rank * n_imp + s0)

    timed(run_impute, n_imp, "impute_single_pass")
    Ldraws = 32

    def run_impute_mc():  # posterior-predictive MC, 32 draws per row, every cell decoded (model.py:3304-3421 generalised)
        for s0 in range(0, n_imp, mc_chunk):
            nb = min(mc_chunk, n_imp - s0)
            o = eng.encode(table, table.row_base + s0, nb)
            z, comp = eng.posterior_draws(o["enc_out"], Ldraws, seed=10, row_offset=rank * n_imp + s0, row_major=True)
            eng.decode_sample(z, component=comp, seed=11, row_offset=(rank * n_imp + s0) * Ldraws)

    timed(run_impute_mc, n_imp, "impute_mc32")
    res["impute_mc32"]["decoded_rows_per_s"] = res["impute_mc32"]["rows_per_s"] * Ldraws
    return res


def m20_schema():
    """SURVEY.md §8d config #1 ("MIMIC-demo-shaped"): 8 real, 5 pos, 2 count, 3 cat (C = 2, 2, 4), 2 ordinal (C = 4, 5)."""

    spec = {}
    for i in range(8):
        spec[f"r{i}"] = {"type": "real"}
    for i in range(5):
        spec[f"p{i}"] = {"type": "pos"}
    for i in range(2):
        spec[f"c{i}"] = {"type": "count"}
    for i, c in enumerate((2, 2, 4)):
        spec[f"k{i}"] = {"type": "cat", "n_classes": c}
    for i, c in enumerate((4, 5)):
        spec[f"o{i}"] = {"type": "ordinal", "n_classes": c}
    return spec


def m20_subrecord(device):
    """BASELINE config #1 through the PRODUCT: ``SUAVE.fit`` on a 10 000-row M20 frame (8 000 train / 2 000 validation),
    latent 32, K = 10, hidden (256, 128), batch 256 -- the runs SURVEY.md section 6 timed on the reference (PyTorch
    eager, 8 CPU threads, survey container), with the numpy oracle's step on this host beside them."""

    import pandas as pd
    import torch

    from suave_amd.model import SUAVE
    from suave_amd.types import Schema

    spec = m20_schema()
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
    out = {"workload": "M20: 10 000 rows x 20 mixed cols (8 real / 5 pos / 2 count / 3 cat / 2 ordinal, 10% missing), "
                       "val_split 0.2, latent 32, K=10, hidden (256,128), batch 256, SUAVE.fit end to end (host loop, "
                       "per-epoch validation, pandas -> packed table included)"}
    common = dict(latent_dim=32, n_components=10, hidden_dims=(256, 128), batch_size=256, val_split=0.2, random_state=0,
                  device=str(device))

    def timed(fn):
        fn()  # warm-up: code objects, allocator, first-call plan creation
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, float(m._train_history[-1]["reconstruction"])

    t, rec = timed(lambda: SUAVE(Schema(spec), behaviour="unsupervised", **common).fit(X, epochs=5))
    out["unsupervised_warmup5"] = {"seconds": t, "rows_per_s": 5 * 8000 / t, "final_reconstruction": rec}
    t, rec = timed(lambda: SUAVE(Schema(spec), behaviour="supervised", **common).fit(
        X, y, warmup_epochs=3, head_epochs=1, finetune_epochs=1, decoder_refine_epochs=0))
    out["supervised_3_1_1_0"] = {"seconds": t, "elbo_step_rows_per_s": 32000 / t, "final_reconstruction": rec,
                                 "reference_survey": {"seconds": 7.2, "rows_per_s": 4400.0,
                                                      "note": "SURVEY.md section 6: reference PyTorch eager, 8 threads, survey container"}}
    stats = {nme: ({"mean": 50.0, "std": 12.0} if sp["type"] == "real" else {"mean_log": 1.2, "std_log": 0.45}
                   if sp["type"] == "pos" else {"offset": 1.0} if sp["type"] == "count" else {}) for nme, sp in spec.items()}
    out["cpu_baseline"] = cpu_baseline(spec, stats, 0.1, budget_s=6.0, batch=256, latent=32, hidden=(256, 128), label="M20")
    return out


def w1024_subrecord(device, steps=5, warmup=2):
    """BASELINE config #4 on the driver's clock: the W1024 training step (batch 32 768) timed like the headline one,
    GEMM-class roofline from a second, event-bracketed pass; HBM traffic per launch from the committed PMC passes."""

    import torch

    from suave_amd import lib as slib
    from suave_amd.engine import Engine, PackedTable
    from suave_amd.plan import ModelLayout
    from suave_amd.types import Schema

    spec, stats = w1024_schema()
    layout = ModelLayout(Schema(spec), K_COMP, 128, (384, 256), 0.1, "supervised", stats)
    eng = Engine(layout, device)
    eng.load_state(layout.init_params(seed=0))
    rows, B = 65536, 32768
    xnum, codes, mask = synth_table_device(layout, rows, device, seed=20201021)
    table = PackedTable(xnum, codes, mask, device)
    st = {"step": 0, "perm": None}

    def one_step():
        i = st["step"] % (rows // B)
        if i == 0:
            st["perm"] = torch.randperm(rows, device=device, dtype=torch.int32)
        st["step"] += 1
        eng.elbo(table, 0, B, row_index=st["perm"][i * B:(i + 1) * B], beta_scale=1.0, tau=1.0, grad=True, global_rows=B,
                 seed=1234, step=st["step"], accumulate_sums=True)
        eng.adam_step(1e-3)

    for _ in range(warmup):
        one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = (C.c_double * 8)()
    cnt = (C.c_int64 * 8)()
    slib.load().suave_profile_enable(1)
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    slib.load().suave_profile_enable(0)
    slib.check(slib.load().suave_profile_read(ms, cnt, 8), "suave_profile_read")
    rec = {"workload": "W1024: 65 536 rows x 1024 mixed cols (400 real / 200 pos / 100 count / 224 cat up to 256 classes / "
                       "100 ordinal, 10% masked), latent 128, K=10, hidden (384,256), dropout 0.10, fwd+bwd+Adam",
           "batch": B, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
           "rows_per_s": B * steps / elapsed,
           "kernel_ms_per_step": {PROF_CLASSES[i]: ms[i] / steps for i in range(8) if cnt[i] > 0}}
    if cnt[0] > 0:
        flops = gemm_flops_per_row(layout) * B * steps
        achieved = flops / (ms[0] * 1e-3) / 1e12
        rec["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS / 3, "unit": "TFLOP/s",
                           "frac": achieved / (PEAK_BF16_MFMA_TFLOPS / 3), "avg_launch_us": 1e3 * ms[0] / cnt[0],
                           "launches_per_step": cnt[0] // steps, "traffic": None}
        for name in ("r03_pmc_traffic_w1024.json", "r02_pmc_traffic_w1024.json"):
            f = os.path.join(ROOT, "profiles", name)
            if os.path.exists(f):
                with open(f) as fh:
                    tt = json.load(fh)
                rec["roofline"]["traffic"] = tt.get("gemm", {}).get("hbm_bytes_per_launch")
                rec["roofline"]["traffic_source"] = name
                ps = tt.get("per_step", {}).get("all_library_kernels", {})
                rec["hbm_bytes_per_step"] = 1e6 * (ps.get("read_MB", 0.0) + ps.get("write_MB", 0.0)) or None
                break
    del eng, table, xnum, codes, mask
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["s64", "w1024"], default="s64",
                    help="s64 = BASELINE config #2/#3 (headline); w1024 = config #4 (wide schema, latent 128)")
    ap.add_argument("--batch", type=int, default=0, help="rows per GPU per step (default 65536 for s64, 32768 for w1024)")
    ap.add_argument("--rows", type=int, default=0, help="table rows per GPU (default 1M at N=1, 2M at N>1)")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-subrecords", action="store_true",
                    help="skip the w1024 (config #4) and m20 (config #1) sub-records of the default S64 line")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profile-timed", action="store_true",
                    help="bracket the launches of the timed region itself with HIP events (default: a second pass of K steps)")
    ap.add_argument("--gen-rows", type=int, default=8_000_000, help="rows per GPU for the sample() measurement (0 = skip)")
    ap.add_argument("--gen-chunk", type=int, default=0, help="rows per decode_sample call in the generation measurements (0 = default)")
    ap.add_argument("--mc-chunk", type=int, default=0, help="table rows per call in the 32-draw imputation (0 = default)")
    args = ap.parse_args()

    import torch

    from suave_amd import lib as slib
    from suave_amd import parallel
    from suave_amd.engine import Engine, PackedTable
    from suave_amd.plan import ModelLayout
    from suave_amd.types import Schema

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # Rehearsal of the multi-rank path on a one-GPU box (SUAVE_BENCH_REHEARSAL=1): every rank uses cuda:0 and
    # the collectives go through gloo -- same code path, not a measurement.
    rehearsal = os.environ.get("SUAVE_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("SUAVE_BENCH_WATCHDOG", "120")), exit=True)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    wide = args.config == "w1024"
    spec, stats = w1024_schema() if wide else s64_schema()
    latent, hidden = (128, (384, 256)) if wide else (LATENT, HIDDEN)
    if not args.batch:
        args.batch = 32768 if wide else 65536
    if wide and not args.rows:
        args.rows = 131072
    layout = ModelLayout(Schema(spec), K_COMP, latent, hidden, args.dropout, "supervised", stats)
    eng = Engine(layout, device)
    eng.load_state(layout.init_params(seed=0))  # identical replicas on every rank
    rows = args.rows or (1_000_000 if world == 1 else 2_000_000)
    B = min(args.batch, rows)
    xnum, codes, mask = synth_table_device(layout, rows, device, seed=20201021 + rank)
    # This rank's block of the table's rows.  N > 1 is the product's row-sharded data path (SUAVE.fit(sharded=True),
    # suave_amd/parallel.py): the table has rows * N rows of which this rank holds [rank * rows, (rank + 1) * rows);
    # a global batch is a slice of one permutation of ALL row ids (same generator on every rank) and each rank
    # computes the rows of it that it holds, addressed -- and keyed, for every in-kernel generator -- by global row id.
    table = PackedTable(xnum, codes, mask, device, row_base=rank * rows, n_global=rows * world)

    # N > 1: the decoder-side gradient bucket is all-reduced behind the library's event while the encoder backward
    # still runs, the encoder bucket after it (suave_amd/parallel.py DataParallel, suave_elbo_backward_overlap)
    dp = parallel.DataParallel(layout, device) if dist is not None else None
    n_batches = max(1, rows // B)
    warmup_steps = max(1, 10 * n_batches)  # kl_warmup_epochs = 10 (model.py:59, :1448)
    state = {"perm": None, "step": 0}
    lr = 1e-3
    gen_perm = torch.Generator(device=device)
    gen_perm.manual_seed(4321)

    def sharded_batches():  # endless: epoch after epoch of (global batch rows, this rank's rows of the batch)
        while True:
            perm = torch.randperm(rows * world, device=device, generator=gen_perm, dtype=torch.int64)
            for b, e, loc in parallel.sharded_epoch_batches(perm, B * world, table.row_base, rows):
                if e - b == B * world:  # (full batches only: the timed steps all do the same work)
                    yield e - b, loc

    batch_iter = sharded_batches() if world > 1 else None

    def one_step():
        state["step"] += 1
        beta = 1.5 * min(state["step"], warmup_steps) / warmup_steps  # losses.kl_warmup
        if world > 1:
            nb, idx = next(batch_iter)
        else:
            i = (state["step"] - 1) % n_batches
            if i == 0:  # per-epoch permutation (model.py:1465)
                state["perm"] = torch.randperm(rows, device=device, dtype=torch.int32)
            idx = state["perm"][i * B:(i + 1) * B]
            nb = B
        eng.elbo(table, 0, int(idx.numel()), row_index=idx, beta_scale=beta, tau=1.0, grad=True, global_rows=nb,
                 seed=1234, step=state["step"], accumulate_sums=True, ready_event=dp.ready if dp else None)
        if dp is not None:  # RCCL sum of the flat gradient (each rank already scaled by 1/global_rows)
            dp.reduce_gradients(eng.grads)
        eng.adam_step(lr)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    prof = not args.no_profile
    # timed region: EXACTLY K steps, un-instrumented (HIP event brackets around every launch cost ~10 % of a step)
    if args.profile_timed and prof:
        slib.load().suave_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    ms = (C.c_double * 8)()
    cnt = (C.c_int64 * 8)()
    prof_steps = args.steps
    if prof:
        if not args.profile_timed:  # the same K steps again, every launch bracketed by HIP events on its stream
            slib.load().suave_profile_enable(1)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            fence()
            instrumented_ms = 1e3 * (time.perf_counter() - t1) / max(args.steps, 1)
        else:
            instrumented_ms = 1e3 * elapsed / max(args.steps, 1)
        slib.load().suave_profile_enable(0)
        slib.check(slib.load().suave_profile_read(ms, cnt, 8), "suave_profile_read")
    if dist is not None:
        t = torch.tensor([elapsed], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sums = eng.sums.cpu().numpy()
    if rank == 0:
        print(f"[bench] training: {1e3 * elapsed / max(args.steps, 1):.4f} ms/step", file=sys.stderr, flush=True)
    gen = None
    if args.gen_rows > 0 and not wide:
        gen = generation_rates(eng, table, rows, args.gen_rows, device, rank, world, dist, fence, prof,
                               chunk=args.gen_chunk or 1 << 20, mc_chunk=args.mc_chunk or 1 << 15)
    elif args.gen_rows > 0:  # W1024: y_full is 68 KB a row -- smaller chunks, 1M generated rows, the table's rows imputed
        gen = generation_rates(eng, table, rows, min(args.gen_rows, 1 << 20), device, rank, world, dist, fence, prof,
                               chunk=1 << 16, mc_chunk=1 << 11, n_imp_max=1 << 17)

    if rank == 0:
        steps = max(args.steps, 1)
        value = world * B * steps / elapsed
        line = {
            "metric": "ELBO train rows/sec", "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("W1024: %d rows x 1024 mixed cols (400 real/200 pos/100 count/224 cat up to 256 classes/"
                                    "100 ordinal, 10%% masked), latent 128, K=10, hidden (384,256), dropout %.2f, "
                                    "fwd+bwd+Adam" % (rows, args.dropout)) if wide else
                       ("S64: 1M rows x 64 mixed cols (13 real/13 pos/13 count/13 cat/12 ordinal, 10%% masked), "
                        "latent 32, K=10, hidden (256,128), dropout %.2f, fwd+bwd+Adam" % args.dropout)
                       if world == 1 else
                       ("S64 data-parallel: %dM rows x 64 cols row-sharded over %d GPUs, RCCL all-reduce of the flat "
                        "gradient per step" % (rows * world // 1_000_000, world)),
                       "rows_per_gpu": rows, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"dp{world}"},
            "mean_elbo_terms": {"reconstruction": sums[0] / max(sums[3], 1), "categorical_kl": sums[1] / max(sums[3], 1),
                                "gaussian_kl": sums[2] / max(sums[3], 1)},
        }
        if prof and cnt[0] > 0:
            gemm_flops = gemm_flops_per_row(layout) * B * steps
            achieved = gemm_flops / (ms[0] * 1e-3) / 1e12
            executed = gemm_flops_executed_per_row(layout) * B * steps / (ms[0] * 1e-3) / 1e12
            import ctypes as _C
            mode = _C.c_int(-1)
            slib.check(slib.load().suave_gemm_arithmetic_get(_C.byref(mode)), "suave_gemm_arithmetic_get")
            n_prod, how = ARITHMETIC[mode.value]
            peak = PEAK_FP32_MFMA_TFLOPS if n_prod == 1 else PEAK_BF16_MFMA_TFLOPS / n_prod
            line["roofline"] = {"bound": "mfma", "kernel": "gemm_kernel / gemm_pair_kernel, all %d launches/step.  fp32 in, fp32 "
                                "accumulate, fp32 out; arithmetic mode %d: %s (fp32-grade error; the data gradient of the grouped "
                                "per-component head and the narrow weight-gradient tiles stay on the exact-fp32 MFMA).  achieved = "
                                "algorithmic fp32 matrix FLOPs / GEMM time; peak = the dense MFMA peak of the instruction that runs "
                                "them / plane products per fp32 product (%.0f / %d); vs_fp32_mfma_peak prices the same rate against "
                                "the 157.3 TFLOP/s of the fp32 MFMA; matrix_pipe_frac = %d x executed FLOPs / dense 16-bit peak"
                                % (cnt[0] // steps, mode.value, how, PEAK_BF16_MFMA_TFLOPS, n_prod, n_prod),
                                "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                                "frac": achieved / peak, "traffic": None,
                                "vs_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                                "arithmetic_mode": mode.value,
                                "avg_launch_us": 1e3 * ms[0] / cnt[0],
                                "algorithmic_flops_per_launch": gemm_flops / cnt[0],
                                "executed_tflops": executed,
                                "matrix_pipe_frac": n_prod * executed / (PEAK_FP32_MFMA_TFLOPS if n_prod == 1 else PEAK_BF16_MFMA_TFLOPS)}
            # the same launches priced on the FLOPs actually executed (the algorithmic count above includes the dense
            # encoder head, of which only the selected component's rows are computed)
            line["roofline_executed"] = {"bound": "mfma", "achieved": executed, "peak": peak,
                                         "unit": "TFLOP/s", "frac": executed / peak,
                                         "kernel": "same GEMM launches; executed fp32 FLOPs against the same peak"}
            line["kernel_ms_per_step"] = {PROF_CLASSES[i]: ms[i] / steps for i in range(8) if cnt[i] > 0}
            line["kernel_timing"] = ("HIP events around every launch, on the launch stream, " +
                                     ("inside the timed region" if args.profile_timed else
                                      "over a second pass of the same K steps right after the timed region") +
                                     "; instrumented ms_per_step = %.4f" % instrumented_ms)
            # ~3.0 KB/row (S64) / ~196 KB/row (W1024) fwd+bwd for the fused likelihood+KL kernel(s)
            loss_bytes = (196.0e3 if wide else 3.0e3) * B
            if cnt[1] > 0:
                line["roofline_likelihood_kl"] = {"bound": "hbm", "achieved": loss_bytes * steps / (ms[1] * 1e-3) / 1e9,
                                                  "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                  "frac": loss_bytes * steps / (ms[1] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                                  "traffic": None}
        if gen is not None:
            line["generation"] = gen
        pmc = next((f for f in (os.path.join(ROOT, "profiles", n) for n in
                                (("r03_pmc_traffic_w1024.json", "r02_pmc_traffic_w1024.json") if wide else
                                 ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")))
                    if os.path.exists(f)), "")
        if "roofline" in line and pmc and B == (32768 if wide else 65536):  # HBM bytes per launch from the committed rocprofv3 --pmc passes
            with open(pmc) as fh:
                t = json.load(fh)
            line["roofline"]["traffic"] = t.get("gemm", {}).get("hbm_bytes_per_launch")
            line["roofline"]["traffic_source"] = t.get("source")
            if "roofline_likelihood_kl" in line:
                line["roofline_likelihood_kl"]["traffic"] = t.get("likelihood_kl", {}).get("hbm_bytes_per_launch")
        if world == 1 and not wide and not args.no_subrecords:
            print("[bench] w1024 sub-record (config #4) ...", file=sys.stderr, flush=True)
            line["w1024"] = w1024_subrecord(device)
            print("[bench] m20 sub-record (config #1: SUAVE.fit) ...", file=sys.stderr, flush=True)
            line["m20"] = m20_subrecord(device)
        if world == 1 and not args.no_cpu_baseline:
            print("[bench] cpu baseline (numpy oracle) ...", file=sys.stderr, flush=True)
            line["cpu_baseline"] = (cpu_baseline(spec, stats, args.dropout, batch=128, latent=latent, hidden=hidden, label="W1024")
                                    if wide else cpu_baseline(spec, stats, args.dropout))
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
