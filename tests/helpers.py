"""Shared helpers for the test-suite: golden fixture loading and synthetic tables."""

from __future__ import annotations

import json
import os
from typing import Dict

import numpy as np

from oracle import suave_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str):
    path = os.path.join(GOLDEN, name + ".npz")
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def columns_from_json(text) -> list:
    cols = []
    for c in json.loads(str(text)):
        cols.append(orc.Column(c["name"], c["type"], c.get("n_classes"), c.get("y_dim"), c.get("stats") or {}))
    return cols


def fixture_model(fx) -> tuple:
    meta = json.loads(str(fx["meta"]))
    cols = columns_from_json(fx["columns"])
    om = orc.OracleModel(cols, n_components=meta["K"], latent_dim=meta["D"], hidden=tuple(meta["hidden"]),
                         dropout=meta.get("dropout", 0.0), behaviour=meta.get("behaviour", "supervised"))
    P = {k[len("param/"):]: v for k, v in fx.items() if k.startswith("param/")}
    return om, P, meta


def fixture_noise(fx, om, meta) -> Dict[str, object]:
    noise: Dict[str, object] = {"exp": fx["noise/exp"], "eps": fx["noise/eps"]}
    nh = len(om.hidden)
    if meta.get("dropout", 0.0) > 0:
        noise["enc_drop"] = [fx[f"noise/enc_drop{i}"] for i in range(nh)]
        if meta["behaviour"] == "supervised":
            noise["dec_drop"] = [fx[f"noise/dec_drop0_{i}"] for i in range(nh)]
        else:
            noise["dec_drop"] = [[fx[f"noise/dec_drop{j}_{i}"] for i in range(nh)] for j in range(om.K)]
    return noise


# ---------------------------------------------------------------------------
# synthetic tables
# ---------------------------------------------------------------------------
def s64_columns(n_cols: int = 64, real_variant: str = "std"):
    """Columns cycle [real,pos,count,cat,ordinal]; class counts cycle [2,3,4,5,8]."""

    cyc = ["real", "pos", "count", "cat", "ordinal"]
    classes = [2, 3, 4, 5, 8]
    cols = []
    n_seen = {"cat": 0, "ordinal": 0}  # each discrete type cycles through the class counts on its own
    for j in range(n_cols):
        t = cyc[j % 5]
        if t == "real":
            st = {"mean": 0.0, "std": 1.0} if real_variant == "std" else {"mean": 50.0, "std": 12.0}
            cols.append(orc.Column(f"f{j}", t, stats=st))
        elif t == "pos":
            cols.append(orc.Column(f"f{j}", t, stats={"mean_log": 1.2, "std_log": 0.45}))
        elif t == "count":
            cols.append(orc.Column(f"f{j}", t, stats={"offset": 1.0}))
        else:
            cols.append(orc.Column(f"f{j}", t, n_classes=classes[n_seen[t] % 5]))
            n_seen[t] += 1
    return cols


def w1024_columns(n_real=400, n_pos=200, n_count=100, n_cat=224, n_ord=100, seed=4):
    """categorical class counts cycle [2,4,8,16,32,64,128,256], ordinal [3,5,7,10];
    the column types are interleaved by a seeded shuffle.  Smaller counts give the same shape in miniature."""

    cat_c, ord_c = [2, 4, 8, 16, 32, 64, 128, 256], [3, 5, 7, 10]
    cols = []
    cols += [("real", None)] * n_real + [("pos", None)] * n_pos + [("count", None)] * n_count
    cols += [("cat", cat_c[i % 8]) for i in range(n_cat)] + [("ordinal", ord_c[i % 4]) for i in range(n_ord)]
    order = np.random.default_rng(seed).permutation(len(cols))
    out = []
    for j, i in enumerate(order):
        t, c = cols[i]
        if t == "real":
            out.append(orc.Column(f"f{j}", t, stats={"mean": 50.0, "std": 12.0} if j % 2 else {"mean": 0.0, "std": 1.0}))
        elif t == "pos":
            out.append(orc.Column(f"f{j}", t, stats={"mean_log": 1.2, "std_log": 0.45}))
        elif t == "count":
            out.append(orc.Column(f"f{j}", t, stats={"offset": 1.0}))
        else:
            out.append(orc.Column(f"f{j}", t, n_classes=c))
    return out


def synth_packed(cols, B, mask_rate, seed):
    rng = np.random.default_rng(seed)
    n_num = sum(c.type in ("real", "pos", "count") for c in cols)
    xnum = np.zeros((B, n_num), np.float32)
    codes = np.zeros((B, len(cols) - n_num), np.int16)
    mask = (rng.random((B, len(cols))) >= mask_rate).astype(np.uint8)
    i_num = i_disc = 0
    for j, c in enumerate(cols):
        if c.type == "real":
            xnum[:, i_num] = rng.standard_normal(B)
        elif c.type == "pos":
            raw = rng.lognormal(1.0, 0.5, B)
            xnum[:, i_num] = (np.log1p(raw) - c.stats["mean_log"]) / c.stats["std_log"]
        elif c.type == "count":
            cnt = rng.poisson(3.0, B)
            xnum[:, i_num] = np.log(cnt + c.stats["offset"] + (1.0 if c.stats["offset"] == 0 else 0.0))
        if c.type in ("real", "pos", "count"):
            xnum[mask[:, j] == 0, i_num] = 0.0
            i_num += 1
        else:
            codes[:, i_disc] = rng.integers(0, c.n_classes, B)
            codes[mask[:, j] == 0, i_disc] = -1
            i_disc += 1
    return xnum, codes, mask


def rel_err(a, b, floor=1e-6):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = max(float(np.max(np.abs(b))), floor)
    return float(np.max(np.abs(a - b))) / scale


# ---------------------------------------------------------------------------
# product-side helpers (GPU tests)
# ---------------------------------------------------------------------------
def layout_from_columns(cols, *, K, D, hidden, dropout=0.0, behaviour="supervised"):
    from suave_amd.plan import ModelLayout
    from suave_amd.types import Schema

    spec = {}
    stats = {}
    for c in cols:
        entry = {"type": c.type}
        if c.n_classes:
            entry["n_classes"] = c.n_classes
        if c.y_dim:
            entry["y_dim"] = c.y_dim
        spec[c.name] = entry
        stats[c.name] = dict(c.stats or {})
    return ModelLayout(Schema(spec), K, D, tuple(hidden), float(dropout), behaviour, stats)


def torch_noise(noise, device):
    import torch

    out = {}
    if "exp" in noise:
        out["exp_noise"] = torch.from_numpy(np.ascontiguousarray(noise["exp"], np.float32)).to(device)
    if "eps" in noise:
        out["eps"] = torch.from_numpy(np.ascontiguousarray(noise["eps"], np.float32)).to(device)
    if noise.get("enc_drop") is not None:
        out["enc_keep"] = [torch.from_numpy(np.ascontiguousarray(m, np.uint8)).to(device) for m in noise["enc_drop"]]
    if noise.get("dec_drop") is not None:
        out["dec_keep"] = [torch.from_numpy(np.ascontiguousarray(m, np.uint8)).to(device) for m in noise["dec_drop"]]
    return out


# ---------------------------------------------------------------------------
# schedule-phase fixtures (tests/golden/phases_*.npz)
# ---------------------------------------------------------------------------
def phase_record(fx, pre):
    """Draws, snapshots and per-step gradients of one recorded phase ("joint/", "refine_prior/", ...)."""

    out = {"perms": [fx[pre + f"perm/{i}"] for i in range(int(fx[pre + "n_perms"]))],
           "steps": [{"exp": fx[pre + f"exp/{i}"], "eps": fx[pre + f"eps/{i}"]} for i in range(int(fx[pre + "n_steps"]))],
           "grads": []}
    for i in range(int(fx[pre + "n_opt_steps"])):
        key = pre + f"grad/{i}/"
        out["grads"].append({k[len(key):]: v for k, v in fx.items() if k.startswith(key)})
    for when in ("before", "after"):
        key = pre + when + "/param/"
        out[when] = {k[len(key):]: v for k, v in fx.items() if k.startswith(key)}
        key = pre + when + "/clf/"
        out[when + "_clf"] = {k[len(key):]: v for k, v in fx.items() if k.startswith(key)}
    return out


def update_close(new, old, ref_new, ref_old, lr, n_steps, what=""):
    """Parameters after a few Adam steps against the reference's.  Adam normalises the gradient, so an entry whose
    gradient is within the 1e-4 gradient tolerance of zero may move by up to lr per step in either direction: at most
    0.1 % of the entries may differ by more than 0.2 % of the largest update (observed on MI355X: none, largest
    difference 4e-6 of the largest update), and none by more than the trust region 2 * lr * n_steps.  The typical
    size of the large updates must be the reference's to 1 % -- a tensor stepped with another group's learning rate
    (the joint phase has four, model.py:1894-1909) is off by the ratio of the rates."""

    d = np.asarray(new, np.float64) - np.asarray(old, np.float64)
    dr = np.asarray(ref_new, np.float64) - np.asarray(ref_old, np.float64)
    err = np.abs(d - dr)
    scale = max(float(np.abs(dr).max()), 1e-12)
    assert float(err.max()) <= 2.0 * lr * n_steps + 1e-7, (what, float(err.max()))
    assert float((err > 0.002 * scale + 1e-7).mean()) <= 0.001, (what, float((err > 0.002 * scale).mean()))
    big = np.abs(dr) > 0.25 * scale
    if big.any() and scale > 1e-7:
        ratio = float(np.median(np.abs(d[big]) / np.abs(dr[big])))
        assert abs(ratio - 1.0) <= 0.01, (what, ratio)


# ---------------------------------------------------------------------------
# persistence fixture (tests/golden/persistence_*.npz): the reference's archive stored as data
# ---------------------------------------------------------------------------
def payload_from_fixture(fx, behaviour):
    """Rebuild the nested payload `SUAVE.save` of the reference wrote (model.py:4573-4578) from the fixture's
    arrays and JSON leaves: torch tensors where it held tensors, numpy arrays where it held arrays."""

    import torch
    from collections import OrderedDict

    root = behaviour + "/payload"
    leaves = json.loads(str(fx[behaviour + "/leaves"]))
    payload: dict = {}
    for path, leaf in leaves.items():
        keys = path[len(root) + 1:].split("/")
        node = payload
        for k in keys[:-1]:
            node = node.setdefault(k, OrderedDict() if k in ("encoder", "decoder", "state_dict", "mean_state_dict") else {})
        kind = leaf["kind"]
        if kind == "tensor":
            value = torch.from_numpy(np.array(fx[path]))
        elif kind == "ndarray":
            value = np.array(fx[path])
        elif kind == "pairs":
            value = {k: v for k, v in leaf["value"]}
        elif kind == "dict" and "value" in leaf and not leaf["value"]:
            value = {}
        else:
            value = leaf.get("value")
        node[keys[-1]] = value
    return payload
