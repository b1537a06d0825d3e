"""CPU emulation of the split-precision GEMM arithmetics on the REAL tensors of an S64 ELBO step: every `@` of the
numpy oracle (forward, data gradients, weight gradients) is replaced by the plane products a candidate MFMA path
would run, and the resulting gradients are compared with the plain fp32 oracle the parity tests use (bar: 1e-4 of
the tensor's scale).  Schemes:

  bf16x3   three bf16 planes (truncation), six products          -- the library's default path
  bf16x2   two bf16 planes, three products                       -- rejected in round 1 (5e-2 on std = 12 columns)
  fp16x2s  fp16 hi + fp16 (residual * 2^11) lo, three products, the two cross products in a second accumulator
           scaled by 2^-11 at the end                            -- candidate: half the MFMAs of bf16x3

Run on the CPU (no GPU, no library): python scripts/split_emulate.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import suave_oracle as orc  # noqa: E402
from tests.helpers import s64_columns, synth_packed  # noqa: E402

SCHEME = "fp32"


def bf16_planes(x, n):
    out = []
    r = x.astype(np.float32)
    for _ in range(n):
        h = (r.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        out.append(h.astype(np.float64))
        r = (r - h).astype(np.float32)
    return out


def fp16_planes(x):
    x = x.astype(np.float32)
    hi = np.clip(x, -65504.0, 65504.0).astype(np.float16).astype(np.float32)
    r = (x - hi).astype(np.float32)
    lo = (r * np.float32(2048.0)).astype(np.float16).astype(np.float32)
    return hi.astype(np.float64), lo.astype(np.float64)


def split_matmul(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    if SCHEME == "fp32" or a.ndim != 2 or b.ndim != 2 or min(a.shape + b.shape) < 16:
        return a @ b  # (likelihood heads: plain fp32 FMAs in the fused kernel, not on the MFMA path)
    if SCHEME in ("bf16x3", "bf16x2"):
        n = 3 if SCHEME == "bf16x3" else 2
        pa, pb = bf16_planes(a, n), bf16_planes(b, n)
        acc = np.zeros((a.shape[0], b.shape[1]), dtype=np.float64)
        for i in range(n):
            for j in range(n):
                if i + j <= n - 1:
                    acc += pa[i] @ pb[j]
        return acc.astype(np.float32)
    if SCHEME == "fp16x2s":
        a0, a1 = fp16_planes(a)
        b0, b1 = fp16_planes(b)
        main = (a0 @ b0).astype(np.float32)  # fp32 accumulators
        cross = (a0 @ b1 + a1 @ b0).astype(np.float32)
        return (main + cross * np.float32(1.0 / 2048.0)).astype(np.float32)
    raise ValueError(SCHEME)


class SplitArr(np.ndarray):
    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        raw = [np.asarray(i) if isinstance(i, np.ndarray) else i for i in inputs]
        if ufunc is np.matmul and method == "__call__" and out is None:
            res = split_matmul(raw[0], raw[1])
        else:
            if out is not None:
                kwargs["out"] = tuple(np.asarray(o) for o in out)
            res = getattr(ufunc, method)(*raw, **kwargs)
        if isinstance(res, np.ndarray):
            return res.view(SplitArr)
        if isinstance(res, tuple):
            return tuple(r.view(SplitArr) if isinstance(r, np.ndarray) else r for r in res)
        return res


def run(B, dropout, variant):
    global SCHEME
    cols = s64_columns(64, real_variant=variant)
    om = orc.OracleModel(cols, n_components=10, latent_dim=32, hidden=(256, 128), dropout=dropout)
    P = om.init_params(seed=3)
    rng = np.random.default_rng(5)
    for k in P:
        if ".heads." in k:
            P[k] = (0.3 * rng.standard_normal(P[k].shape)).astype(np.float32)
    xnum, codes, mask = synth_packed(cols, B, 0.1, seed=20201021)
    x_enc, data, masks = orc.dense_from_packed(om, xnum, codes, mask)
    noise = {"exp": rng.exponential(size=(B, 10)).astype(np.float32), "eps": rng.standard_normal((B, 32)).astype(np.float32)}
    if dropout > 0:
        noise["enc_drop"] = [(rng.random((B, h)) >= dropout).astype(np.uint8) for h in (256, 128)]
        noise["dec_drop"] = [(rng.random((B, h)) >= dropout).astype(np.uint8) for h in (256, 128)]
    SCHEME = "fp32"
    ref = orc.elbo_supervised(om, P, x_enc, data, masks, noise, beta_scale=1.5, tau=1.0, training=True)
    Ps = {k: v.view(SplitArr) for k, v in P.items()}
    gmax = max(float(np.abs(v).max()) for v in ref["grads"].values() if v.size)
    for scheme in ("bf16x3", "bf16x2", "fp16x2s"):
        SCHEME = scheme
        got = orc.elbo_supervised(om, Ps, x_enc.view(SplitArr), data, masks, noise, beta_scale=1.5, tau=1.0, training=True)
        worst = ("", 0.0)
        for k, r in ref["grads"].items():
            g = np.asarray(got["grads"][k]).reshape(r.shape)
            scale = max(float(np.abs(r).max()), 1e-3 * gmax, 1e-9)
            e = float(np.abs(g - np.asarray(r)).max()) / scale
            if e > worst[1]:
                worst = (k, e)
        rec = float(np.abs(np.asarray(got["reconstruction"]) - ref["reconstruction"]).max() /
                    max(1e-6, float(np.abs(ref["reconstruction"]).max())))
        flips = int((np.asarray(got["component_indices"]) != ref["component_indices"]).sum())
        print(f"{scheme:8s} B={B} dropout={dropout} {variant}: worst grad err/scale {worst[1]:.2e} ({worst[0]}), "
              f"recon {rec:.2e}, component flips {flips}", flush=True)


if __name__ == "__main__":
    run(256, 0.0, "std")
    run(1000, 0.1, "scaled")
    run(4096, 0.1, "scaled")
