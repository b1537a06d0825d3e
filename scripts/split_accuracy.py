"""Gradient / ELBO-term error of the HIP step against the numpy oracle on the S64 schema, for the current
SUAVE_GEMM_SPLIT mode: prints the worst err / scale over all parameter tensors (the parity bar is 1e-4)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import suave_oracle as orc
from tests.helpers import layout_from_columns, s64_columns, synth_packed, torch_noise, rel_err
from suave_amd.engine import Engine, PackedTable

for B, dropout in ((4096, 0.1), (1000, 0.1), (256, 0.0)):
    cols = s64_columns(64, real_variant="std" if dropout == 0 else "scaled")
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
    ref = orc.elbo_supervised(om, P, x_enc, data, masks, noise, beta_scale=1.5, tau=1.0, training=True)
    eng = Engine(layout_from_columns(cols, K=10, D=32, hidden=(256, 128), dropout=dropout), "cuda:0")
    eng.load_state(P)
    table = PackedTable(xnum, codes, mask, eng.device)
    res = eng.elbo(table, 0, B, beta_scale=1.5, tau=1.0, grad=True, noise=torch_noise(noise, eng.device),
                   want=("reconstruction", "categorical_kl", "gaussian_kl", "latent", "component"))
    torch.cuda.synchronize()
    got = eng.grad_state()
    gmax = max(float(np.abs(v).max()) for v in ref["grads"].values() if v.size)
    worst = ("", 0.0)
    for k, r in ref["grads"].items():
        g = np.asarray(got[k]).reshape(r.shape)
        scale = max(float(np.abs(r).max()), 1e-3 * gmax, 1e-9)
        e = float(np.abs(g - r).max()) / scale
        if e > worst[1]:
            worst = (k, e)
    flips = int((res["component"].cpu().numpy().astype(np.int64) != ref["component_indices"]).sum())
    print(f"SPLIT={os.environ.get('SUAVE_GEMM_SPLIT', '4 (default)')} B={B} dropout={dropout}: worst grad err/scale {worst[1]:.2e} ({worst[0]}), "
          f"recon {rel_err(res['reconstruction'].cpu().numpy(), ref['reconstruction']):.2e}, "
          f"latent {rel_err(res['latent'].cpu().numpy(), ref['latent']):.2e}, component flips {flips}")
