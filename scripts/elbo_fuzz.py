"""Fuzz of the HIP ELBO step (forward, backward) against the numpy oracle on random schemas, model shapes, batch sizes
and both behaviours, injected noise.  usage: python scripts/elbo_fuzz.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import suave_oracle as orc
from suave_amd.engine import Engine, PackedTable
from tests.helpers import layout_from_columns, rel_err, synth_packed, torch_noise

# The parity bar of the fixtures is 1e-4.  Random heads (0.3 N(0,1) weights) produce rows the fp32 arithmetic itself
# conditions worse than that -- an ordinal probability sigmoid(a) - sigmoid(b) ~ 1e-5 carries a 1e-2 relative rounding
# error on either side, the Gumbel-softmax Jacobian of a one-row batch cancels O(100) terms -- so the fuzz asks for
# 5e-4 (2e-3 on batches below 32 rows); a wrong kernel is off by orders of magnitude more, or raises.
TOL = 5e-4


def random_columns(rng, n_cols, wide):
    cols = []
    for j in range(n_cols):
        t = rng.choice(["real", "pos", "count", "cat", "ordinal"])
        if t == "real":
            cols.append(orc.Column(f"c{j}", "real", stats={"mean": float(rng.normal()), "std": float(0.5 + rng.random() * 3)}))
        elif t == "pos":
            cols.append(orc.Column(f"c{j}", "pos", stats={"mean_log": float(rng.normal() * 0.3), "std_log": float(0.3 + rng.random())}))
        elif t == "count":
            cols.append(orc.Column(f"c{j}", "count", stats={"offset": float(rng.integers(0, 2))}))
        else:
            hi = 40 if (wide and t == "cat") else 17
            cols.append(orc.Column(f"c{j}", t, n_classes=int(rng.integers(2, hi))))
    return cols


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(n_cases):
        behaviour = str(rng.choice(["supervised", "unsupervised"]))
        n_cols = int(rng.integers(1, 30))
        K = int(rng.integers(1, 11)) if rng.random() < 0.8 else int(rng.integers(11, 33))  # (k_loss: KMAX 8 / 12 / 16 / 32)
        D = int(rng.choice([2, 3, 8, 16, 32, 33]))
        hidden = tuple(int(rng.choice([8, 24, 32, 64, 100])) for _ in range(int(rng.integers(1, 3))))
        B = int(rng.choice([1, 63, 64, 65, 256, 300, 777]))
        if behaviour == "unsupervised":
            B = min(B, 300)
        cols = random_columns(rng, n_cols, wide=bool(rng.integers(0, 2)))
        om = orc.OracleModel(cols, n_components=K, latent_dim=D, hidden=hidden, dropout=0.0, behaviour=behaviour)
        P = om.init_params(seed=int(rng.integers(1 << 30)))
        for k in P:
            if ".heads." in k or "prior" in k:
                P[k] = (0.3 * rng.standard_normal(P[k].shape)).astype(np.float32)
        xnum, codes, mask = synth_packed(cols, B, 0.15, seed=int(rng.integers(1 << 30)))
        x_enc, data, masks = orc.dense_from_packed(om, xnum, codes, mask)
        tau = float(rng.choice([1.0, 0.7]))
        beta = float(rng.choice([1.0, 1.5, 0.3]))
        if behaviour == "supervised":
            noise = {"exp": rng.exponential(size=(B, K)).astype(np.float32), "eps": rng.standard_normal((B, D)).astype(np.float32)}
            ref = orc.elbo_supervised(om, P, x_enc, data, masks, noise, beta_scale=beta, tau=tau, training=True)
        else:
            noise = {"exp": rng.exponential(size=(B, K)).astype(np.float32), "eps": rng.standard_normal((B, K, D)).astype(np.float32)}
            ref = orc.elbo_unsupervised(om, P, x_enc, data, masks, noise, beta_scale=beta, temperature=tau, training=True)
        from suave_amd import lib as slib
        small = int(rng.choice([8192, 0]))   # small-batch split-K kernels, or the 128 x 128 tile kernels
        if os.environ.get("FUZZ_SMALL"):
            small = int(os.environ["FUZZ_SMALL"])
        slib.check(slib.load().suave_small_batch_rows_set(small), "suave_small_batch_rows_set")
        print(f"case {case}: {behaviour[:5]} cols={n_cols} K={K} D={D} hidden={hidden} B={B} tau={tau} small={small} ...", flush=True)
        eng = Engine(layout_from_columns(cols, K=K, D=D, hidden=hidden, dropout=0.0, behaviour=behaviour), "cuda:0")
        eng.load_state(P)
        table = PackedTable(xnum, codes, mask, eng.device)
        res = eng.elbo(table, 0, B, beta_scale=beta, tau=tau, grad=True, noise=torch_noise(noise, eng.device),
                       want=("reconstruction", "categorical_kl", "gaussian_kl", "latent"))
        torch.cuda.synchronize()
        slib.check(slib.load().suave_small_batch_rows_set(8192), "suave_small_batch_rows_set")
        # (a KL term near zero is a cancellation of O(1) logarithms: 1e-4 relative + 1e-6 absolute, as np.allclose)
        errs = {k: float(np.abs(res[k].cpu().numpy() - ref[k]).max()) / (float(np.abs(ref[k]).max()) + 1e-2)
                for k in ("reconstruction", "categorical_kl", "gaussian_kl", "latent")}
        got = eng.grad_state()
        gmax = max(float(np.abs(v).max()) for v in ref["grads"].values() if v.size)
        gerr, gwhere = 0.0, ""
        for k, r in ref["grads"].items():
            g = np.asarray(got[k]).reshape(r.shape)
            if not r.size:
                continue
            scale = max(float(np.abs(r).max()), 1e-3 * gmax, 1e-9)
            e = float(np.abs(g - r).max()) / scale
            if not np.isfinite(e) or e > gerr:
                gerr, gwhere = e, k
        tol = TOL if B >= 32 else 2e-3
        worst = max(list(errs.values()) + [gerr])
        status = "ok" if np.isfinite(worst) and worst < tol else "MISMATCH"
        if status != "ok" and max(errs.values()) < tol and np.isfinite(gerr) and gerr < 3e-2:
            # A hidden unit whose pre-activation is ~0 in some row takes relu' = 0 or 1 depending on the last bits of the
            # matrix product: the default arithmetic (fp16 planes, ~1e-6) and fp32 can then differ by that row's share
            # of a gradient.  Such a case must agree with the oracle on the exact-fp32 arithmetic.
            slib.check(slib.load().suave_gemm_arithmetic_set(0), "suave_gemm_arithmetic_set")
            eng.elbo(table, 0, B, beta_scale=beta, tau=tau, grad=True, noise=torch_noise(noise, eng.device))
            torch.cuda.synchronize()
            slib.check(slib.load().suave_gemm_arithmetic_set(4), "suave_gemm_arithmetic_set")
            got0 = eng.grad_state()
            g0 = max(float(np.abs(np.asarray(got0[k]).reshape(r.shape) - r).max()) / max(float(np.abs(r).max()), 1e-3 * gmax, 1e-9)
                     for k, r in ref["grads"].items() if r.size)
            if g0 < tol:
                status = "ok (relu boundary: exact arithmetic %.1e)" % g0
        if not status.startswith("ok") and max(errs.values()) < tol and np.isfinite(gerr) and gerr < 3e-2:
            # ... or the pre-activation is zero to fp32 rounding itself (|pre| < 1e-5 against O(1) terms): then exact
            # fp32 on the GPU and the oracle's float64 may disagree on relu' too (seed 31, case 277)
            def margin_of(prefix, h):
                i, m = 0, np.inf
                while f"{prefix}.{i}.weight" in P:
                    pre = h @ P[f"{prefix}.{i}.weight"].astype(np.float64).T + P[f"{prefix}.{i}.bias"]
                    m = min(m, float(np.abs(pre).min()))
                    h, i = np.maximum(pre, 0.0), i + 2
                return m
            if behaviour == "supervised":
                z = ref["latent"].astype(np.float64)
            else:  # every component's latent goes through the decoder
                z = (ref["component_mu"] + noise["eps"] * np.exp(0.5 * ref["component_logvar"])).reshape(B * K, D).astype(np.float64)
            m_enc, m_dec = margin_of("encoder.backbone", x_enc.astype(np.float64)), margin_of("decoder.backbone", z)
            # (a flipped decoder unit changes dz and with it every encoder gradient; a flipped encoder unit only those)
            if m_dec < 1e-5 or (m_enc < 1e-5 and gwhere.startswith("encoder.backbone")):
                status = "ok (relu boundary: |pre-activation| encoder %.1e decoder %.1e)" % (m_enc, m_dec)
        bad += not status.startswith("ok")
        print(f"case {case}: {behaviour[:5]} cols={n_cols} K={K} D={D} hidden={hidden} B={B} tau={tau} small={small} "
              f"terms={max(errs.values()):.1e} grad={gerr:.1e} ({gwhere}) {status}", flush=True)
        if not status.startswith("ok") and gwhere:
            r = ref["grads"][gwhere]
            g = np.asarray(got[gwhere]).reshape(r.shape)
            np.set_printoptions(precision=5, linewidth=200, suppress=True)
            for k2, r2 in ref["grads"].items():
                g2 = np.asarray(got[k2]).reshape(r2.shape)
                if r2.size and float(np.abs(g2 - r2).max()) > 1e-4 * max(float(np.abs(r2).max()), 1e-3 * gmax):
                    print("    off:", k2, r2.shape, "max err", float(np.abs(g2 - r2).max()), "at", np.unravel_index(np.abs(g2 - r2).argmax(), r2.shape))
            print("    got\n", g[-4:], "\n    ref\n", r[-4:], "\n    worst at", np.unravel_index(np.abs(g - r).argmax(), r.shape), flush=True)
        if not status.startswith("ok"):
            print("   ", errs, {k: (float(np.abs(ref[k]).max()), float(np.abs(res[k].cpu().numpy() - ref[k]).max())) for k in errs}, flush=True)
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
