"""Fuzz of the fused decode kernel against the per-layer path on random schemas / model shapes: the same rows through
component indices (k_decode_fused) and through one-hot dense assignments (GEMMs + k_sample_heads), same in-kernel
generator keys -- discrete cells must agree bit for bit, numeric ones to 2e-5.  usage: python scripts/decode_fuzz.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import suave_oracle as orc
from suave_amd.engine import Engine
from tests.helpers import layout_from_columns


def random_columns(rng, n_cols):
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
            cols.append(orc.Column(f"c{j}", t, n_classes=int(rng.integers(2, 17))))
    return cols


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(n_cases):
        n_cols = int(rng.integers(1, 48))
        K = int(rng.integers(1, 13))
        D = int(rng.choice([4, 8, 16, 32, 48, 64]))
        hidden = (int(rng.choice([32, 64, 96, 128, 256])), int(rng.choice([32, 64, 128])))
        cols = random_columns(rng, n_cols)
        om = orc.OracleModel(cols, n_components=K, latent_dim=D, hidden=hidden)
        P = om.init_params(seed=int(rng.integers(1 << 30)))
        for k in P:
            if ".heads." in k:
                P[k] = (0.5 * rng.standard_normal(P[k].shape)).astype(np.float32)
        eng = Engine(layout_from_columns(cols, K=K, D=D, hidden=hidden), "cuda:0")
        eng.load_state(P)
        n = int(rng.choice([1, 31, 128, 129, 1000, 4097]))
        z = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32)).cuda()
        comp = torch.from_numpy(rng.integers(0, K, n).astype(np.int32)).cuda()
        onehot = torch.eye(K, device="cuda")[comp.long()].contiguous()
        seed, off = int(rng.integers(1 << 30)), int(rng.integers(1 << 20))
        a_num, a_codes, a_y = eng.decode_sample(z, component=comp, seed=seed, row_offset=off, want_y=True)
        b_num, b_codes, b_y = eng.decode_sample(z, assignments=onehot, seed=seed, row_offset=off, want_y=True)
        torch.cuda.synchronize()
        ok_codes = torch.equal(a_codes, b_codes)
        num_a, num_b = a_num.double(), b_num.double()
        is_count = torch.tensor([c.type == "count" for c in cols if c.type in ("real", "pos", "count")], device="cuda")
        rel = ((num_a - num_b).abs() / (1.0 + num_b.abs())).max().item() if num_a.numel() else 0.0
        cnt_ok = bool((num_a[:, is_count] == num_b[:, is_count]).all()) if num_a.numel() and is_count.any() else True
        yerr = ((a_y - b_y).abs().max() / (1e-6 + b_y.abs().max())).item() if a_y.numel() else 0.0
        status = "ok" if ok_codes and cnt_ok and rel < 2e-4 and yerr < 1e-4 else "MISMATCH"
        bad += status != "ok"
        print(f"case {case}: cols={n_cols} K={K} D={D} hidden={hidden} n={n} codes={'=' if ok_codes else '!'} counts={'=' if cnt_ok else '!'} "
              f"num_rel={rel:.1e} y_rel={yerr:.1e} {status}", flush=True)
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
