import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from oracle import suave_oracle as orc
from suave_amd.engine import Engine, PackedTable
from tests.helpers import layout_from_columns, rel_err, s64_columns, synth_packed, torch_noise
dev = torch.device("cuda:0")
cols = s64_columns(20)
K, D, hidden, B = 4, 8, (32, 16), 200
om = orc.OracleModel(cols, n_components=K, latent_dim=D, hidden=hidden, dropout=0.0)
P = om.init_params(seed=1)
xnum, codes, mask = synth_packed(cols, B, 0.1, seed=3)
x_enc, data, masks = orc.dense_from_packed(om, xnum, codes, mask)
rng = np.random.default_rng(0)
noise = {"exp": rng.exponential(size=(B, K)).astype(np.float32), "eps": rng.standard_normal((B, D)).astype(np.float32)}
ref = orc.elbo_supervised(om, P, x_enc, data, masks, noise, beta_scale=1.0, tau=1.0)
P64 = {k: v.astype(np.float64) for k, v in P.items()}
n64 = {k: v.astype(np.float64) for k, v in noise.items()}
d64 = {k: v.astype(np.float64) for k, v in data.items()}
m64 = {k: v.astype(np.float64) for k, v in masks.items()}
ref64 = orc.elbo_supervised(om, P64, x_enc.astype(np.float64), d64, m64, n64, beta_scale=1.0, tau=1.0)
eng = Engine(layout_from_columns(cols, K=K, D=D, hidden=hidden), dev)
eng.load_state(P)
table = PackedTable(xnum, codes, mask, dev)
res = eng.elbo(table, 0, B, beta_scale=1.0, grad=True, noise=torch_noise(noise, dev), want=("reconstruction","component"))
torch.cuda.synchronize()
got = eng.grad_state()
gmax = max(float(np.abs(v).max()) for v in ref["grads"].values())
print("kmatch", np.array_equal(res["component"].cpu().numpy(), ref["component_indices"]), "gmax", gmax)
for k, v in ref["grads"].items():
    e_gpu = float(np.abs(got[k].reshape(v.shape) - ref64["grads"][k]).max())
    e_orc = float(np.abs(v - ref64["grads"][k]).max())
    s = float(np.abs(v).max())
    if e_gpu > 3e-5 * max(s, 1e-3*gmax):
        print(k, "scale", s, "gpu-vs-f64", e_gpu, "oracle32-vs-f64", e_orc)
