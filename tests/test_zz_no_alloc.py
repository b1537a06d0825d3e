"""suave_decode_sample_static -- the entry point that allocates nothing -- against suave_decode_sample.

Both run the same kernels, so every output (sampled values, codes, y_full) must be bit-identical.  Two calls in
a row share the library's static scratch, the second with a smaller input; a request beyond the buffer fails
with a clear error.  Collected last (file name): it must pass alone in a fresh process and after the suite.
"""

import numpy as np
import pytest

from oracle import suave_oracle as orc
from tests.helpers import layout_from_columns, s64_columns

pytestmark = pytest.mark.gpu


def _run_pair(eng, n, seed, rng, dense_s):
    import torch

    dev = eng.device
    L = eng.layout
    K, D = L.n_components, L.latent_dim
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    z = t(rng.standard_normal((n, D)).astype(np.float32))
    comp = t(rng.integers(0, K, n).astype(np.int32))
    s = t(rng.dirichlet(np.ones(K), n).astype(np.float32)) if dense_s else None
    kw = dict(component=None if dense_s else comp, assignments=s, seed=seed, row_offset=1000)
    ref_num, ref_codes, _ = eng.decode_sample(z, want_y=True, **kw)
    y_ref = torch.empty((n, L.y_rows), dtype=torch.float32, device=dev)
    # the product call again for the raw y buffer (decode_sample maps wide-column slots back to schema order)
    out_num = torch.full((n, max(L.n_num, 1)), float("nan"), dtype=torch.float32, device=dev)
    out_codes = torch.full((n, max(L.n_disc, 1)), -7, dtype=torch.int16, device=dev)
    y_out = torch.full((n, L.y_rows), float("nan"), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated(dev)
    eng.decode_sample_static(z, out_num, out_codes, y_out=y_out, **kw)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated(dev) == before  # no tensor was created on the way
    assert torch.equal(out_num[:, : L.n_num], ref_num)
    assert torch.equal(out_codes[:, : L.n_disc], ref_codes)
    assert bool(torch.isfinite(y_out).all())
    # y_full of the existing entry point, raw layout
    import ctypes as C
    ws = eng.workspace(n)
    base = (ws.data_ptr() + 255) // 256 * 256
    o2 = torch.empty_like(out_num)
    c2 = torch.empty_like(out_codes)
    rc = eng.lib.suave_decode_sample(eng._plan, eng.params.data_ptr(), z.data_ptr(), None if s is None else s.data_ptr(),
                                     None if dense_s else comp.data_ptr(), n, None, None, None, seed, 1000, base,
                                     ws.numel() - (base - ws.data_ptr()), eng._stream(), o2.data_ptr(), int(o2.shape[1]),
                                     c2.data_ptr(), int(c2.shape[1]), y_ref.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(y_out, y_ref)
    del C


def test_static_decode_matches_workspace_decode_twice_then_smaller():
    cols = s64_columns(64, real_variant="scaled")
    om = orc.OracleModel(cols, n_components=10, latent_dim=32, hidden=(256, 128))
    P = om.init_params(seed=11)
    from suave_amd.engine import Engine

    eng = Engine(layout_from_columns(cols, K=10, D=32, hidden=(256, 128)), "cuda:0")
    eng.load_state(P)
    rng = np.random.default_rng(12)
    _run_pair(eng, 9000, 5, rng, dense_s=False)   # fills most of the static buffer
    _run_pair(eng, 700, 6, rng, dense_s=True)     # smaller input over the previous call's leftovers
    _run_pair(eng, 1, 7, rng, dense_s=False)


def test_static_decode_rejects_oversized_requests_and_accepts_empty():
    import torch
    from suave_amd import lib as slib
    from suave_amd.engine import Engine

    cols = s64_columns(64)
    om = orc.OracleModel(cols, n_components=10, latent_dim=32, hidden=(256, 128))
    eng = Engine(layout_from_columns(cols, K=10, D=32, hidden=(256, 128)), "cuda:0")
    eng.load_state(om.init_params(seed=1))
    L = eng.layout
    n = 1 << 18  # far beyond what the 32 MiB buffer holds for this model
    z = torch.zeros((n, 32), device=eng.device)
    comp = torch.zeros((n,), dtype=torch.int32, device=eng.device)
    out_num = torch.empty((n, L.n_num), device=eng.device)
    out_codes = torch.empty((n, L.n_disc), dtype=torch.int16, device=eng.device)
    # dense (soft) assignments take the per-layer path, whose activations and y_full live in the static buffer
    s = torch.full((n, 10), 0.1, device=eng.device)
    with pytest.raises(slib.SuaveHipError, match="static buffer holds"):
        eng.decode_sample_static(z, out_num, out_codes, assignments=s)
    import ctypes as C

    fit = C.c_int64(-1)
    slib.check(eng.lib.suave_decode_static_max_rows(eng._plan, 1, C.byref(fit)), "suave_decode_static_max_rows")
    assert 14000 <= fit.value < n  # (include/suave_hip.h: 14 979 for this model)
    eng.decode_sample_static(z[:fit.value], out_num, out_codes, assignments=s[:fit.value])  # exactly the limit
    with pytest.raises(slib.SuaveHipError, match=f"\\({fit.value} rows of this model"):
        eng.decode_sample_static(z[:fit.value + 1], out_num, out_codes, assignments=s[:fit.value + 1])
    slib.check(eng.lib.suave_decode_static_max_rows(eng._plan, 0, C.byref(fit)), "suave_decode_static_max_rows")
    assert fit.value >= n
    # component indices take the fused kernel: the buffer only holds the weight images, any row count goes
    eng.decode_sample_static(z, out_num, out_codes, component=comp)
    eng.decode_sample_static(z[:0], out_num, out_codes, component=comp)      # empty request
    torch.cuda.synchronize()
