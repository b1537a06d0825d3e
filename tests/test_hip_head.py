"""The classifier head in the library (csrc/head.hip, include/suave_hip.h suave_head_t) against its torch statement
``supervised.ClassificationHead`` -- itself the reference's module (suave/modules/heads.py:12-91) with row-keyed
dropout: logits, class-weighted cross-entropy, parameter gradients, d CE / d latent through the joint ELBO step, and
torch.optim.Adam's update.  Tolerance 1e-5 relative (fp32 summation order); the joint step's ELBO gradients to 1e-4."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _head(C, D, dropout, weighted, seed=0):
    import torch
    from suave_amd import supervised as sup

    torch.manual_seed(seed)
    cw = (0.5 + torch.rand(C)).numpy() if weighted else None
    clf = sup.ClassificationHead(D, C, class_weight=cw, dropout=dropout).to("cuda:0")
    with torch.no_grad():
        clf.linear.weight.mul_(3.0)
        clf.linear.bias.uniform_(-0.5, 0.5)
    return clf


def _engine(D=32):
    from oracle import suave_oracle as orc
    from suave_amd.engine import Engine
    from tests.helpers import layout_from_columns, s64_columns

    cols = s64_columns(8, real_variant="scaled")
    om = orc.OracleModel(cols, n_components=3, latent_dim=D, hidden=(32, 16))
    eng = Engine(layout_from_columns(cols, K=3, D=D, hidden=(32, 16)), "cuda:0")
    eng.load_state(om.init_params(seed=1))
    return cols, om, eng


@pytest.mark.parametrize("C,D,dropout,weighted,n,N", [(2, 32, 0.0, False, 256, 256), (3, 32, 0.25, True, 1000, 5000),
                                                      (10, 8, 0.1, True, 70000, 70000), (64, 128, 0.1, True, 777, 2000),
                                                      (5, 7, 0.3, False, 1, 10)])
def test_head_step_matches_torch(C, D, dropout, weighted, n, N):
    import torch
    from suave_amd import supervised as sup

    _, _, eng = _engine()
    clf = _head(C, D, dropout, weighted)
    ref = _head(C, D, dropout, weighted)  # same seed: same parameters, kept on torch
    g = torch.Generator(device="cuda:0").manual_seed(5)
    latent = torch.randn((N, D), device="cuda:0", generator=g)
    y = torch.randint(0, C, (N,), device="cuda:0", generator=g)
    rows = torch.randperm(N, device="cuda:0", generator=g)[:n]
    hd = sup.DeviceHead(clf, y)
    assert sup.DeviceHead.supports(clf)
    clf.train(), ref.train()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    logits = torch.empty((n, C), device="cuda:0")
    for step in range(3):
        key = 1000 + step
        # gradient only, then the same batch again with the fused Adam update
        eng.head_step(hd.struct(training=True, loss_scale=1.0, loss_rows=float(n), key=key), latent, rows.to(torch.int32),
                      logits=logits)
        out = ref(latent[rows], rows=rows, key=key)
        loss = ref.loss(out, y[rows])
        opt.zero_grad()
        loss.backward()
        scale = float(out.detach().abs().max()) + 1.0
        assert float((logits - out.detach()).abs().max()) <= 1e-5 * scale
        gref = torch.cat([ref.linear.weight.grad.reshape(-1), ref.linear.bias.grad])
        assert float((hd.grads - gref).abs().max()) <= 1e-5 * float(gref.abs().max()) + 1e-8
        before = float(hd.loss_sum.item())
        eng.head_step(hd.struct(training=True, loss_scale=1.0, loss_rows=float(n), key=key, adam_lr=1e-2, want_grads=False),
                      latent, rows.to(torch.int32))
        opt.step()
        torch.cuda.synchronize()
        assert abs((float(hd.loss_sum.item()) - before) / n - float(loss.detach())) <= 1e-5 * (abs(float(loss.detach())) + 1.0)
        pref = torch.cat([ref.linear.weight.detach().reshape(-1), ref.linear.bias.detach()])
        # Adam normalises the gradient: entries whose gradient is ~0 may step either way by lr
        diff = (hd.flat - pref).abs()
        assert float((diff > 2e-5).float().mean()) <= 0.002 and float(diff.max()) <= 3 * 1e-2
        # the module's parameters ARE the flat buffer
        assert clf.linear.weight.data_ptr() == hd.flat.data_ptr() and torch.equal(clf.linear.bias, hd.flat[C * D:])
    # eval mode: no dropout; loss scale and an external weight sum scale the gradient linearly
    clf.eval(), ref.eval()
    wsum = torch.full((1,), 123.0, device="cuda:0")
    eng.head_step(hd.struct(training=False, loss_scale=0.7, loss_rows=1.0, weight_sum=wsum), latent, rows.to(torch.int32),
                  logits=logits)
    out = ref(latent[rows])
    terms = ref.loss_terms(out, y[rows]) * (0.7 / 123.0)
    ref.zero_grad()
    terms.backward()
    gref = torch.cat([ref.linear.weight.grad.reshape(-1), ref.linear.bias.grad])
    assert float((logits - out).abs().max()) <= 1e-5 * (float(out.abs().max()) + 1.0)
    assert float((hd.grads - gref).abs().max()) <= 1e-5 * float(gref.abs().max()) + 1e-9


def test_head_rejects_shapes_outside_the_kernel():
    import ctypes as C

    from suave_amd import lib as slib

    n = C.c_size_t(0)
    assert slib.load().suave_head_workspace_bytes(65, 32, C.byref(n)) != 0
    assert slib.load().suave_head_workspace_bytes(2, 257, C.byref(n)) != 0
    slib.check(slib.load().suave_head_workspace_bytes(64, 128, C.byref(n)), "suave_head_workspace_bytes")
    assert n.value > 0


@pytest.mark.parametrize("dropout,use_index", [(0.0, False), (0.2, True)])
def test_joint_step_in_one_pass_matches_forward_torch_head_backward(dropout, use_index):
    """suave_elbo_joint_backward == suave_elbo_forward -> torch head (autograd) -> suave_elbo_backward(dlatent)."""

    import torch
    from suave_amd import supervised as sup
    from suave_amd.engine import PackedTable
    from tests.helpers import synth_packed

    cols, om, eng = _engine(D=32)
    N, B = 3000, 1111
    xnum, codes, mask = synth_packed(cols, N, 0.1, seed=4)
    table = PackedTable(xnum, codes, mask, eng.device)
    clf = _head(3, 32, dropout, True)
    ref = _head(3, 32, dropout, True)
    clf.train(), ref.train()
    g = torch.Generator(device="cuda:0").manual_seed(9)
    y = torch.randint(0, 3, (N,), device="cuda:0", generator=g)
    rows = torch.randperm(N, device="cuda:0", generator=g)[:B].to(torch.int32) if use_index else None
    row_ids = rows.long() if use_index else torch.arange(100, 100 + B, device="cuda:0")
    kw = dict(row_index=rows, beta_scale=1.3, tau=1.0, seed=3, step=7) if use_index else dict(beta_scale=1.3, tau=1.0, seed=3, step=7)
    start = 0 if use_index else 100
    hd = sup.DeviceHead(clf, y)
    weight, key = 0.8, 4242
    eng.elbo(table, start, B, grad=True, global_rows=B, head=hd.struct(training=True, loss_scale=weight, loss_rows=float(B), key=key), **kw)
    torch.cuda.synchronize()
    g_joint, g_head, ce_sum = eng.grads.clone(), hd.grads.clone(), float(hd.loss_sum.item())
    fwd = eng.elbo(table, start, B, grad=False, training=True, want=("latent",), **kw)
    latent = fwd["latent"].requires_grad_(True)
    loss = ref.loss(ref(latent, rows=row_ids, key=key), y[row_ids])
    (loss * weight).backward()
    eng.elbo(table, start, B, grad=True, global_rows=B, dlatent=latent.grad.contiguous(), **kw)
    torch.cuda.synchronize()
    scale = float(eng.grads.abs().max())
    assert float((g_joint - eng.grads).abs().max()) <= 1e-5 * scale
    gref = torch.cat([ref.linear.weight.grad.reshape(-1), ref.linear.bias.grad])
    assert float((g_head - gref).abs().max()) <= 1e-5 * float(gref.abs().max()) + 1e-9
    assert abs(ce_sum / B - float(loss)) <= 1e-5 * (abs(float(loss)) + 1.0)
