"""Archive interchange with the reference (`SUAVE.save` / `SUAVE.load`,
suave/model.py:4456-4579, :4581-4950).

`tests/golden/persistence_mixed7.npz` holds the archive the reference's own `save` wrote for a supervised and an
unsupervised model, as data (every tensor + the JSON of every other leaf).  Its generator also asserted, in the
build container, that the reference's `SUAVE.load` reads an archive assembled by `suave_amd.persist.build_payload`
into identical modules (`ours_loaded_by_reference`).  Here: the build reads the reference's archive, and writes
the reference's key tree.
"""

import json
import os

import numpy as np
import pytest

from tests.helpers import load_golden, payload_from_fixture, rel_err

BEHAVIOURS = ["supervised", "unsupervised"]
# leaves whose Python type may differ between the two writers; the reference's loader normalises each on reading:
# classes / train_target_indices via np.asarray (model.py:4780-4783, 4825-4828)
LOOSE = {"artefacts/classes", "artefacts/train_target_indices"}


@pytest.mark.parametrize("behaviour", BEHAVIOURS)
def test_reference_archive_parameters_are_read_by_name(behaviour):
    from suave_amd import persist

    fx = load_golden("persistence_mixed7")
    assert bool(fx[behaviour + "/ours_loaded_by_reference"])
    payload = payload_from_fixture(fx, behaviour)
    assert tuple(payload) == ("metadata", "modules", "artefacts")
    assert set(payload["metadata"]) == set(persist.METADATA_KEYS)
    assert set(payload["modules"]) == set(persist.MODULE_KEYS)
    assert set(payload["artefacts"]) == set(persist.ARTEFACT_KEYS)
    state = persist.state_from_payload(payload)
    want = {k[len(behaviour) + 7:]: v for k, v in fx.items() if k.startswith(behaviour + "/param/")}
    assert set(state) == set(want)
    for k, v in want.items():
        assert np.array_equal(state[k], v), k


@pytest.mark.parametrize("behaviour", BEHAVIOURS)
def test_build_payload_emits_the_reference_key_tree(behaviour):
    import torch
    from suave_amd import persist
    from suave_amd.types import Schema

    fx = load_golden("persistence_mixed7")
    ref = payload_from_fixture(fx, behaviour)
    art, mods = ref["artefacts"], ref["modules"]
    state = persist.state_from_payload(ref)
    clf = mods["classifier"]
    cache = {"mean": art["train_latent_mu"], "post_logvar": art["train_latent_logvar"],
             "logits": art["train_component_logits"], "mu": art["train_component_mu"],
             "logvar": art["train_component_logvar"], "probs": art["train_component_probs"]}
    ours = persist.build_payload(
        metadata=dict(ref["metadata"]), state=state, schema=Schema(ref["metadata"]["schema"]),
        classifier_state=None if clf is None else dict(clf["state_dict"]),
        class_weight=None if clf is None else clf["class_weight"], use_weight=bool(clf and clf["use_weight"]),
        temperature_scaler=mods["temperature_scaler"], norm_stats=art["normalization"], classes=art["classes"],
        class_to_index=art["class_to_index"], is_calibrated=art["is_calibrated"], train_cache=cache,
        train_target_indices=art["train_target_indices"], warmup_val_history=art["warmup_val_history"],
        joint_val_metrics=art["joint_val_metrics"])
    mine, theirs = persist.tree(ours), json.loads(str(fx[behaviour + "/tree"]))
    assert set(mine) == set(theirs), set(mine) ^ set(theirs)
    for path, kind in theirs.items():
        if path in LOOSE:
            continue
        assert mine[path] == kind, (path, mine[path], kind)  # tensors: same dtype and shape; leaves: same Python type
    # same values too
    for mod in ("encoder", "decoder"):
        for k, v in mods[mod].items():
            assert torch.equal(ours["modules"][mod][k], v), (mod, k)
    for k, v in mods["prior"].items():
        if isinstance(v, torch.Tensor):
            assert rel_err(ours["modules"]["prior"][k].numpy(), v.numpy()) < 1e-6, k
    assert ours["metadata"]["schema"] == ref["metadata"]["schema"]
    assert ours["artefacts"]["normalization"] == art["normalization"]
    assert ours["artefacts"]["feature_layout"] == art["feature_layout"]


@pytest.mark.parametrize("behaviour", BEHAVIOURS)
def test_archive_is_read_without_unpickling_code(behaviour, tmp_path):
    """An archive with the reference's numpy leaves (`classes`, `train_target_indices`) loads under
    `weights_only=True` through the allow-list of numpy's array types."""

    import torch
    from suave_amd import persist

    fx = load_golden("persistence_mixed7")
    ref = payload_from_fixture(fx, behaviour)
    path = os.path.join(tmp_path, "ref_layout.pt")
    torch.save(ref, path)
    back = persist.load_archive(path)
    assert persist.tree(back) == json.loads(str(fx[behaviour + "/tree"]))
    with open(path, "wb") as fh:
        fh.write(b"not an archive")
    with pytest.raises(ValueError, match="Unexpected model archive format"):
        persist.load_archive(path)


@pytest.mark.gpu
@pytest.mark.parametrize("behaviour", BEHAVIOURS)
def test_load_reference_archive_and_save_it_back(behaviour, tmp_path):
    """`suave_amd.SUAVE.load` on the reference's archive -> a working model on the GPU; `save` -> the same key
    tree and tensors again."""

    import torch
    from suave_amd import persist
    from suave_amd.model import SUAVE

    fx = load_golden("persistence_mixed7")
    ref = payload_from_fixture(fx, behaviour)
    path = os.path.join(tmp_path, "ref_layout.pt")
    torch.save(ref, path)
    m = SUAVE.load(path, device="cuda:0")
    want = {k[len(behaviour) + 7:]: v for k, v in fx.items() if k.startswith(behaviour + "/param/")}
    got = m._engine.state()
    for k, v in want.items():
        assert np.array_equal(got[k].reshape(v.shape), v), k
    meta = ref["metadata"]
    assert m.hidden_dims == tuple(meta["hidden_dims"]) and m.dropout == meta["dropout"] and m._inference_tau == 0.85
    assert m._norm_stats_per_col == ref["artefacts"]["normalization"]
    frame = m.sample(32) if behaviour == "unsupervised" else m.sample(32, conditional=True, y=np.asarray([10, 20] * 16))
    assert list(frame.columns) == list(m.schema.feature_names) and len(frame) == 32 and not frame.isna().any().any()
    if behaviour == "supervised":
        assert m._is_calibrated and abs(m._temperature_scaler.temperature - 1.7) < 1e-12
        assert list(m._classes) == [10, 20, 30]
        clf = ref["modules"]["classifier"]["state_dict"]
        for k, v in m._classifier.state_dict().items():
            assert torch.equal(v.cpu(), clf[k]), k
    out = os.path.join(tmp_path, "ours.pt")
    assert str(m.save(out)) == out
    again = persist.load_archive(out)
    mine, theirs = persist.tree(again), json.loads(str(fx[behaviour + "/tree"]))
    assert set(mine) == set(theirs)
    for p_, kind in theirs.items():
        if p_ not in LOOSE:
            assert mine[p_] == kind, (p_, mine[p_], kind)
    for mod in ("encoder", "decoder"):
        for k, v in ref["modules"][mod].items():
            assert torch.equal(again["modules"][mod][k], v), (mod, k)
    m2 = SUAVE.load(out, device="cuda:0")
    for k, v in want.items():
        assert np.array_equal(m2._engine.state()[k].reshape(v.shape), v), k
