"""Host data layer against the reference: norm statistics and the packed-table writer reproduce
standardize + _apply_training_normalization + _prepare_training_tensors (golden: tensorise_mixed8)."""

import json

import numpy as np
import pandas as pd

from oracle import suave_oracle as orc
from suave_amd import data as data_utils
from suave_amd.types import Schema
from tests.helpers import load_golden


def test_pack_frame_matches_reference_encoder_inputs():
    fx = load_golden("tensorise_mixed8")
    schema = Schema(json.loads(str(fx["schema"])))
    X = pd.DataFrame({k[len("raw/"):]: fx[k] for k in fx if k.startswith("raw/")})[list(schema.feature_names)]
    stats = data_utils.fit_norm_stats(X, schema)
    ref_stats = json.loads(str(fx["stats"]))
    for col, st in ref_stats.items():
        for key, val in st.items():
            if isinstance(val, float):
                assert abs(stats[col][key] - val) <= 1e-9 * max(1.0, abs(val)), (col, key)
    xnum, codes, mask = data_utils.pack_frame(X, schema, stats)
    cols = [orc.Column(n, schema[n].type, schema[n].n_classes, schema[n].y_dim, stats[n]) for n in schema.feature_names]
    om = orc.OracleModel(cols, n_components=2, latent_dim=4, hidden=(8,))
    x_enc, _, _ = orc.dense_from_packed(om, xnum, codes, mask)
    np.testing.assert_allclose(x_enc, fx["enc_inputs"], rtol=1e-6, atol=1e-6)


def test_pack_frame_edge_cases():
    schema = Schema({"r": {"type": "real"}, "c": {"type": "cat", "n_classes": 2}, "o": {"type": "ordinal", "n_classes": 3}})
    X = pd.DataFrame({"r": [1.0, np.nan], "c": ["a", "b"], "o": [0, 2]})
    stats = data_utils.fit_norm_stats(X, schema)
    # unseen category and empty frame
    Y = pd.DataFrame({"r": [2.0], "c": ["zzz"], "o": [1]})
    xnum, codes, mask = data_utils.pack_frame(Y, schema, stats)
    assert codes[0, 0] == -1 and mask[0, 1] == 0 and codes[0, 1] == 1
    xnum, codes, mask = data_utils.pack_frame(X.iloc[:0], schema, stats)
    assert xnum.shape == (0, 1) and codes.shape == (0, 2) and mask.shape == (0, 3)
    frame = data_utils.frame_from_packed(np.array([[0.5]], np.float32), np.array([[1, 2]], np.int16), schema, stats)
    assert frame["c"].iloc[0] == "b" and frame["o"].iloc[0] == 2 and frame["o"].cat.ordered
