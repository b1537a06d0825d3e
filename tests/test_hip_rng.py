"""The library's word -> variate maps (csrc/common.h: u01, exp_draw, normal2) at the extreme Philox words, and an
M20-shaped unsupervised fit that once drew a uniform of exactly 1 (Exp(1) draw 0, Gumbel +inf, NaN from there on)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _draws(words):
    import torch
    from suave_amd import lib as _lib

    lib = _lib.load()
    w = torch.tensor(np.asarray(words, np.uint32).view(np.int32), device="cuda:0")
    u, e, n = (torch.empty(w.numel(), device="cuda:0") for _ in range(3))
    _lib.check(lib.suave_rng_draws(w.data_ptr(), w.numel(), torch.cuda.current_stream().cuda_stream, u.data_ptr(), e.data_ptr(),
                                   n.data_ptr()), "suave_rng_draws")
    torch.cuda.synchronize()
    return u.cpu().numpy(), e.cpu().numpy(), n.cpu().numpy()


def test_variates_stay_finite_at_the_extreme_words():
    edge = [0, 1, 0xFF, 0x100, 0x1FF, 0x200, 0x7FFFFFFF, 0x80000000, 0xFFFFFDFF, 0xFFFFFE00, 0xFFFFFEFF, 0xFFFFFF00, 0xFFFFFFFF]
    words = np.array([a for a in edge for _ in (0, 1)] + [a for a in edge for b in edge for a in (a, b)], np.uint32)
    u, e, n = _draws(words)
    assert np.all(u > 0) and np.all(u < 1)
    assert np.all(np.isfinite(e)) and np.all(e > 0)
    assert np.all(np.isfinite(np.log(e)))  # the Gumbel draw -log(e)
    assert np.all(np.isfinite(n)) and np.abs(n).max() < 6.0
    # the map itself: (m + 1/2) / 2^23 over the word's top 23 bits, exact in binary32
    want = ((words >> 9).astype(np.float64) + 0.5) / 2.0 ** 23
    assert np.array_equal(u.astype(np.float64), want)
    assert np.allclose(e, -np.log(want), rtol=2e-6, atol=1e-7)


def test_variate_moments():
    rng = np.random.default_rng(7)
    words = rng.integers(0, 2 ** 32, size=1 << 20, dtype=np.uint64).astype(np.uint32)
    u, e, n = _draws(words)
    m = len(words)
    assert abs(u.mean() - 0.5) < 4 * np.sqrt(1 / 12 / m)
    assert abs(e.mean() - 1.0) < 4 / np.sqrt(m) and abs(e.var() - 1.0) < 4 * np.sqrt(8 / m)
    assert abs(n.mean()) < 4 / np.sqrt(m) and abs(n.var() - 1.0) < 4 * np.sqrt(2 / m)
    assert abs(np.corrcoef(n[0::2], n[1::2])[0, 1]) < 4 / np.sqrt(m / 2)


def test_m20_unsupervised_fit_stays_finite():
    """BASELINE config #1's shape (20 mixed columns, K=10, D=32, batch 256) as bench.py's m20 leg fits it."""

    import bench

    rec = bench.m20_subrecord("cuda:0")
    assert np.isfinite(rec["unsupervised_warmup5"]["final_reconstruction"]), rec
    assert np.isfinite(rec["supervised_3_1_1_0"]["final_reconstruction"]), rec
