"""The MFMA GEMM of the dense layers against torch float64 on ragged shapes, in both arithmetic modes.

The kernel is floating point, so this is the one place with a plain-PyTorch reference (the ELBO parity tests
check the same kernels end to end against the oracle).  Tolerance: 2e-5 of the largest |reference| entry -- the
error of an fp32 FMA chain; a bf16 product alone would miss it by two orders of magnitude.  Padding columns of
the operands (between the logical width and the leading dimension) are poisoned with NaN: nothing past the
matrix edge may reach an output.
"""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPI_BIAS, EPI_RELU, EPI_MASK, EPI_SLAB = 0, 1, 2, 3
TOL = 2e-5


def _ru(x, m=4):
    return (x + m - 1) // m * m


def _debug_gemm():
    from suave_amd import lib as slib

    L = slib.load()
    f = L.suave_debug_gemm  # tuning / test entry point of libsuave_hip.so (one plain launch on device pointers)
    vp, ci = C.c_void_p, C.c_int
    f.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, C.c_float, vp]
    f.restype = ci
    return slib, L, f


def _padded(rows, cols, ld, gen, dev, poison=True):
    import torch

    t = torch.full((rows, ld), float("nan") if poison else 0.0, device=dev, dtype=torch.float32)
    t[:, :cols] = torch.randn(rows, cols, generator=gen, device=dev)
    return t


@pytest.fixture(params=[4, 3, 0], ids=["f16x2", "bf16x3", "exact"])
def arithmetic(request):
    slib, L, _ = _debug_gemm()
    before = C.c_int(-1)
    slib.check(L.suave_gemm_arithmetic_get(C.byref(before)), "suave_gemm_arithmetic_get")
    slib.check(L.suave_gemm_arithmetic_set(request.param), "suave_gemm_arithmetic_set")
    yield request.param
    slib.check(L.suave_gemm_arithmetic_set(before.value), "suave_gemm_arithmetic_set")


SHAPES = [(1, 1, 1), (5, 3, 7), (128, 128, 16), (129, 127, 17), (1000, 141, 128), (257, 65, 33), (300, 650, 260),
          (4096, 256, 205), (2048, 64, 32), (513, 40, 96),
          (4096, 10, 128), (2048, 16, 33), (1500, 3, 260), (65536, 10, 128)]  # skinny outputs (component logits, class heads): N <= 16 on long batches


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("epi", [EPI_BIAS, EPI_RELU, EPI_MASK])
def test_row_major_forms(arithmetic, M, N, K, epi):
    """C = A @ W^T (+ bias | + bias, ReLU | masked by H > 0): forward layers and data gradients on W^T."""

    import torch

    slib, L, f = _debug_gemm()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1000 * M + 10 * N + K)
    lda, ldb, ldc = _ru(K), _ru(K), _ru(N + 1)
    A, W = _padded(M, K, lda, gen, dev), _padded(N, K, ldb, gen, dev)
    bias = torch.randn(N, generator=gen, device=dev)
    H = torch.randn(M, ldc, generator=gen, device=dev)
    out = torch.full((M, ldc), 7.5, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    rc = f(A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), H.data_ptr() if epi == EPI_MASK else None, M, N, K,
           lda, ldb, ldc, 0, 0, epi, 1, 0.0, s)
    assert rc == 0, L.suave_last_error()
    torch.cuda.synchronize()
    ref = A[:, :K].double() @ W[:, :K].double().t()
    if epi == EPI_BIAS:
        ref = ref + bias.double()
    elif epi == EPI_RELU:
        ref = torch.relu(ref + bias.double())
    else:
        ref = torch.where(H[:, :N] > 0, ref, torch.zeros_like(ref))
    scale = max(float(ref.abs().max()), 1e-6)
    got = out[:, :N].double()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= TOL * scale
    assert bool((out[:, N:] == 7.5).all())  # nothing written past the N real columns


def test_fp16_plane_range_is_loud_and_small_operands_keep_their_precision():
    """Mode 4 cuts operands into fp16 planes: an element beyond fp16's range must give a non-finite result (never a
    silently clipped one) while mode 3 handles it; operands far below fp16's normal range (1e-7: subnormal hi plane)
    still come out with an ABSOLUTE error of ~1e-11 per element thanks to the scaled residual plane."""

    import torch

    slib, L, f = _debug_gemm()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(5)
    M, N, K = 256, 128, 64
    lda, ldb, ldc = _ru(K), _ru(K), _ru(N + 1)
    bias = torch.zeros(N, device=dev)
    before = C.c_int(-1)
    slib.check(L.suave_gemm_arithmetic_get(C.byref(before)), "suave_gemm_arithmetic_get")

    def run(A, W, mode):
        slib.check(L.suave_gemm_arithmetic_set(mode), "suave_gemm_arithmetic_set")
        out = torch.zeros(M, ldc, device=dev)
        rc = f(A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), None, M, N, K, lda, ldb, ldc, 0, 0, EPI_BIAS, 1,
               0.0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, L.suave_last_error()
        torch.cuda.synchronize()
        return out[:, :N].double()

    try:
        A, W = _padded(M, K, lda, gen, dev, poison=False), _padded(N, K, ldb, gen, dev, poison=False)
        A[3, 5] = 1.0e5
        ref = A[:, :K].double() @ W[:, :K].double().t()
        assert not torch.isfinite(run(A, W, 4)[3]).all()          # loud
        assert torch.isfinite(run(A, W, 4)[4:]).all()             # and confined to the rows that hold the element
        got3 = run(A, W, 3)
        assert float((got3 - ref).abs().max()) <= TOL * float(ref.abs().max())
        A2 = _padded(M, K, lda, gen, dev, poison=False) * 1.0e-7
        ref2 = A2[:, :K].double() @ W[:, :K].double().t()
        got4 = run(A2, W, 4)
        assert float((got4 - ref2).abs().max()) <= 1e-4 * float(ref2.abs().max())
    finally:
        slib.check(L.suave_gemm_arithmetic_set(before.value), "suave_gemm_arithmetic_set")


@pytest.mark.parametrize("M,N,K", [(129, 127, 17), (1000, 141, 128), (300, 650, 260), (513, 40, 96)])
def test_k_major_weight_operand(arithmetic, M, N, K):
    """C = A @ B with B stored [K, N] (the data-gradient form on the untransposed weight)."""

    import torch

    slib, L, f = _debug_gemm()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(7 * M + N + K)
    lda, ldb, ldc = _ru(K), _ru(N), _ru(N + 1)
    A, B = _padded(M, K, lda, gen, dev), _padded(K, N, ldb, gen, dev)
    bias = torch.zeros(N, device=dev)
    out = torch.zeros(M, ldc, device=dev)
    rc = f(A.data_ptr(), B.data_ptr(), out.data_ptr(), bias.data_ptr(), None, M, N, K, lda, ldb, ldc, 0, 1, EPI_BIAS, 1, 0.0,
           torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.suave_last_error()
    torch.cuda.synchronize()
    ref = A[:, :K].double() @ B[:, :N].double()
    assert float((out[:, :N].double() - ref).abs().max()) <= TOL * max(float(ref.abs().max()), 1e-6)


@pytest.mark.parametrize("M,N,K,splits", [(256, 205, 4096, 8), (128, 256, 8192, 16), (141, 128, 5000, 7), (37, 19, 300, 3),
                                          (256, 33, 4096, 4), (650, 132, 2500, 5)])
def test_weight_gradient_slabs(arithmetic, M, N, K, splits):
    """sum_s C[s] = A^T @ B with both operands stored [K, features] and the K range cut into slabs (dW = dC^T A)."""

    import torch

    slib, L, f = _debug_gemm()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(M + 3 * N + K + splits)
    lda, ldb, ldc = _ru(M), _ru(N), _ru(N + 1)
    A, B = _padded(K, M, lda, gen, dev), _padded(K, N, ldb, gen, dev)
    slabs = torch.zeros(splits, M, ldc, device=dev)
    rc = f(A.data_ptr(), B.data_ptr(), slabs.data_ptr(), None, None, M, N, K, lda, ldb, ldc, 1, 1, EPI_SLAB, splits, 0.0,
           torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.suave_last_error()
    torch.cuda.synchronize()
    got = slabs[:, :, :N].double().sum(0)
    ref = A[:, :M].double().t() @ B[:, :N].double()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= TOL * max(float(ref.abs().max()), 1e-6)
