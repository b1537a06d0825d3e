"""Tuning aid: time the fp32 MFMA GEMM on the S64 layer shapes through the debug entry point."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from suave_amd import lib as _lib
if os.environ.get("SUAVE_LIB"):  # ablation builds (tuning only)
    _lib._LIB_PATH = os.environ["SUAVE_LIB"]
L = _lib.load()
f = L.suave_debug_gemm
vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
f.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, cf, vp]
dev = torch.device("cuda:0")
B = int(os.environ.get("GB", 65536))

def ru(x, m=4):
    return (x + m - 1) // m * m

def run(name, M, N, K, akm, bkm, epi, splits=1, drop=0.0, check=False):
    lda = ru(M) if akm else ru(K)
    ldb = ru(N) if bkm else ru(K)
    ldc = ru(N + 1)
    A = torch.randn((K if akm else M), lda, device=dev)
    Bm = torch.randn((K if bkm else N), ldb, device=dev)
    C = torch.zeros(max(splits, 1) * M, ldc, device=dev)
    bias = torch.randn(N, device=dev)
    H = torch.randn(M, ldc, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    def call():
        rc = f(A.data_ptr(), Bm.data_ptr(), C.data_ptr(), bias.data_ptr(), H.data_ptr() if epi == 2 else None, M, N, K, lda, ldb, ldc,
               int(akm), int(bkm), epi, splits, drop, s)
        assert rc == 0, L.suave_last_error()
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * M * N * K
    byts = 4.0 * (M * K + N * K + max(splits, 1) * M * N + (M * N if epi == 2 else 0))
    print(f"{name:28s} M={M:6d} N={N:4d} K={K:6d} splits={splits:4d} {us:8.1f} us {fl / us / 1e6:7.1f} TF {byts / us / 1e6:6.2f} TB/s", flush=True)
    if check and epi == 0 and not akm:
        Am = A[:, :K]
        Bw = Bm[:K, :N].t() if bkm else Bm[:, :K]
        ref = Am.double() @ Bw.double().t() + bias.double()
        err = (C[:M, :N].double() - ref).abs().max().item() / ref.abs().max().item()
        print("   max rel err", err)

EPI_BIAS, EPI_RELU, EPI_MASK, EPI_SLAB = 0, 1, 2, 3
run("big square (ff,bias)", 4096, 4096, 4096, 0, 0, EPI_BIAS, check=True)
run("enc L1 fwd", B, 256, 205, 0, 0, EPI_RELU, drop=0.1)
run("enc L1 fwd nodrop", B, 256, 205, 0, 0, EPI_RELU)
run("enc L2 fwd", B, 128, 256, 0, 0, EPI_RELU, drop=0.1)
run("dec L1 fwd", B, 256, 32, 0, 0, EPI_RELU, drop=0.1)
run("yproj fwd", B, 141, 128, 0, 0, EPI_BIAS)
run("dec L2 bwd-data", B, 256, 128, 0, 1, EPI_MASK, drop=0.1)
run("yproj bwd-data", B, 128, 141, 0, 1, EPI_MASK, drop=0.1)
run("enc L1 dW", 256, 205, B, 1, 1, EPI_SLAB, splits=64)
run("enc L2 dW", 128, 256, B, 1, 1, EPI_SLAB, splits=128)
run("long-K ff", 8192, 256, 4096, 0, 0, EPI_BIAS)
run("long-K ft", 8192, 256, 4096, 0, 1, EPI_BIAS)
for sp in (32, 64, 128, 256):
    run("enc L1 dW sweep", 256, 205, B, 1, 1, EPI_SLAB, splits=sp)
for sp in (64, 128, 256, 512):
    run("enc L2 dW sweep", 128, 256, B, 1, 1, EPI_SLAB, splits=sp)
