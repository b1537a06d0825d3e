"""Tuning aid: one launch of each S64 GEMM shape.  Under the profiler the interpreter itself goes after `--`
(`rocprofv3 --pmc ... -- python3 scripts/gemm_pmc.py`): no shebang / env hop after the GPU is initialised."""
import os, sys
sys.argv = [sys.argv[0]]
os.environ.setdefault("GB", "65536")
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_bench.py")).read().split("EPI_BIAS, EPI_RELU")[0])
EPI_BIAS, EPI_RELU, EPI_MASK, EPI_SLAB = 0, 1, 2, 3
run("big square", 4096, 4096, 4096, 0, 0, EPI_BIAS)
run("enc L1 fwd", B, 256, 205, 0, 0, EPI_RELU, drop=0.1)
run("enc L2 fwd", B, 128, 256, 0, 0, EPI_RELU, drop=0.1)
run("dec L2 bwd-data", B, 256, 128, 0, 1, EPI_MASK, drop=0.1)
run("enc L2 dW", 128, 256, B, 1, 1, EPI_SLAB, splits=128)
