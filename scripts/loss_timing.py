#!/usr/bin/env python3
"""Phase timing of the likelihood kernel (needs a build with -DSUAVE_LOSS_TIMING; debugging aid only)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess
import torch
import bench
from suave_amd import lib as _lib
L = _lib.load()
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-profile"]
out = (ctypes.c_ulonglong * 8)()
L.suave_debug_loss_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.suave_debug_loss_timing(out, 1)
bench.main()
torch.cuda.synchronize()
L.suave_debug_loss_timing(out, 1)
v = list(out)
n = max(v[5], 1)
names = ["stage_y", "stage_cells", "columns", "barrier_wait", "tail(reduce,writeback,KL)"]
print("wave-tiles:", v[5])
for i, nm in enumerate(names):
    print(f"{nm:28s} {v[i] / n:10.0f} cycles per wave-tile")
