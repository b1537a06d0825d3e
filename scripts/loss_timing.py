"""Phase timing of the likelihood kernel (needs a build with SUAVE_EXTRA_FLAGS=-DSUAVE_LOSS_TIMING; debugging aid)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from suave_amd import lib as _lib
if os.environ.get("SUAVE_LIB"):
    _lib._LIB_PATH = os.environ["SUAVE_LIB"]
L = _lib.load()
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-profile", "--no-subrecords", "--gen-rows", "0"]
out = (ctypes.c_ulonglong * 16)()
L.suave_debug_loss_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.suave_debug_loss_timing(out, 1)
bench.main()
torch.cuda.synchronize()
L.suave_debug_loss_timing(out, 1)
v = list(out)
tiles = max(v[13], 1)
print("wave-tiles:", v[13], "waves:", v[14])
print(f"staging            {v[0] / tiles:10.0f} cycles per wave-tile")
for t, nm in enumerate(["real", "pos", "count", "cat", "ordinal"]):
    print(f"column {nm:8s}    {v[1 + t] / max(v[8 + t], 1):10.0f} cycles per column  ({v[8 + t]} columns)")
print(f"outputs+writeback  {v[6] / tiles:10.0f} cycles per wave-tile")
print(f"whole wave         {v[7] / max(v[14], 1):10.0f} cycles per wave")
print(f"  recon/sum part   {v[15] / tiles:10.0f} cycles per wave-tile")
