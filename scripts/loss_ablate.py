"""Timing-only ablations of the likelihood kernel: runs the S64 training bench on the library named by SUAVE_LIB
(a build with SUAVE_EXTRA_FLAGS=-DSUAVE_LOSS_ABL=<bits>; results are wrong by construction) and prints the
per-kernel-class milliseconds.  debugging aid"""
import io, json, os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from suave_amd import lib as _lib
if os.environ.get("SUAVE_LIB"):
    _lib._LIB_PATH = os.environ["SUAVE_LIB"]
sys.argv = ["bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-subrecords", "--gen-rows", "0"] + sys.argv[1:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print(os.path.basename(os.environ.get("SUAVE_LIB", "default")), round(d["ms_per_step"], 4),
      {k: round(v, 4) for k, v in d["kernel_ms_per_step"].items()})
