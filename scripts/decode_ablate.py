"""Timing-only ablations of the fused decode kernel: the bench's generation legs on the library named by SUAVE_LIB
(a build of decode.hip with -DDEC_ABL=<n>; results are wrong by construction).  debugging aid"""
import io, json, os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from suave_amd import lib as _lib
if os.environ.get("SUAVE_LIB"):
    _lib._LIB_PATH = os.environ["SUAVE_LIB"]
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-subrecords"] + sys.argv[1:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
g = d["generation"]
print(os.path.basename(os.environ.get("SUAVE_LIB", "default")), {k: (round(v["rows_per_s"] / 1e6, 1), round(v.get("sample_kernel_ms", 0), 2)) for k, v in g.items() if "rows_per_s" in v})
