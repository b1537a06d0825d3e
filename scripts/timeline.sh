#!/bin/bash
# One training step kernel by kernel (rocprofv3 --kernel-trace -> scripts/step_timeline.py).
#   usage (through gpurun, repo root): bash scripts/timeline.sh <out.txt> [bench args]
set -e -o pipefail
OUT=${1:-gpurun_out/timeline.txt}; shift || true
export TMPDIR=/tmp
D=gpurun_out/_tl
rm -rf $D
timeout -k 10 150 rocprofv3 --kernel-trace -d $D --output-format csv -o tl -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --gen-rows 0 "$@" > $D.log 2>&1
python3 scripts/step_timeline.py $(find $D -name '*kernel_trace.csv' | tail -1) > $OUT
rm -rf $D
