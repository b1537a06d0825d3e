#!/bin/bash
# Kernel statistics of the generation passes of bench.py (sample / impute / MC-32), training reduced to one step.
#   usage (through gpurun, repo root): bash scripts/gen_stats.sh <out.csv> [bench args]
set -e -o pipefail
OUT=${1:-gpurun_out/gen_stats.csv}; shift || true
export TMPDIR=/tmp
D=gpurun_out/_gs
rm -rf $D
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $D --output-format csv -o gs -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile "$@" > $D.log 2>&1
python3 scripts/trim_prof.py $(find $D -name '*kernel_stats.csv' | tail -1) $OUT > /dev/null
rm -rf $D
