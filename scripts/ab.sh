#!/bin/bash
# Same-box A/B of an environment knob on the S64 step: bash scripts/ab.sh VAR=value [repeats]   (through gpurun)
set -e -o pipefail
KV=$1; N=${2:-3}
run() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-profile --gen-rows 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step' % d['ms_per_step'])"; }
for i in $(seq $N); do
  echo -n "base      : "; run
  echo -n "$KV : "; env $KV bash -c "$(declare -f run); run"
done
