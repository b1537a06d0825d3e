#!/bin/bash
# Same-box sweep of environment knobs on the S64 step: bash scripts/ab_multi.sh "A=1" "B=2 C=3" ...   (through gpurun)
set -e -o pipefail
run() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-profile --gen-rows 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step' % d['ms_per_step'])"; }
for rep in 1 2; do
  echo -n "base : "; run
  for kv in "$@"; do echo -n "$kv : "; env $kv bash -c "$(declare -f run); run"; done
done
