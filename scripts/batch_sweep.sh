#!/bin/bash
# Step time against the batch size (one line per batch): bash scripts/batch_sweep.sh   (through gpurun, repo root)
set -e -o pipefail
for b in 256 1024 4096 16384 32768 131072 262144; do
  timeout -k 10 120 python bench.py --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-subrecords --no-profile --gen-rows 0 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch %7d: %8.3f M rows/s  %.4f ms/step' % ($b, d['value']/1e6, d['ms_per_step']))"
done
