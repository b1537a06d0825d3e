#!/bin/bash
# tuning aid: likelihood-kernel chunk width x workers per CU   (usage: sweep_loss.sh [bench args])
for cfg in "20 16" "24 28" "28 32" "14 16" "20 32"; do set -- $cfg; cy=$1; w=$2
  SUAVE_LOSS_CHUNK_Y=$cy SUAVE_LOSS_WPC=$w timeout -k 10 200 python bench.py --no-cpu-baseline --gen-rows 0 ${BENCH_ARGS} > gpurun_out/sweep_l.log 2>&1 || exit 1
  python -c "
import json
l=[x for x in open('gpurun_out/sweep_l.log') if x.startswith('{')][-1]; d=json.loads(l)
print('chunk_y=$cy wpc=$w', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})"
done
