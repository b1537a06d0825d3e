#!/bin/bash
# tuning aid: likelihood-kernel chunk width x workers per CU   (usage: sweep_loss.sh [bench args])
for cfg in "24 28" "28 28" "32 28" "20 32" "16 36" "36 28" "24 32"; do set -- $cfg
  SUAVE_LOSS_CHUNK_Y=$1 SUAVE_LOSS_WPC=$2 timeout -k 10 200 python scripts/loss_ablate.py "${@:3}" $BENCH_ARGS 2>/dev/null | tail -n 1 | sed "s/^default/chunk_y=$1 wpc=$2/" || exit 1
done
