#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ... -- bench args ; prints ms_per_step and kernel ms per class for each value
var=$1; shift
vals=()
while [ "$1" != "--" ]; do vals+=("$1"); shift; done
shift
for v in "${vals[@]}"; do
  env $var=$v timeout -k 10 150 python bench.py --no-cpu-baseline --gen-rows 0 "$@" > gpurun_out/sweep_$v.log 2>&1 || exit 1
  python -c "
import json,sys
l=[x for x in open('gpurun_out/sweep_$v.log') if x.startswith('{')][-1]; d=json.loads(l)
print('$var=$v', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d.get('kernel_ms_per_step',{}).items()})"
done
