#!/bin/bash
# Round-end measurement pass on the GPU box (run through gpurun from the repo root):
#   kernel stats of the default bench command and of the training-only command, the FETCH_SIZE / WRITE_SIZE PMC
#   passes (separate runs, counters only), one-step timelines of S64 and W1024, and the W1024 bench line.
# Every GPU step has its own time limit and the chain stops at the first failure.
set -e -o pipefail
R=${1:-r02}
O=gpurun_out
export TMPDIR=/tmp
RAW="$O/${R}_stats $O/${R}_stats_train $O/${R}_fetch $O/${R}_write $O/${R}_fetch_t $O/${R}_write_t $O/${R}_tl_s64 $O/${R}_tl_w $O/${R}_fetch_w $O/${R}_write_w $O/${R}_pmc_sq/p1 $O/${R}_pmc_sq/p2 $O/${R}_pmc_sq/p3"
trap 'echo "exit status $? at line $LINENO"; rm -rf $RAW' EXIT  # the raw traces exceed what gpurun copies back
rm -rf $O/${R}_stats $O/${R}_stats_train $O/${R}_fetch $O/${R}_write $O/${R}_fetch_t $O/${R}_write_t $O/${R}_tl_s64 $O/${R}_tl_w
echo "[1/9] kernel stats, default command"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${R}_stats --output-format csv -o bench -- python bench.py > $O/${R}_bench_stats.log 2>&1
echo "[2/9] kernel stats, training only"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/${R}_stats_train --output-format csv -o bench -- python bench.py --gen-rows 0 --no-cpu-baseline --no-subrecords > $O/${R}_bench_stats_train.log 2>&1
echo "[3/9] PMC FETCH_SIZE, training only"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/${R}_fetch_t --output-format csv -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-subrecords --no-profile --gen-rows 0 > $O/${R}_fetch_t.log 2>&1
echo "[4/9] PMC WRITE_SIZE, training only"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/${R}_write_t --output-format csv -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-subrecords --no-profile --gen-rows 0 > $O/${R}_write_t.log 2>&1
echo "[5/9] PMC FETCH_SIZE, with generation"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/${R}_fetch --output-format csv -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-subrecords --no-profile > $O/${R}_fetch.log 2>&1
echo "[6/9] PMC WRITE_SIZE, with generation"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/${R}_write --output-format csv -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-subrecords --no-profile > $O/${R}_write.log 2>&1
echo "[7/9] timeline S64"
timeout -k 10 120 rocprofv3 --kernel-trace -d $O/${R}_tl_s64 --output-format csv -o tl -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-subrecords --no-profile --gen-rows 0 > $O/${R}_tl_s64.log 2>&1
echo "[8/9] timeline W1024"
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/${R}_tl_w --output-format csv -o tl -- python bench.py --config w1024 --steps 3 --warmup 2 --no-cpu-baseline --no-subrecords --no-profile --gen-rows 0 > $O/${R}_tl_w.log 2>&1
echo "[summaries]"
mkdir -p $O/${R}_out
python scripts/trim_prof.py $(find $O/${R}_stats -name '*kernel_stats.csv' | tail -1) $O/${R}_out/${R}_bench_kernel_stats.csv > /dev/null
python scripts/trim_prof.py $(find $O/${R}_stats_train -name '*kernel_stats.csv' | tail -1) $O/${R}_out/${R}_bench_train_kernel_stats.csv > /dev/null
python scripts/pmc_traffic.py $O/${R}_fetch_t $O/${R}_write_t 7 > $O/${R}_out/${R}_pmc_traffic_train.json
python scripts/pmc_traffic.py $O/${R}_fetch $O/${R}_write > $O/${R}_out/${R}_pmc_traffic_gen.json
python scripts/step_timeline.py $(find $O/${R}_tl_s64 -name '*kernel_trace.csv' | tail -1) > $O/${R}_out/${R}_step_timeline.txt
python scripts/step_timeline.py $(find $O/${R}_tl_w -name '*kernel_trace.csv' | tail -1) > $O/${R}_out/${R}_w1024_step_timeline.txt
grep -h "^{\"metric\"" $O/${R}_bench_stats.log | tail -1 > $O/${R}_out/${R}_bench_default_under_rocprof.json || true
rm -rf $O/${R}_stats $O/${R}_stats_train $O/${R}_fetch $O/${R}_write $O/${R}_fetch_t $O/${R}_write_t $O/${R}_tl_s64 $O/${R}_tl_w
echo "[8b] PMC FETCH_SIZE / WRITE_SIZE, W1024 training only"
rm -rf $O/${R}_fetch_w $O/${R}_write_w
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/${R}_fetch_w --output-format csv -o pmc -- python bench.py --config w1024 --steps 3 --warmup 1 --no-cpu-baseline --no-subrecords --no-profile --gen-rows 0 > $O/${R}_fetch_w.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/${R}_write_w --output-format csv -o pmc -- python bench.py --config w1024 --steps 3 --warmup 1 --no-cpu-baseline --no-subrecords --no-profile --gen-rows 0 > $O/${R}_write_w.log 2>&1
python scripts/pmc_traffic.py $O/${R}_fetch_w $O/${R}_write_w 4 > $O/${R}_out/${R}_pmc_traffic_w1024.json
rm -rf $O/${R}_fetch_w $O/${R}_write_w
echo "[8c] SQ counter passes (VALU / LDS / MFMA / waits per kernel), S64 training step"
if bash scripts/pmc_kernels.sh ${R}_pmc_sq; then cp $O/${R}_pmc_sq_summary.txt $O/${R}_out/${R}_pmc_sq_summary.txt; else echo "SQ passes failed: see $O/${R}_pmc_sq/*.log (rerun scripts/pmc_kernels.sh on its own)"; fi
echo "[9/9] W1024 bench line"
timeout -k 10 600 python bench.py --config w1024 > $O/${R}_bench_w1024.json 2> $O/${R}_bench_w1024.err
echo done
