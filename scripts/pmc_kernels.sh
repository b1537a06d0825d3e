#!/bin/bash
# SQ counter passes over the training step (separate rocprofv3 runs, counters only, <= 8 SQ counters each), summarised
# per kernel by scripts/rocpd_summary.py.   usage (through gpurun, repo root): bash scripts/pmc_kernels.sh <tag> [bench args]
set -e -o pipefail
TAG=${1:-pmc}; shift || true
O=gpurun_out/$TAG
export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-subrecords --no-profile --gen-rows 0 $*"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $O/p1 -- python3 bench.py $ARGS > $O/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $O/p2 -- python3 bench.py $ARGS > $O/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE SQ_WAVES_EQ_64 SQ_LEVEL_WAVES -d $O/p3 -- python3 bench.py $ARGS > $O/p3.log 2>&1 || echo "pass 3 failed (counter names)"
python3 scripts/rocpd_summary.py $(find $O -name '*_results.db' | sort) > gpurun_out/${TAG}_summary.txt
rm -rf $O/p1 $O/p2 $O/p3
