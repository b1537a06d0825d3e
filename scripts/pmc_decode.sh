#!/bin/bash
# SQ counter passes over the fused decode kernel (separate rocprofv3 runs, counters only, <= 8 SQ counters each).
# usage (through gpurun, repo root): bash scripts/pmc_decode.sh <tag>
set -e -o pipefail
TAG=${1:-pmc_decode}
O=gpurun_out/$TAG
export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $O/p1 -- python3 scripts/decode_probe.py > $O/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $O/p2 -- python3 scripts/decode_probe.py > $O/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_WAVES_EQ_64 SQ_LEVEL_WAVES -d $O/p3 -- python3 scripts/decode_probe.py > $O/p3.log 2>&1 || echo "pass 3 failed (counter names)"
python3 scripts/rocpd_summary.py $(find $O -name '*_results.db' | sort) > gpurun_out/${TAG}_summary.txt
rm -rf $O/p1 $O/p2 $O/p3
