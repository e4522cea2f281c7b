#!/bin/bash
# What do the SQ "VALU active" counters count on gfx950, and at which clock do the VALU-bound kernels run?  
# Runs scripts/ubench/valu_rates (kernels of KNOWN instruction counts: iters x 64 instructions of one class per wave, 4 waves per SIMD on
# every SIMD) under rocprofv3 with the counters the blend kernels' "VALU busy" figure is built from, in separate passes, next to a
# kernel trace for the durations.  scripts/valu_calibration.py turns the CSVs into profiles/r06_valu_calibration.txt.
#   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/valu_rates scripts/ubench/valu_rates.hip     (here; the binary travels)
#   gpurun -- bash scripts/gpu_valu_calibration.sh r6
set -u
TAG=${1:-r6}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_valu_cal
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BIN=$GRAFT_REPO_ROOT/scripts/ubench/valu_rates
$BIN > $OUT/plain.txt 2>&1
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name --output-format csv -- $BIN > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run p1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run p2 GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM
run p3 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES
run p4 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
find $OUT -name "*counter_collection.csv" | while read f; do d=$(basename $(dirname $(dirname $f))); cp $f $OUT/${d}_counters.csv; done
find $OUT -name "*kernel_trace.csv" | while read f; do d=$(basename $(dirname $(dirname $f))); cp $f $OUT/${d}_trace.csv; done
for d in p1 p2 p3 p4; do rm -rf $OUT/$d; done
ls -la $OUT
