#!/bin/bash
# Produces the artefacts that go under profiles/ for one BASELINE config: bench line, rocprofv3 kernel stats, PMC counters.
#   scripts/gpu_profile.sh <tag> [config]      e.g. r02 C2 -> gpurun_out/profiles_r02/ ;  r02 C4 -> gpurun_out/profiles_r02_C4/
set -u
TAG=${1:-r02}
CFG=${2:-C2}
SUF=""; [ "$CFG" != "C2" ] && SUF="_$CFG"
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG$SUF
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="$GRAFT_REPO_ROOT/bench.py --config $CFG"
EXTRA=""; [ "$CFG" != "C2" ] && EXTRA="--no-cpu-baseline"
python $B --steps 20 --warmup 10 $EXTRA > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 400 $OUT/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python $B --steps 20 --warmup 10 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1; echo "rocprof rc=$?"
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc --output-format csv -- python $B --steps 9 --warmup 9 --no-cpu-baseline > $OUT/pmc_tcc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES -d $OUT/pmc_sq --output-format csv -- python $B --steps 9 --warmup 9 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
find $OUT -name "*counter_collection.csv" | while read f; do d=$(basename $(dirname $(dirname $f))); cp $f $OUT/${d}_counters.csv; done
rm -rf $OUT/kt $OUT/pmc_tcc $OUT/pmc_sq
ls $OUT
