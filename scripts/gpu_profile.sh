#!/bin/bash
# Produces the artefacts that go under profiles/: bench line, rocprofv3 kernel stats, PMC traffic.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1; echo "rocprof rc=$?"
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_tcc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES -d $OUT/pmc_sq --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
find $OUT -name "*counter_collection.csv" | while read f; do d=$(basename $(dirname $(dirname $f))); cp $f $OUT/${d}_counters.csv; done
rm -rf $OUT/kt $OUT/pmc_tcc $OUT/pmc_sq
ls -la $OUT
