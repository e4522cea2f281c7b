#!/bin/bash
# bench line + rocprofv3 kernel-trace summary of the same command on the SAME box (final library)
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_r03v
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 100 rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 10 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1; echo "rocprof rc=$?"
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/kt
python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 10 --no-cpu-baseline > $OUT/bench2.json 2> $OUT/bench2.err; echo "bench2 rc=$?"
head -3 $OUT/kernel_stats.csv | cut -c1-160
