#!/bin/bash
# refresh of the rocprofv3 kernel-trace summary with the final library (the r03 csv predates the fold of two small launches into the emission)
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_r03u
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 100 rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 10 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1; echo "rocprof rc=$?"
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/kt
tail -c 300 $OUT/rocprof_bench.log; head -5 $OUT/kernel_stats.csv | cut -c1-160
