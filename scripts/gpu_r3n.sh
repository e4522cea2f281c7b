#!/bin/bash
# same hunt with the GPU left idle between processes (cold clocks at the first launches, as after a long test collection)
cd /root/repo
fails=0; n=0
for i in $(seq 1 28); do
  sleep 4
  timeout 120 python -m pytest tests/test_golden.py -m gpu -q -p no:cacheprovider > gpurun_out/r3n_run.log 2>&1
  rc=$?; n=$((n+1))
  if [ $rc -ne 0 ]; then fails=$((fails+1)); cp gpurun_out/r3n_run.log gpurun_out/r3n_fail_$i.log; echo "run $i rc=$rc"; grep -m3 "elements outside\|Error" gpurun_out/r3n_run.log | cut -c1-300; fi
done
for i in $(seq 29 90); do
  timeout 120 python -m pytest tests/test_golden.py -m gpu -q -p no:cacheprovider > gpurun_out/r3n_run.log 2>&1
  rc=$?; n=$((n+1))
  if [ $rc -ne 0 ]; then fails=$((fails+1)); cp gpurun_out/r3n_run.log gpurun_out/r3n_fail_$i.log; echo "run $i rc=$rc"; grep -m3 "elements outside\|Error" gpurun_out/r3n_run.log | cut -c1-300; fi
done
echo "$n runs, $fails failed"
ls gpurun_out | grep golden_state
