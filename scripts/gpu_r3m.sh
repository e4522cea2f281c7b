#!/bin/bash
# hunt for the one-off g_C1 map mismatch: fresh processes, the golden GPU tests in suite order; a mismatch leaves its state buffers
cd /root/repo
N=${1:-70}
RADEGS_GOLDEN_DUMP=1 timeout 120 python -m pytest tests/test_golden.py -m gpu -q -x -p no:cacheprovider -k "g_C1" > gpurun_out/r3m_good.log 2>&1; echo "baseline rc=$?"
fails=0
for i in $(seq 1 $N); do
  timeout 120 python -m pytest tests/test_golden.py -m gpu -q -p no:cacheprovider > gpurun_out/r3m_run.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); cp gpurun_out/r3m_run.log gpurun_out/r3m_fail_$i.log; echo "run $i rc=$rc"; grep -m3 "elements outside\|Error" gpurun_out/r3m_run.log | cut -c1-300; fi
done
echo "$N runs, $fails failed"
ls gpurun_out | grep golden_state
