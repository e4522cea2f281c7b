#!/bin/bash
# round 3: half-row backward (8 lanes x 4 px): parity subset + same-box A/B
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_executed_grad.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3b_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3b_pytest.log
for v in 0 1 0 1; do
  RADEGS_STREAMS_BWD8=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3b_bwd8_$v.log 2>&1
  tail -1 gpurun_out/r3b_bwd8_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BWD8=$v', d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['stages_ms'].items() if 'blend' in k})" 2>/dev/null || tail -3 gpurun_out/r3b_bwd8_$v.log
done
