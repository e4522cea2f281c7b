#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== rccl + parity quick"; timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3f_pytest.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r3f_pytest.log
for v in 0 1 0 1; do
  RADEGS_EIG_CACHE=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3f_eig$v.log 2>&1
  tail -1 gpurun_out/r3f_eig$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('EIG=$v', d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['stages_ms'].items() if 'preprocess' in k})" 2>/dev/null || tail -3 gpurun_out/r3f_eig$v.log
done
