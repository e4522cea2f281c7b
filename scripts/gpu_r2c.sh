#!/bin/bash
# full parity file + bench variants given as arguments ("ENV=.. ENV=.." strings)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest parity"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_parity.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_parity.log
for cfg in "$@"; do
  echo "== bench $cfg"
  env $cfg timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_tmp.log 2>&1
  tail -1 gpurun_out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages_ms'])" 2>/dev/null || tail -5 gpurun_out/bench_tmp.log
done
