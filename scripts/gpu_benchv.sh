#!/bin/bash
# bench variants: each argument is an "ENV=.. ENV=.." string
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "$@"; do
  echo "== bench $cfg"
  env $cfg timeout 300 python bench.py --steps 18 --warmup 9 --no-cpu-baseline > gpurun_out/bench_tmp.log 2>&1
  tail -1 gpurun_out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print(d['value'], d['ms_per_step'], {k: round(v,3) for k,v in s.items()})" 2>/dev/null || tail -5 gpurun_out/bench_tmp.log
done
