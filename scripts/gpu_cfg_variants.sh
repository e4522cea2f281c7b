#!/bin/bash
# one BASELINE config under several environment settings: gpu_cfg_variants.sh C4 "A=1" "RADEGS_BWD_PPL=4" ...
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
c=$1; shift
for cfg in "$@"; do
  echo "== $c $cfg"
  env $cfg timeout 600 python bench.py --config $c --steps 12 --warmup 5 --no-cpu-baseline > gpurun_out/bench_tmp.log 2>&1
  tail -1 gpurun_out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print(d['value'], d['ms_per_step'], {k: round(v,3) for k,v in s.items()})" 2>/dev/null || tail -5 gpurun_out/bench_tmp.log
done
