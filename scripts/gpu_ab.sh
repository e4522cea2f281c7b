#!/bin/bash
# same-box A/B of alternative builds: scripts/gpu_ab.sh name1 name2 ...   ("base" = the in-tree library)
set -u
mkdir -p gpurun_out
for n in "$@"; do
  if [ "$n" = base ]; then unset RADEGS_LIB; else export RADEGS_LIB=$PWD/gpurun_ab/libradegs_$n.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_$n.log 2>&1
  tail -1 gpurun_out/ab_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['stages_ms'].items()})" 2>/dev/null || tail -3 gpurun_out/ab_$n.log
done
