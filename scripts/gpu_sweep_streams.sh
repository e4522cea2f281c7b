#!/bin/bash
# streams vs tile-wide kernels across splat sizes (and the coord-map mode): where is the crossover?
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { # label, env, bench args...
  local label=$1; local envs=$2; shift 2
  env $envs timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --views 3 "$@" > gpurun_out/bench_tmp.log 2>&1
  tail -1 gpurun_out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('$label', 'R/P=%.1f'%(d['config']['num_rendered']/$P), d['ms_per_step'], {k: round(s[k],3) for k in ('block_lists','blend_fwd','blend_bwd')})" 2>/dev/null || tail -3 gpurun_out/bench_tmp.log
}
P=400000
for mu in 1.5 3 5 8 12; do
  for st in 1 0; do run "mu=$mu streams=$st" "RADEGS_STREAMS=$st" --points $P --mu-px $mu; done
done
P=400000
for mu in 1.5 3 5 8 12; do
  for st in 1 0; do run "coord map mu=$mu streams=$st" "RADEGS_STREAMS=$st" --config C4 --points $P --mu-px $mu; done
done
