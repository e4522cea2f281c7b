#!/bin/bash
# tile-wide backward: one wave per 16x8 strip (2 px/lane) against one wave per tile (4 px/lane), across splat sizes
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { # label, env, bench args...
  local label=$1; local envs=$2; shift 2
  env $envs timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --views 3 "$@" > gpurun_out/bench_tmp.log 2>&1
  tail -1 gpurun_out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('$label', 'R/P=%.1f'%(d['config']['num_rendered']/$P), d['ms_per_step'], {k: round(s[k],3) for k in ('blend_fwd','blend_bwd')})" 2>/dev/null || tail -3 gpurun_out/bench_tmp.log
}
P=400000
for mu in 5 8 12 18; do
  for ppl in 2 4; do run "depth mu=$mu bwd_ppl=$ppl" "RADEGS_STREAMS=0 RADEGS_BWD_PPL=$ppl" --points $P --mu-px $mu; done
done
for mu in 5 8 12; do
  for ppl in 2 4; do run "coord mu=$mu bwd_ppl=$ppl" "RADEGS_STREAMS=0 RADEGS_BWD_PPL=$ppl" --config C4 --points $P --mu-px $mu; done
done
