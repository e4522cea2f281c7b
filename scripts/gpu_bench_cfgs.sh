#!/bin/bash
# bench lines of the BASELINE configs (args: extra bench flags)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in C2 C3 C4 C5; do
  timeout 600 python bench.py --config $c --steps 18 --warmup 9 --no-cpu-baseline "$@" > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
  tail -1 gpurun_out/bench_$c.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('$c', d['value'], d['ms_per_step'], d['speculation'], d['config']['num_rendered_min_max'], {k: round(v,3) for k,v in s.items()})" 2>/dev/null || tail -5 gpurun_out/bench_$c.err
done
