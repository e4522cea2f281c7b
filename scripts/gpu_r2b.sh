#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest (stream paths)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "blend_paths or entry_streams_heavy or tile_wide_backward or small_scene or config_C1" > gpurun_out/pytest_streams.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_streams.log
for cfg in "$@"; do
  echo "== bench $cfg"
  env $cfg timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_tmp.log 2>&1
  tail -1 gpurun_out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print(d['value'], d['ms_per_step'], {k: s[k] for k in ('block_lists','blend_fwd','blend_bwd')})" 2>/dev/null || tail -5 gpurun_out/bench_tmp.log
done
