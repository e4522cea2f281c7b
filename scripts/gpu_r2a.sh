#!/bin/bash
# round-2 quick session: stream-path parity tests + bench with entry streams on/off
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest (stream paths)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "blend_paths or entry_streams_heavy or tile_wide_backward or small_scene or config_C1 or speculative" > gpurun_out/pytest_streams.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_streams.log
for st in 1 0; do
  echo "== bench STREAMS=$st"
  RADEGS_STREAMS=$st timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_streams_$st.log 2>&1
  tail -1 gpurun_out/bench_streams_$st.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages_ms'])" 2>/dev/null || tail -5 gpurun_out/bench_streams_$st.log
done
