#!/bin/bash
# quick GPU session: parity tests + PPL sweep of the bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1 ; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
CFGS=("$@"); [ ${#CFGS[@]} -eq 0 ] && CFGS=(2:4:1 2:2:1)
for cfg in "${CFGS[@]}"; do
  IFS=: read fp bp dpp <<< "$cfg"
  echo "== bench PPL fwd=$fp bwd=$bp dpp=$dpp"; RADEGS_FWD_PPL=$fp RADEGS_BWD_PPL=$bp RADEGS_BWD_DPP=$dpp timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_q_${fp}_${bp}_${dpp}.log 2>&1; tail -1 gpurun_out/bench_q_${fp}_${bp}_${dpp}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages_ms'])" 2>/dev/null || tail -3 gpurun_out/bench_q_${fp}_${bp}_${dpp}.log
done
