#!/bin/bash
cd /root/repo
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r3s_bench_$i.json 2> gpurun_out/r3s_bench_$i.err; echo "bench $i rc=$?"
  python - gpurun_out/r3s_bench_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'], sum(d['stages_ms'].values()), d['stages_ms']['blend_bwd'])
PY
done
nproc; uptime
