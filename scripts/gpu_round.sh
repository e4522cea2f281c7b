#!/bin/bash
# One GPU-box session: diagnostics, tests, smoke, bench, rocprof.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -6 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
echo "== diag" ; timeout 600 python scripts/gpu_diag.py > gpurun_out/diag.log 2>&1 ; echo "diag rc=$?"
tail -60 gpurun_out/diag.log
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1 ; echo "pytest rc=$?"
tail -40 gpurun_out/pytest_gpu.log
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -5 gpurun_out/smoke.log
echo "== bench" ; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1 ; echo "bench rc=$?" ; tail -5 gpurun_out/bench.log
for fp in 1 2 4; do for bp in 1 2 4; do
  echo "== bench PPL fwd=$fp bwd=$bp"; RADEGS_FWD_PPL=$fp RADEGS_BWD_PPL=$bp timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_ppl_${fp}_${bp}.log 2>&1; tail -1 gpurun_out/bench_ppl_${fp}_${bp}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages_ms'])" 2>/dev/null || tail -3 gpurun_out/bench_ppl_${fp}_${bp}.log
done; done
echo "== rocprof" ; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1) ; echo "rocprof rc=$?"
find gpurun_out/prof -name "*stats*" | head
