#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== pytest gpu (full)"; timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3i_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3i_pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3i_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r3i_smoke.log
echo "== trace"; bash scripts/gpu_trace.sh > gpurun_out/r3i_trace.log 2>&1; head -3 gpurun_out/r3i_trace.log
echo "== bench"; timeout 600 python bench.py > gpurun_out/r3i_bench.log 2>&1; tail -1 gpurun_out/r3i_bench.log | cut -c1-250
