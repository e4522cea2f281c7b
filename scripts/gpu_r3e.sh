#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== pytest gpu (full)"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3e_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r3e_pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3e_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r3e_smoke.log
echo "== profile C2"; bash scripts/gpu_profile.sh r03 C2 > gpurun_out/r3e_profile.log 2>&1; tail -5 gpurun_out/r3e_profile.log
