#!/bin/bash
# round 3, first GPU session: the full -m gpu suite against the reference-produced goldens + the slip default; smoke; one bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3a_pytest_gpu.log 2>&1 ; echo "pytest rc=$?"
tail -15 gpurun_out/r3a_pytest_gpu.log
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3a_smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 gpurun_out/r3a_smoke.log
echo "== bench" ; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a_bench.log 2>&1 ; echo "bench rc=$?" ; tail -1 gpurun_out/r3a_bench.log | cut -c1-600
