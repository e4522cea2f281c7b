#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_vs_compiled_reference.py tests/test_gpu_executed_grad.py tests/test_gpu_rccl.py -m gpu -q -x -s -p no:cacheprovider > gpurun_out/r3q_pytest.log 2>&1; echo "rc=$?"
grep "executed-mode difference\|passed\|failed" gpurun_out/r3q_pytest.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3q_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3q_smoke.log
