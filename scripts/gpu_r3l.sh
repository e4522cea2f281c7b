#!/bin/bash
# poison experiment: does any kernel read what this call has not written / leave a pixel unwritten?
cd /root/repo
export RADEGS_DEBUG_POISON=1
timeout 600 python -m pytest tests/test_golden.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider > gpurun_out/r3l_poison_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3l_poison_pytest.log
timeout 400 python scripts/gpu_stress_golden.py 300 > gpurun_out/r3l_poison_stress.log 2>&1; echo "stress rc=$?"; tail -8 gpurun_out/r3l_poison_stress.log
RADEGS_SPECULATE=0 timeout 400 python scripts/gpu_stress_golden.py 200 > gpurun_out/r3l_poison_stress_nospec.log 2>&1; echo "stress nospec rc=$?"; tail -8 gpurun_out/r3l_poison_stress_nospec.log
