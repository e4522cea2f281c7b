#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_poison.py tests/test_golden.py tests/test_gpu_rccl.py tests/test_gpu_view_parallel.py -m gpu -q -x -p no:cacheprovider --durations=5 > gpurun_out/r3o_pytest.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r3o_pytest.log
