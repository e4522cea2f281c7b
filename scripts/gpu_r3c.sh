#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== rccl"; timeout 900 python -m pytest tests/test_gpu_rccl.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3c_rccl.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r3c_rccl.log
echo "== fuzz table (in-tree)"; timeout 900 python scripts/gpu_fuzz_table.py 0:40 > gpurun_out/r3c_fuzz_intree.txt 2> gpurun_out/r3c_fuzz_intree.err; tail -3 gpurun_out/r3c_fuzz_intree.txt
echo "== fuzz table (exact bwd)"; RADEGS_LIB=$PWD/gpurun_ab/libradegs_bwdexact.so timeout 900 python scripts/gpu_fuzz_table.py 0:40 > gpurun_out/r3c_fuzz_exact.txt 2> gpurun_out/r3c_fuzz_exact.err; tail -3 gpurun_out/r3c_fuzz_exact.txt
echo "== bench both flags"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --flags both > gpurun_out/r3c_bench_both.log 2>&1; tail -1 gpurun_out/r3c_bench_both.log | cut -c1-300
echo "== full C4/C5 arbiter"; timeout 1500 python -m pytest tests/test_gpu_full.py -m gpu -q -x -k "full_size_oracle_parity" -p no:cacheprovider > gpurun_out/r3c_full.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r3c_full.log
