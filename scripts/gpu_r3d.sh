#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== parity subset"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_rccl.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3d_pytest.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r3d_pytest.log
for cfg in C2 C4; do for v in 0 1 0 1; do
  RADEGS_EIG_CACHE=$v timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3d_${cfg}_eig$v.log 2>&1
  tail -1 gpurun_out/r3d_${cfg}_eig$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg EIG=$v', d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['stages_ms'].items() if 'preprocess' in k})" 2>/dev/null || tail -3 gpurun_out/r3d_${cfg}_eig$v.log
done; done
echo "== fuzz table (exact bwd, tile-wide seeds)"; RADEGS_LIB=$PWD/gpurun_ab/libradegs_bwdexact.so timeout 900 python scripts/gpu_fuzz_table.py 0:40 > gpurun_out/r3c_fuzz_exact.txt 2> gpurun_out/r3c_fuzz_exact.err; grep -v " ok$" gpurun_out/r3c_fuzz_exact.txt
