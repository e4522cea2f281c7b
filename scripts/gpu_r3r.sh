#!/bin/bash
cd /root/repo
timeout 400 python bench.py > gpurun_out/r3r_bench.json 2> gpurun_out/r3r_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r3r_bench.json
timeout 200 python bench.py --flags both --no-cpu-baseline > gpurun_out/r3r_bench_both.json 2> gpurun_out/r3r_bench_both.err; echo "both rc=$?"; head -c 400 gpurun_out/r3r_bench_both.json
