#!/bin/bash
# Bench lines of every profiled workload again, AFTER the PMC summaries of the same library are in profiles/ (bench.py attributes
# `traffic`, `valu_roofline` and `stage_bytes_moved` from the newest committed PMC pass of the workload): gpurun_out/profiles_<tag>*/bench.json
set -u
TAG=${1:-r05}
export TMPDIR=/tmp
run() { # suffix, bench args...
  local suf=$1; shift
  local out=gpurun_out/profiles_$TAG$suf
  mkdir -p $out
  local nocpu="--no-cpu-baseline --no-other-configs"; [ -z "$suf" ] && nocpu=""
  timeout 900 python bench.py --steps 20 --warmup 10 $nocpu "$@" > $out/bench.json 2> $out/bench.err; echo "bench$suf rc=$? $(tail -c 120 $out/bench.json)"
}
run ""
run _C2_both --flags both
run _C2_both_fwd --flags both --mode forward
run _C3 --config C3
run _C4 --config C4
run _C5 --config C5
run _C2H --config C2H
