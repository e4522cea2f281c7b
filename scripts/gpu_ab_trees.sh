#!/bin/bash
# Same-box A/B of SEVERAL trees (this one + gpurun_ab/<name>tree ...), alternating:   scripts/gpu_ab_trees.sh <tag> <rounds> <name> [<name> ...]
set -u
TAG=$1; N=$2; shift; shift
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  for t in here "$@"; do
    if [ $t = here ]; then d=$GRAFT_REPO_ROOT; else d=$GRAFT_REPO_ROOT/gpurun_ab/${t}tree; fi
    (cd $d && timeout 600 python bench.py $ARGS 2>/dev/null | tail -1) > gpurun_out/${TAG}_ab_${t}_$i.json
    python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_ab_${t}_$i.json"))
st = d.get("stages_ms", {})
print("$t", $i, d["value"], d["ms_per_step"], "span", d.get("step_gpu_span_ms", {}).get("median"), {k: round(st[k], 4) for k in ("preprocess_fwd", "blend_fwd", "acc_zero", "blend_bwd", "preprocess_bwd") if k in st})
PY
  done
done
