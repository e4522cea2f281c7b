#!/bin/bash
# Same-box A/B of two whole TREES (an older commit copied to gpurun_ab/<name>tree with its own built library and bench.py):
#   scripts/gpu_ab_tree.sh <tag> <name> [rounds] [bench args...]
# alternates `python bench.py` of this tree and of gpurun_ab/<name>tree, prints value / ms_per_step / blend stages of every run.
set -u
TAG=$1; NAME=$2; N=${3:-3}; shift; shift; [ $# -gt 0 ] && shift
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $*"
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  for t in here $NAME; do
    if [ $t = here ]; then d=$GRAFT_REPO_ROOT; else d=$GRAFT_REPO_ROOT/gpurun_ab/${NAME}tree; fi
    (cd $d && timeout 600 python bench.py $ARGS 2>/dev/null | tail -1) > gpurun_out/${TAG}_ab_${t}_$i.json
    python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_ab_${t}_$i.json"))
st = d.get("stages_ms", {})
print("$t", $i, d["value"], d["ms_per_step"], "span", d.get("step_gpu_span_ms", {}).get("median"), {k: round(st[k], 4) for k in ("preprocess_fwd", "blend_fwd", "acc_zero", "blend_bwd", "preprocess_bwd", "block_lists") if k in st})
PY
  done
done
