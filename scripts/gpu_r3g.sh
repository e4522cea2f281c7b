#!/bin/bash
# round 3: bench lines of C3..C5 (+ profiles), the train iteration with a per-kernel table
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in C3 C4 C5; do bash scripts/gpu_profile.sh r03 $c > gpurun_out/r3g_profile_$c.log 2>&1; tail -c 300 gpurun_out/profiles_r03_$c/bench.json | head -c 10 >/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/profiles_r03_$c/bench.json").read().strip().split("\n")[-1])
print("$c", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], {k: round(v,3) for k,v in d["stages_ms"].items()})
PY
done
echo "== train iteration"; timeout 600 python scripts/gpu_train_iter.py 2>&1 | tail -1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3g_train_prof --output-format csv -- python $GRAFT_REPO_ROOT/scripts/gpu_train_iter.py > $GRAFT_REPO_ROOT/gpurun_out/r3g_train_prof.log 2>&1)
find gpurun_out/r3g_train_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3g_train_kernel_stats.csv \;
rm -rf gpurun_out/r3g_train_prof
head -25 gpurun_out/r3g_train_kernel_stats.csv | cut -c1-160
