#!/bin/bash
# One GPU-box session, parameterised (replaces the per-run scripts of rounds 2 and 3).  Everything lands in gpurun_out/<tag>_*.
#   scripts/gpu_session.sh <tag> <step> [<step> ...]
# steps:  cold[:N]        N fresh processes of tests/cold_first_launch.py (default 12; the first is the box's first GPU process)
#         pytest[:expr]   the GPU test-suite (-k expr)
#         smoke           __graft_entry__.smoke()
#         bench[:args]    bench.py with extra args (commas for spaces), e.g. bench:--config,C4   bench:--flags,both
#         env:K=V         export K=V for the steps that follow (env:K= unsets)
#         ab:name         RADEGS_LIB=gpurun_ab/libradegs_<name>.so for the steps that follow (ab:base = in-tree)
#         prof[:args]     rocprofv3 --kernel-trace --stats of bench.py (10 steps) -> gpurun_out/<tag>_prof/
#         pmc[:args]      scripts/gpu_pmc.sh (separate --pmc passes)
#         py:script[,args]  python scripts/<script> args
set -u
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
n=0
for step in "$@"; do
  n=$((n + 1))
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  args=${arg//,/ }
  case $kind in
    cold)
      N=${arg:-12}
      for i in $(seq 1 $N); do timeout 300 python tests/cold_first_launch.py g_C1 --tag ${TAG} > gpurun_out/${TAG}_cold_last.log 2>&1 || { echo "COLD MISMATCH (process $i)"; cat gpurun_out/${TAG}_cold_last.log | tail -5; }; done
      python - <<PY
import json
rows = [json.loads(l) for l in open("gpurun_out/cold_first_launch.jsonl") if l.startswith("{")]
rows = [r for r in rows if r.get("tag") == "${TAG}"]
open("gpurun_out/cold_${TAG}.jsonl", "w").write("".join(json.dumps(r) + "\n" for r in rows))   # per call: the merge back does not append
print("cold: %d processes, %d clean, first_on_box in %d, mismatches: %s" % (len(rows), sum(r["ok"] for r in rows), sum(r["first_on_box"] for r in rows), [r.get("differs") for r in rows if not r["ok"]]))
PY
      ;;
    pytest)
      if [ -n "$arg" ]; then timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 -k "$args" > gpurun_out/${TAG}_pytest_$n.log 2>&1; else timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 > gpurun_out/${TAG}_pytest_$n.log 2>&1; fi
      echo "pytest rc=$?"; tail -4 gpurun_out/${TAG}_pytest_$n.log ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.log ;;
    bench)
      timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $args > gpurun_out/${TAG}_bench_$n.log 2>&1
      tail -1 gpurun_out/${TAG}_bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench[$n] ${RADEGS_LIB:-intree} $args', d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['stages_ms'].items() if v})" 2>/dev/null || tail -3 gpurun_out/${TAG}_bench_$n.log ;;
    benchfull)
      timeout 900 python bench.py $args > gpurun_out/${TAG}_benchfull_$n.log 2>&1; tail -1 gpurun_out/${TAG}_benchfull_$n.log | cut -c1-400 ;;
    env) k=${arg%%=*}; v=${arg#*=}; if [ -z "$v" ]; then unset $k; else export $k="$v"; fi ;;
    ab) if [ "$arg" = base ]; then unset RADEGS_LIB; else export RADEGS_LIB=$PWD/gpurun_ab/libradegs_$arg.so; fi ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$n -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs $args > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$n.log 2>&1)
      echo "prof rc=$?"; f=$(find gpurun_out/${TAG}_prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -d, -f1-5 | cut -c1-150 ;;
    pmc) bash scripts/gpu_pmc.sh $args ;;
    py) s=${arg%%,*}; a=""; [ "$s" != "$arg" ] && a=${arg#*,}; timeout 1200 python scripts/$s ${a//,/ } > gpurun_out/${TAG}_py_$n.log 2>&1; echo "py $s rc=$?"; tail -12 gpurun_out/${TAG}_py_$n.log ;;
    *) echo "unknown step $step" ;;
  esac
done
