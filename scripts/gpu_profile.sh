#!/bin/bash
# Produces the artefacts that go under profiles/ for one workload: bench line, rocprofv3 kernel stats, PMC counters (separate passes).
#   scripts/gpu_profile.sh <tag> [config] [suffix] [extra bench args...]
#   e.g.  r04 C2                         -> gpurun_out/profiles_r04/
#         r04 C4                         -> gpurun_out/profiles_r04_C4/
#         r04 C2 both --flags both       -> gpurun_out/profiles_r04_C2_both/
#         r04 C2 both_fwd --flags both --mode forward
# then scripts/summarize_profiles.py <tag>[_<config>][_<suffix>] copies the summaries into profiles/.
set -u
TAG=${1:-r04}
CFG=${2:-C2}
SFX=${3:-}
shift; shift; [ $# -gt 0 ] && shift
EXTRA_ARGS="$*"
SUF=""; [ "$CFG" != "C2" ] && SUF="_$CFG"
[ -n "$SFX" ] && SUF="_${CFG}_${SFX}"
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG$SUF
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="$GRAFT_REPO_ROOT/bench.py --config $CFG $EXTRA_ARGS"
NOCPU=""; { [ "$CFG" != "C2" ] || [ -n "$SFX" ]; } && NOCPU="--no-cpu-baseline"
python $B --steps 20 --warmup 10 $NOCPU > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python $B --steps 20 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/rocprof_bench.log 2>&1; echo "rocprof rc=$?"
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc --output-format csv -- python $B --steps 9 --warmup 9 --no-cpu-baseline --no-other-configs > $OUT/pmc_tcc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES -d $OUT/pmc_sq --output-format csv -- python $B --steps 9 --warmup 9 --no-cpu-baseline --no-other-configs > $OUT/pmc_sq.log 2>&1
# third pass (round 6): the kernel's own clock (GRBM_GUI_ACTIVE, summed over the 8 XCDs) and the LDS side -- what scripts/valu_busy.py needs
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY -d $OUT/pmc_clk --output-format csv -- python $B --steps 9 --warmup 9 --no-cpu-baseline --no-other-configs > $OUT/pmc_clk.log 2>&1
find $OUT -name "*counter_collection.csv" | while read f; do d=$(basename $(dirname $(dirname $f))); cp $f $OUT/${d}_counters.csv; done
rm -rf $OUT/kt $OUT/pmc_tcc $OUT/pmc_sq $OUT/pmc_clk
ls $OUT
