#!/bin/bash
# The round's closing GPU session: cold-start processes, the whole GPU suite (with durations), smoke, the profiles of every workload
# (bench line + rocprofv3 kernel stats + PMC passes), the opt-in full-size oracle run, a wider randomised sweep, a kernel timeline.
#   scripts/gpu_final_round.sh <tag>      e.g. r05   -> gpurun_out/<tag>z_* and gpurun_out/profiles_<tag>*/
set -u
TAG=${1:-r05}
export TMPDIR=/tmp
bash scripts/gpu_session.sh ${TAG}z cold:16 pytest smoke
bash scripts/gpu_profile.sh $TAG C2
bash scripts/gpu_profile.sh $TAG C2 both --flags both
bash scripts/gpu_profile.sh $TAG C2 both_fwd --flags both --mode forward
bash scripts/gpu_profile.sh $TAG C3
bash scripts/gpu_profile.sh $TAG C4
bash scripts/gpu_profile.sh $TAG C5
bash scripts/gpu_profile.sh $TAG C2H
RADEGS_FULL_ORACLE=1 timeout 900 python -m pytest tests/test_gpu_full.py -m gpu -q -p no:cacheprovider -k full_size_oracle --durations=5 > gpurun_out/${TAG}_full_size_oracle.log 2>&1; echo "full-size oracle rc=$?"; tail -3 gpurun_out/${TAG}_full_size_oracle.log
timeout 900 python scripts/gpu_fuzz_table.py 0:120 > gpurun_out/${TAG}_fuzz_table_0_120.txt 2> gpurun_out/${TAG}_fuzz_table.err; echo "fuzz rc=$?"; tail -2 gpurun_out/${TAG}_fuzz_table_0_120.txt
bash scripts/gpu_trace.sh > gpurun_out/${TAG}_step_timeline.txt 2>&1; head -3 gpurun_out/${TAG}_step_timeline.txt
