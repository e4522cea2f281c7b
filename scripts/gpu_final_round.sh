#!/bin/bash
# The round's closing GPU session (round 6): cold-start processes, smoke, the profiles of every workload (bench line + rocprofv3 kernel
# stats + three PMC passes incl. the kernel's own clock), a kernel timeline, the view-parallel exchange at N = 1 under the launcher.
# (The GPU suite, the arbiter tables and the gradient-parity file are separate calls: gpu_session.sh <tag> pytest, gpu_arbiter_table.py,
# gpu_grad_parity.py.)      scripts/gpu_final_round.sh <tag>      e.g. r06   -> gpurun_out/<tag>z_* and gpurun_out/profiles_<tag>*/
set -u
TAG=${1:-r06}
export TMPDIR=/tmp
bash scripts/gpu_session.sh ${TAG}z cold:12 smoke
bash scripts/gpu_profile.sh $TAG C2
bash scripts/gpu_profile.sh $TAG C2 both --flags both
bash scripts/gpu_profile.sh $TAG C2 both_fwd --flags both --mode forward
bash scripts/gpu_profile.sh $TAG C3
bash scripts/gpu_profile.sh $TAG C4
bash scripts/gpu_profile.sh $TAG C5
bash scripts/gpu_profile.sh $TAG C2H
bash scripts/gpu_trace.sh > gpurun_out/${TAG}_step_timeline.txt 2>&1; head -3 gpurun_out/${TAG}_step_timeline.txt
# the exchange's own timing on the N = 1 line, so that the first real SCALE run can be read against it
for x in factored allreduce; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 20 --warmup 5 \
    --force-allreduce --exchange $x --no-cpu-baseline --no-other-configs > gpurun_out/${TAG}_bench_force_exchange_$x.json 2> gpurun_out/${TAG}_bench_force_exchange_$x.err
  echo "force-exchange $x rc=$? $(tail -c 200 gpurun_out/${TAG}_bench_force_exchange_$x.json)"
done
