#!/bin/bash
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters_list.txt 2>&1
run() { # name counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$name --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc/$name.log 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE GRBM_COUNT
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run tcc2 TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum
find $GRAFT_REPO_ROOT/gpurun_out/pmc -name "*.csv" | head -30
