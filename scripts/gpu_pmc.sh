#!/bin/bash
# SQ counters of the kernels whose name contains $1 (default: blend), three counter sets in three runs
set -u
export PMC_FILTER=${1:-blend}
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$name --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/pmc/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_ACTIVE_INST_MISC
python3 - <<'PY'
import csv, glob, collections, os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc'
for run in ['sq1','sq2','sq3']:
    fs=glob.glob(f'{root}/{run}/*/*_counter_collection.csv')
    if not fs: print(run,'no csv'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name']
        if os.environ.get('PMC_FILTER','blend') in k: agg[k.split('(')[0][-44:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(run,k,{c: round(sum(x)/len(x)) for c,x in v.items()})
PY
