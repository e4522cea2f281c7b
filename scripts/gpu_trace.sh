#!/bin/bash
# Kernel timeline of a few bench steps (start/end timestamps) for gap analysis.  Output: gpurun_out/trace/kernel_trace.csv
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 240 rocprofv3 --kernel-trace -d $OUT/kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs > $OUT/log.txt 2>&1; echo "trace rc=$?"
find $OUT/kt -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/kt
python3 - <<'PY'
import csv, os
f=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/trace/kernel_trace.csv'
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:60]) for r in csv.DictReader(open(f))]
rows.sort()
# last complete step: from the last preprocess_fwd to the following preprocess_bwd
idx=[i for i,r in enumerate(rows) if 'preprocess_fwd_kernel' in r[2]]
i0=idx[-2]; i1=idx[-1]
seg=rows[i0:i1]
busy=sum(e-s for s,e,_ in seg); span=seg[-1][1]-seg[0][0]
print('step span us',span/1e3,'busy us',busy/1e3,'kernels',len(seg))
prev=None
for s,e,n in seg:
    gap=(s-prev)/1e3 if prev else 0
    print(f'{gap:8.1f} gap  {(e-s)/1e3:8.1f} us  {n}')
    prev=e
PY
