#!/bin/bash
# timing experiments: which part of blend_bwd costs what
for ppl in 4 2; do for dbg in 0 1 2 3; do
  RADEGS_BWD_PPL=$ppl RADEGS_BWD_DBG=$dbg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ppl',$ppl,'dbg',$dbg, d['stages_ms']['blend_bwd'])"
done; done
