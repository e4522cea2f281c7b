#!/bin/bash
for dbg in 0 4 1 2 3; do
  RADEGS_BWD_PPL=2 RADEGS_BWD_DBG=$dbg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bwd ppl 2 dbg',$dbg, d['stages_ms']['blend_bwd'])"
done
for ppl in 1 2; do for dbg in 0 3; do
  RADEGS_FWD_PPL=$ppl RADEGS_FWD_DBG=$dbg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd ppl',$ppl,'dbg',$dbg, d['stages_ms']['blend_fwd'])"
done; done
RADEGS_BWD_DBG=4 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "small_scene or C1" 2>&1 | tail -2
