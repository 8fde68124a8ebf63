#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp5; mkdir -p $OUT
LTM_VOTE_PERSIST=8 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lanes.py -x -q -k "vote or lanes" > $OUT/pytest_persist.txt 2>&1
tail -3 $OUT/pytest_persist.txt
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-t-total --profile-steps 1"
for P in 0 8 7 6 5; do
  LTM_VOTE_PERSIST=$P $B --lanes 1 --extra-out $OUT/l1_p$P.json 2>/dev/null | tail -1 > $OUT/l1_p$P.line
  LTM_VOTE_PERSIST=$P $B --lanes 2 --extra-out $OUT/l2_p$P.json 2>/dev/null | tail -1 > $OUT/l2_p$P.line
done
for f in $OUT/*.line; do python3 -c "
import json,sys
d=json.load(open('$f')); e=json.load(open('$f'.replace('.line','.json')))
print('$f'.split('/')[-1], d['ms_per_step'], 'vote_ms', e['kernel_classes_ms_per_step'].get('vote_map_cull'), e.get('timed_region_stage_ms'))"; done
