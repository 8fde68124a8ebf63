#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp20; mkdir -p $OUT
LTM_SORT_MERGE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lanes.py -x -q -k "voxel or lanes" 2>&1 | tail -2
for SM in 0 1; do for L in 1 2; do
  LTM_SORT_MERGE=$SM python bench.py --steps 6 --warmup 2 --lanes $L --profile-steps 1 --no-cpu-baseline --no-t-total --extra-out $OUT/e.json 2>/dev/null | tail -1 > $OUT/l.json
  python3 -c "
import json
d=json.load(open('$OUT/l.json')); e=json.load(open('$OUT/e.json')); k=e['kernel_classes_ms_per_step']
print('merge-sort $SM lanes $L:', d['ms_per_step'], 'ms/step; voxel', k['voxel'], k['voxel_scanset'], e.get('timed_region_stage_ms'))"
done; done
