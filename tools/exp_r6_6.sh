#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp6; mkdir -p $OUT
LTM_SORT_SLIM=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lanes.py -x -q -k "voxel or lanes or knn" > $OUT/pytest_slim.txt 2>&1
tail -3 $OUT/pytest_slim.txt
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-t-total --profile-steps 2"
for P in 0 1; do
  LTM_SORT_SLIM=$P $B --lanes 1 --extra-out $OUT/l1_s$P.json 2>/dev/null | tail -1 > $OUT/l1_s$P.line
  LTM_SORT_SLIM=$P $B --lanes 2 --extra-out $OUT/l2_s$P.json 2>/dev/null | tail -1 > $OUT/l2_s$P.line
done
for f in $OUT/*.line; do python3 -c "
import json,sys
d=json.load(open('$f')); e=json.load(open('$f'.replace('.line','.json')))
k=e['kernel_classes_ms_per_step']
print('$f'.split('/')[-1], d['ms_per_step'], 'one-lane', d.get('one_lane_ms_per_step'), 'voxel', k.get('voxel'), k.get('voxel_scanset'), 'knn_build', k.get('knn_build'), e.get('timed_region_stage_ms'))"; done
