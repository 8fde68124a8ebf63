#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp17; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py -x -q -k "occlusion or street or kitti" 2>&1 | tail -4
for ST in 1 0; do
  LTM_OCCLUSION_SUBTILE=$ST LTM_OCCLUSION_STATS=1 python bench.py --workload street-2x2000-hdl64e-1res --steps 2 --warmup 1 --lanes 1 --extra-out $OUT/street_st$ST.json 2>$OUT/street_st$ST.err | tail -1 > $OUT/street_st$ST.line
  python3 -c "
import json
d=json.load(open('$OUT/street_st$ST.line')); e=json.load(open('$OUT/street_st$ST.json')); k=e['kernel_classes_ms_per_step']
print('subtile $ST:', d['ms_per_step'], 'ms/step', {x:k[x] for x in ('reproject_map','vote_map_exact','reproject_gather','vote_map_cull')})"
  grep "occlusion" $OUT/street_st$ST.err | tail -3
done
