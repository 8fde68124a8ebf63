#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp18; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_lanes.py -x -q 2>&1 | tail -3
for DM in 1024 0 512 256; do
  LTM_VOTE_DENSE_MIN=$DM python bench.py --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --no-t-total --extra-out $OUT/e.json 2>/dev/null | tail -1 > $OUT/l.json
  python3 -c "
import json
d=json.load(open('$OUT/l.json')); e=json.load(open('$OUT/e.json')); k=e['kernel_classes_ms_per_step']
print('dense_min $DM:', d['ms_per_step'], 'ms/step; vote_map_cull', k['vote_map_cull'])"
done
