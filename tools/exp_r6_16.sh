#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp16; mkdir -p $OUT
for W in lot-cascade-6x500 street-2x200-mls-knn street-2x2000-hdl64e-1res street-2x2000-hdl64e-3res; do
  for L in 2 1; do
    python bench.py --workload $W --steps 2 --warmup 1 --lanes $L --profile-steps 1 --extra-out $OUT/${W}_l$L.json 2>$OUT/${W}_l$L.err | tail -1 > $OUT/${W}_l$L.line
    python3 -c "
import json
d=json.load(open('$OUT/${W}_l$L.line')); e=json.load(open('$OUT/${W}_l$L.json'))
print('$W lanes $L:', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step; one-lane pass', d.get('one_lane_ms_per_step'), e.get('timed_region_stage_ms'))" 2>&1 | tail -1
  done
done
