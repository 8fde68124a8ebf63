#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp11; mkdir -p $OUT
python -m pytest tests/test_gpu_lanes.py -x -q 2>&1 | tail -2
for LF in 0 1 2 3; do
  echo "== LTM_LANE_FRIENDLY=$LF"
  LTM_LANE_FRIENDLY=$LF python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-t-total --profile-steps 1 --extra-out $OUT/py_$LF.json 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('python two-lane', d['ms_per_step'], 'one-lane pass', d['one_lane_ms_per_step'])"
  python3 -c "import json; e=json.load(open('$OUT/py_$LF.json')); print('   ', e['timed_region_stage_ms'])"
done
LTM_LANE_FRIENDLY=0 python tools/exp_r6_9.py 2>&1 | grep -E "ms_per_step|stage"
LTM_LANE_FRIENDLY=3 python tools/exp_r6_9.py 2>&1 | grep -E "ms_per_step|stage"
