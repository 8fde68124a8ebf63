#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_lanes.py -x -q > $OUT/pytest_lanes.txt 2>&1
tail -3 $OUT/pytest_lanes.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-t-total --profile-steps 1"
$B --extra-out $OUT/a_extra.json 2>/dev/null | tail -1 > $OUT/a_chain_prio.json
LTM_HEAVY_PRIORITY=0 $B --extra-out $OUT/b_extra.json 2>/dev/null | tail -1 > $OUT/b_chain_only.json
LTM_HEAVY_CHAIN=0 $B --extra-out $OUT/c_extra.json 2>/dev/null | tail -1 > $OUT/c_prio_only.json
LTM_HEAVY_MIN_BLOCKS=0 $B --extra-out $OUT/d_extra.json 2>/dev/null | tail -1 > $OUT/d_chain_all.json
LTM_HEAVY_MIN_BLOCKS=400000 $B --extra-out $OUT/e_extra.json 2>/dev/null | tail -1 > $OUT/e_chain_400k.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-t-total --extra-out $OUT/trace_extra.json 2>/dev/null | tail -1 > $OUT/trace_line.json
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_timeline.py $f --bin-us 1000 --max-bins 700 > $OUT/timeline.txt 2>&1
cd $ROOT
for f in a_chain_prio b_chain_only c_prio_only d_chain_all e_chain_400k; do python3 -c "
import json,sys
d=json.load(open('$OUT/$f.json')); print('$f', d['value'], d['ms_per_step'], d.get('one_lane_ms_per_step'))
e=json.load(open('$OUT/'+'$f'[0]+'_extra.json')); print('   ', e.get('timed_region_stage_ms'))"; done
