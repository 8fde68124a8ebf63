#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp13; mkdir -p $OUT
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 2 --extra-out $OUT/bench_extra.json 2>/dev/null | tail -1 > $OUT/bench_line.json
python3 -c "
import json
d=json.load(open('$OUT/bench_line.json')); e=json.load(open('$OUT/bench_extra.json'))
print(d['value'], d['ms_per_step'], 'cxx', d.get('cxx_host_ms_per_step'), 'one-lane', d.get('one_lane_ms_per_step'), d.get('t_total_s'))
print(e['timed_region_stage_ms']); print({k:v for k,v in e['kernel_classes_ms_per_step'].items()})"
