#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp7; mkdir -p $OUT
timeout 1700 python -m pytest tests/test_gpu_cli.py tests/test_gpu_vs_ref_compiled.py tests/test_gpu_lanes.py -x -q > $OUT/pytest_cli.txt 2>&1
tail -15 $OUT/pytest_cli.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --extra-out $OUT/bench_extra.json 2>$OUT/bench_err.txt | tail -1 > $OUT/bench_line.json
python3 -c "
import json
d=json.load(open('$OUT/bench_line.json')); print(d['value'], d['ms_per_step'], 'cxx', d.get('cxx_host_ms_per_step'), 'one-lane', d.get('one_lane_ms_per_step'), d.get('t_total_s'))"
tail -3 $OUT/bench_err.txt
