#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp2; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_lanes.py -x -q > $OUT/pytest_lanes.txt 2>&1
tail -5 $OUT/pytest_lanes.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-t-total --extra-out $OUT/lanes2_extra.json 2>$OUT/lanes2_err.txt | tail -1 > $OUT/lanes2_line.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-t-total --lanes 1 --extra-out $OUT/lanes1_extra.json 2>/dev/null | tail -1 > $OUT/lanes1_line.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-t-total --extra-out $OUT/trace_extra.json 2>/dev/null | tail -1 > $OUT/trace_line.json
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_timeline.py $f --bin-us 1000 --max-bins 700 > $OUT/lanes2_timeline.txt 2>&1
ls -la $OUT
