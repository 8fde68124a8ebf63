#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp1; mkdir -p $OUT
./tools/ubench/stream_priority > $OUT/stream_priority.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for MODE in plain overlap; do
  FLAG=""; [ $MODE = overlap ] && FLAG="--overlap-sessions"
  rm -rf /tmp/kt_$MODE
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$MODE -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-t-total $FLAG --extra-out $OUT/${MODE}_extra.json 2>/dev/null | tail -1 > $OUT/${MODE}_line.json
  f=$(find /tmp/kt_$MODE -name "*kernel_trace.csv" | head -1)
  head -1 $f > $OUT/${MODE}_trace_head.txt
  python3 $ROOT/tools/trace_timeline.py $f --bin-us 1000 > $OUT/${MODE}_timeline.txt 2>&1
done
cd $ROOT
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-t-total --extra-out $OUT/plain10_extra.json 2>/dev/null | tail -1 > $OUT/plain10_line.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-t-total --overlap-sessions --extra-out $OUT/overlap10_extra.json 2>/dev/null | tail -1 > $OUT/overlap10_line.json
ls -la $OUT
