#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp15; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PT in 1 0; do
rm -rf /tmp/kt
LTM_SCAN_PRETEST=$PT rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --lanes 1 --no-cpu-baseline --no-t-total --extra-out $OUT/extra.json 2>/dev/null | tail -1 | cut -c1-100
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if any(k in r[0] for k in ("k_scan_rimg","k_image_max","k_scan_qbound")): print("   ", r[0].split("(")[0][:40], "calls", r[1], "total ms", round(float(r[2])/1e6,3), "avg us", round(float(r[3])/1e3,1))
PY
done
