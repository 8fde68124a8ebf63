#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-t-total --extra-out $OUT/trace_extra.json 2>/dev/null | tail -1 > $OUT/trace_line.json
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$OUT/trace_min.csv" <<'PY'
import csv, sys, re
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Kernel_Name", "Queue_Id", "Start_Timestamp", "End_Timestamp"])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "at::native" in n or "ROCPRIM_400001" in n or "rocclr" in n or "Cijk_" in n: continue
    n = re.sub(r"^void\s+", "", n)
    if "rocprim" in n:
        n = "rocprim::" + next((a for a in ("radix_sort_onesweep", "radix_sort_histogram", "radix_sort_block_sort", "merge_sort_block_merge", "merge_sort_block_sort", "lookback_scan_state", "scan", "transform", "partition", "select") if a in n), "other")
    else:
        n = n.split("(")[0]
    w.writerow([n, r.get("Queue_Id", "0"), r["Start_Timestamp"], r["End_Timestamp"]])
PY
gzip -f $OUT/trace_min.csv
ls -la $OUT
