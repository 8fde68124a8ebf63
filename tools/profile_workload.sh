#!/bin/bash
# rocprofv3 kernel-trace summary of one step of any bench workload (run through gpurun from the repository root):
#   bash tools/profile_workload.sh <workload> <tag>
# -> gpurun_out/profiles_<tag>/<tag>_rocprofv3_kernel_stats.csv (+ the bench line of the profiled run)
WL=$1
TAG=$2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_wl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wl -o bench -- python $ROOT/bench.py --workload "$WL" --steps 1 --warmup 1 --no-cpu-baseline --no-t-total 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json"
f=$(find /tmp/prof_wl -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$OUT/${TAG}_rocprofv3_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
head, body = rows[0], rows[1:]
# the synthetic generator's torch kernels (at::native, torch's own rocprim build 400001) run outside the timed region
keep = [r for r in body if "at::native" not in r[0] and "rocclr" not in r[0] and "ROCPRIM_400001" not in r[0]][:50]
csv.writer(open(sys.argv[2], "w")).writerows([head] + keep)
PY
