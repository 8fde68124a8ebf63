#!/bin/bash
# Kernel-time view of a cold `ltm_run` (files -> files) on the configs[1] input: where T_total goes beyond the device work.
#   bash tools/profile_ltm_run.sh [keyframes]      (through gpurun, from the repository root; writes gpurun_out/ltm_run_profile.txt)
KF=${1:-500}
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out"
OUT=$ROOT/gpurun_out/ltm_run_profile.txt
rm -rf /tmp/tt /tmp/prof_tt
python tools/t_total.py --kf $KF --three-res --runs 1 --keep /tmp/tt > /tmp/tt_first.json 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tt -o run -- $ROOT/lt-mapper_amd/host/ltm_run /tmp/tt/params.yaml > /tmp/tt_run.log 2>&1
{
  echo "== plain run"; tail -c 600 /tmp/tt_first.json; echo
  echo "== under rocprofv3 --kernel-trace"; grep "^\[timing\]" /tmp/tt_run.log
  f=$(find /tmp/prof_tt -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time sum %.1f ms over %d kernels, %d launches" % (tot / 1e6, len(rows), sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("  %8.2f ms  %5s calls  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], r["Name"][:90]))
PY
} > "$OUT" 2>&1
cat "$OUT"
