#!/bin/bash
# SQ counters of the projection kernels under tools/ab_kernels.py (rocprofv3 --pmc passes; counters only, no tracing).  Run through gpurun from the repo root:
#   bash tools/pmc_kernels.sh "LTM_MAP_KERNEL=2" "LTM_MAP_KERNEL=4"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  echo "=== $V"
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES"; do
    rm -rf /tmp/pmc_k
    rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_k -- python $ROOT/tools/ab_kernels.py --env "$V" --rounds 1 --kf 200 > /tmp/pmc_k.log 2>&1
    f=$(find /tmp/pmc_k -name "*counter_collection.csv" | head -1)
    [ -z "$f" ] && { tail -5 /tmp/pmc_k.log; continue; }
    python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_map_rimg" not in k and "k_vote_map_cull" not in k: continue
    k = k.split("(")[0][-44:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k in sorted(agg):
    a = agg[k]; w = a.get("SQ_WAVE_CYCLES", 1) or 1
    print(" ", k, "launches", n[k], "WAVE_CYCLES %.4g" % w, {c: (round(v / w, 4) if not c.startswith("SQ_INSTS") and c != "SQ_WAVES" else "%.4g" % v) for c, v in a.items() if c != "SQ_WAVE_CYCLES"})
PY
  done
done
