#!/bin/bash
# SQ issue / stall breakdown of the dominant kernels (rocprofv3 --pmc passes; counters only, no tracing).  Run through gpurun from the repo root.
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAVES"; do
  rm -rf /tmp/pmc_sq
  rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_sq -- python $ROOT/bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-t-total > /tmp/pmc_sq.log 2>&1
  f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "ltm::" not in r["Kernel_Name"]: continue      # the torch kernels of the synthetic generator are not ours
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:3]:
    a = agg[k]; w = a.get("SQ_WAVE_CYCLES", 1) or 1
    print(k, "launches", n[k], "SQ_WAVE_CYCLES %.4g" % w, {c: round(v / w, 4) for c, v in a.items() if c != "SQ_WAVE_CYCLES"})
PY
done
