#!/bin/bash
# SQ stall breakdown of the dominant kernels (one rocprofv3 --pmc pass; counters only, no tracing)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS \
  --output-format csv -d /tmp/pmc_sq -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc_sq.log 2>&1
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:6]:
    a = agg[k]; w = a.get("SQ_WAVE_CYCLES", 1) or 1
    print(k, "launches", n[k], {c: round(v / w, 3) for c, v in a.items() if c != "SQ_WAVE_CYCLES"}, "valu_per_wavecycle", round(a.get("SQ_INSTS_VALU", 0) / w, 3))
PY
