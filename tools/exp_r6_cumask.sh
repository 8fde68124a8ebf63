#!/bin/bash
# EXPERIMENT (round 6): the projection launches of a lane family on ONE stream whose CU mask leaves some compute units out (LTM_HEAVY_CU_RESERVE), so that the other
# lane's small kernels always find free CUs.  Micro-benchmark first, then the step.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp27; mkdir -p $OUT
./tools/ubench/stream_priority > $OUT/stream_priority_with_cu_masks.txt 2>&1
tail -30 $OUT/stream_priority_with_cu_masks.txt
for CFG in "0 0" "16 0" "32 0" "64 0" "16 1" "32 1" "64 1"; do
  set -- $CFG
  LTM_HEAVY_CU_RESERVE=$1 LTM_HEAVY_CU_LAYOUT=$2 python bench.py --steps 10 --warmup 3 --lanes 2 --no-cpu-baseline --no-t-total --extra-out $OUT/r$1_l$2_extra.json 2>/dev/null | tail -1 > $OUT/r$1_l$2.json
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r6_exp27/r*_l?.json")):
    try:
        d = json.loads(open(f).read()); e = json.load(open(f.replace(".json", "_extra.json")))
        print(os.path.basename(f), d["ms_per_step"], "one-lane", d.get("one_lane_ms_per_step"), e.get("timed_region_stage_ms"))
    except Exception as ex:
        print(f, "ERR", ex)
PY
# and: the one-pass scan-image kernel on the street (1.8 points per pixel): never (0) against always (1e9), one-lane class times
for D in 0 1e9; do
  LTM_SCAN_MULTI_MAX_DENSITY=$D python bench.py --workload street-2x2000-hdl64e-3res --steps 3 --warmup 1 --lanes 2 --no-cpu-baseline --no-t-total --extra-out $OUT/street3_d${D}_extra.json 2>/dev/null | tail -1 > $OUT/street3_d${D}.json
  LTM_SCAN_MULTI_MAX_DENSITY=$D python bench.py --workload lot-cascade-6x500 --steps 3 --warmup 1 --lanes 2 --no-cpu-baseline --no-t-total --extra-out $OUT/cascade_d${D}_extra.json 2>/dev/null | tail -1 > $OUT/cascade_d${D}.json
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r6_exp27/*_d*.json")):
    if f.endswith("_extra.json"): continue
    try:
        d = json.loads(open(f).read())
        vs = next((c for c in d["classes"] if c["c"] == "vote_scan"), {})
        print(os.path.basename(f), d["ms_per_step"], "one-lane", d.get("one_lane_ms_per_step"), "vote_scan", vs.get("ms"))
    except Exception as ex:
        print(f, "ERR", ex)
PY
