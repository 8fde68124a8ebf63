#!/bin/bash
# EXPERIMENT (round 6): does capping the resident vote workgroups per CU (extra dynamic LDS) let the other lane's small kernels run under a vote?
# pad 0 = 8 workgroups/CU (18 KB each); 8192 -> 6/CU; 14336 -> 5/CU; 22528 -> 4/CU.  One-lane ms (does the vote itself slow down?) and two-lane ms per pad.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp21; mkdir -p $OUT
for PAD in 0 8192 14336 22528; do
  for L in 2 1; do
    LTM_VOTE_LDS_PAD=$PAD python bench.py --steps 10 --warmup 3 --lanes $L --no-cpu-baseline --no-t-total --extra-out $OUT/pad${PAD}_l${L}_extra.json 2>/dev/null | tail -1 > $OUT/pad${PAD}_l${L}.json
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r6_exp21/pad*_l?.json")):
    try:
        d = json.loads(open(f).read())
        vc = next((c for c in d["classes"] if c["c"] == "vote_map_cull"), {})
        print(os.path.basename(f), d["ms_per_step"], "one-lane", d.get("one_lane_ms_per_step"), "vote_map_cull ms", vc.get("ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
