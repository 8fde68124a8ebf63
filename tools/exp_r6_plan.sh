#!/bin/bash
# EXPERIMENT (round 6): planned votes (ltm_vote_plan_begin) -- new tests, the existing parity tests through the planned path, A/B of the step
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp25; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_vote_plan.py -x -q > $OUT/tests_plan.txt 2>&1; tail -15 $OUT/tests_plan.txt
timeout 1500 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_pipeline.py tests/test_gpu_cli.py tests/test_gpu_vs_ref_compiled.py -x -q > $OUT/tests_parity.txt 2>&1; tail -5 $OUT/tests_parity.txt
for PLAN in 1; do
  for L in 2 1; do
    LTM_VOTE_PLAN=$PLAN python bench.py --steps 10 --warmup 3 --lanes $L --no-cpu-baseline --no-t-total --extra-out $OUT/plan${PLAN}_l${L}_extra.json 2>$OUT/plan${PLAN}_l${L}.err | tail -1 > $OUT/plan${PLAN}_l${L}.json
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r6_exp25/plan?_l?.json")):
    try:
        d = json.loads(open(f).read())
        print(os.path.basename(f), d["ms_per_step"], "one-lane", d.get("one_lane_ms_per_step"), [(c["c"], c["ms"]) for c in d["classes"][:8]])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
