#!/bin/bash
# EXPERIMENT (round 6): where does k_vote_plan_build spend its time?  LTM_PLAN_DBG 1 = stop after phase 1, 2 = no record stores (results wrong: timing only)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp26; mkdir -p $OUT
for DBG in 0 1 2 16; do
  LTM_VOTE_PLAN_SKIP=$(( DBG / 16 )) LTM_PLAN_DBG=$(( DBG % 16 )) LTM_VOTE_PLAN_FRACTION=0.4 python bench.py --steps 4 --warmup 2 --lanes 1 --no-cpu-baseline --no-t-total --extra-out $OUT/dbg${DBG}_extra.json 2>/dev/null | tail -1 > $OUT/dbg${DBG}.json
done
python - <<'PY'
import json
for dbg in (0, 1, 2, 16):
    d = json.load(open(f"gpurun_out/r6_exp26/dbg{dbg}_extra.json"))
    k = d["kernel_classes_ms_per_step"]
    print(dbg, d["ms_per_step"], "build", k.get("vote_plan_build"), "replay", k.get("vote_replay"), "cull", k.get("vote_map_cull"), d.get("vote_plans"), d.get("vote_cull"))
PY
