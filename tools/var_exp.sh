mkdir -p gpurun_out/r2f
for i in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2f/plain_$i.json; done
for i in 1 2 3; do HSA_ENABLE_INTERRUPT=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2f/poll_$i.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2f/*.json")):
    d=json.loads(open(f).read())
    k=d["kernel_classes_ms_per_step"]
    print(f.split("/")[-1], d["ms_per_step"], "cls_sum", round(sum(k.values()),1), {x:k[x] for x in ("reproject_gather","voxel","voxel_scanset","partition","vote_map_cull")})
PY
