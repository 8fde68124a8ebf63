#!/bin/bash
# Collects everything profiles/ holds for one round on the GPU box (run through gpurun from the repository root):
#   bash tools/collect_profiles.sh <tag> [quick]
# Results land in gpurun_out/profiles_<tag>/ (merged back by gpurun); copy what should be judged into profiles/
# (pmc_latest.json keeps its name: bench.py reads it and checks the kernel-source hash inside).
# Counter passes are separate runs with --pmc only (no tracing domains), as the pool requires.
TAG=${1:-rX}
QUICK=${2:-}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py"
COMMIT=$(cat "$ROOT/.commit_for_profiles" 2>/dev/null || echo unknown)

# 1. kernel trace + stats of one step
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o bench -- $BENCH --steps 1 --warmup 0 --no-cpu-baseline --no-t-total 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json"
f=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$OUT/${TAG}_rocprofv3_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
head, body = rows[0], rows[1:]
# the synthetic generator's torch kernels (at::native, torch's own rocprim build 400001) run outside the timed region
keep = [r for r in body if "at::native" not in r[0] and "rocclr" not in r[0] and "ROCPRIM_400001" not in r[0]][:50]
csv.writer(open(sys.argv[2], "w")).writerows([head] + keep)
PY

# 2. HBM traffic (FETCH_SIZE, WRITE_SIZE) and VALU instruction count, each in its own pass
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  rm -rf /tmp/prof_$C
  rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$C -o bench -- $BENCH --steps 1 --warmup 0 --no-cpu-baseline --no-t-total 2>/dev/null | tail -1 > /tmp/bench_$C.json
done
python3 - "$OUT/pmc_latest.json" "$ROOT" "$COMMIT" <<'PY'
import csv, glob, json, sys, collections
sys.path.insert(0, sys.argv[2])
import bench
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: {"sum": 0.0, "dispatches": 0}))
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    for f in glob.glob(f"/tmp/prof_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            # every kernel of the step is kept, rocPRIM's included (the sort passes had no traffic evidence); the synthetic
            # generator's torch kernels (at::native, torch's own rocprim build 400001) run outside the timed region
            if r["Counter_Name"] != c or "at::native" in k or "ROCPRIM_400001" in k or "rocclr" in k or "Cijk_" in k: continue
            if "rocprim" in k:       # template soup: keep the algorithm and the key/value types
                algo = next((a for a in ("radix_sort_onesweep", "radix_sort_histogram", "radix_sort", "merge_sort_block_merge", "merge_sort_block_sort", "merge_sort",
                                         "lookback_scan_state", "scan", "transform", "partition", "select") if a in k), "other")
                k = "rocprim::" + algo + ("<u64,u32>" if "unsigned long, unsigned int" in k else "<u64>" if "unsigned long" in k else "")
            else:
                k = k.split("(")[0]
            e = agg[k][c]; e["sum"] += float(r["Counter_Value"]); e["dispatches"] += 1
dom = next((k for k in agg if "k_vote_map_cull" in k), None)
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU, one pass each, on `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-t-total`; "
               "FETCH/WRITE are in KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read stream); "
               "SQ_INSTS_VALU counts wave instructions (x64 lanes / point-projections = VALU instructions per point); sums over all dispatches of each kernel",
       "workload": bench.DEFAULT_WORKLOAD, "kernels_sha": bench.kernels_sha(), "commit": sys.argv[3], "dominant_kernel": "k_vote_map_cull"}
if dom:
    d = agg[dom]; n = d["FETCH_SIZE"]["dispatches"] or 1
    out.update(dispatches=n, fetch_kib_sum=d["FETCH_SIZE"]["sum"], write_kib_sum=d["WRITE_SIZE"]["sum"],
               hbm_bytes_per_launch=(2.0 * d["FETCH_SIZE"]["sum"] + d["WRITE_SIZE"]["sum"]) * 1024.0 / n)
    try:
        line = json.load(open("/tmp/bench_SQ_INSTS_VALU.json"))
        units = next(r["units_per_step"] for r in line["rooflines"] if r["class"] == "vote_map_cull")
        out.update(valu_wave_insts_sum=d["SQ_INSTS_VALU"]["sum"], point_projections=units, valu_insts_per_point=d["SQ_INSTS_VALU"]["sum"] * 64.0 / units)
    except Exception as e:
        out["valu_error"] = repr(e)
out["all_kernels"] = agg
json.dump(out, open(sys.argv[1], "w"), indent=1)
PY
cp "$OUT/pmc_latest.json" "$OUT/${TAG}_pmc.json"

# 3. the default bench line (CPU baseline on) -- after the counter passes, with their result in place, so that the line carries
#    `traffic` and the VALU figures of THIS kernel source (bench.py checks the hash inside pmc_latest.json)
cp "$OUT/pmc_latest.json" "$ROOT/profiles/pmc_latest.json"
$BENCH --cpu-allcore 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_default.json"
python3 - "$OUT/${TAG}_bench_default.json" "$OUT/cpu_allcore_latest.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
a = (d.get("cpu_baseline") or {}).get("all_cores")
if a and a.get("value"):
    json.dump(a, open(sys.argv[2], "w"), indent=1)
PY

# 4. SQ issue / wait counters of the heaviest kernels, and the VALU issue-rate micro-benchmark (make ubench)
(cd "$ROOT" && bash tools/pmc_sq.sh) > "$OUT/${TAG}_pmc_sq.txt" 2>&1
[ -x "$ROOT/tools/ubench/valu_rate" ] && "$ROOT/tools/ubench/valu_rate" > "$OUT/${TAG}_valu_rate_ubench.txt" 2>&1

# 5. the other configurations (one run each)
if [ -z "$QUICK" ]; then
  $BENCH --workload street-2x2000-hdl64e-1res --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_street_2x2000_hdl64e.json"
  $BENCH --workload street-2x2000-hdl64e-3res --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_street_2x2000_hdl64e_3res.json"
  $BENCH --workload street-2x200-mls-knn --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_street_2x200_mls.json"
  $BENCH --workload lot-cascade-6x500 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_lot_cascade_6x500.json"
fi
ls -la "$OUT"
