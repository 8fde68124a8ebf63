#!/bin/bash
# Collects everything profiles/ holds for one round on the GPU box (run through gpurun from the repository root):
#   bash tools/collect_profiles.sh <tag> [parts]
# parts (default "trace pmc line sq"): space-separated subset of
#   trace        rocprofv3 --kernel-trace --stats of one step of the default workload
#   pmc          three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU) of the default workload -> pmc_latest.json
#   line         the default bench line exactly as the driver runs it (--gpus 1 --steps 20 --warmup 5)
#   sq           SQ issue / wait counters of the heaviest kernels + the VALU issue-rate micro-benchmark
#   w:<workload> for a non-default workload: kernel trace, the three --pmc passes -> pmc_<workload>.json, then its bench line
# Results land in gpurun_out/profiles_<tag>/ (merged back by gpurun); copy what should be judged into profiles/
# (pmc_*.json keep their names: bench.py reads them and checks the kernel-source hash and the workload inside).
# Counter passes are separate runs with --pmc only (no tracing domains), as the pool requires.
TAG=${1:-rX}
PARTS=${2:-"trace pmc line sq"}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py"
COMMIT=$(cat "$ROOT/.commit_for_profiles" 2>/dev/null || echo unknown)

kernel_trace() {   # $1 = workload, $2 = output stem
  rm -rf /tmp/prof_kt
  # round 6: --lanes 1 -- per-kernel durations and counters are taken from the ONE-lane order (launches that overlap count each other's time); the bench line itself runs two lanes
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o bench -- $BENCH --workload $1 --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-t-total \
      --extra-out "$OUT/${2}_bench_under_rocprof_extra.json" 2>/dev/null | tail -1 > "$OUT/${2}_bench_under_rocprof.json"
  f=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$OUT/${2}_rocprofv3_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
head, body = rows[0], rows[1:]
# the synthetic generator's torch kernels (at::native, torch's own rocprim build 400001) run outside the timed region
keep = [r for r in body if "at::native" not in r[0] and "rocclr" not in r[0] and "ROCPRIM_400001" not in r[0]][:50]
csv.writer(open(sys.argv[2], "w")).writerows([head] + keep)
PY
}

pmc_passes() {   # $1 = workload, $2 = output file
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    rm -rf /tmp/prof_$C
    rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$C -o bench -- $BENCH --workload $1 --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-t-total --extra-out /tmp/extra_$C.json > /dev/null 2>&1
  done
  python3 - "$2" "$ROOT" "$COMMIT" "$1" <<'PY'
import csv, glob, json, sys, collections
sys.path.insert(0, sys.argv[2])
import bench
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: {"sum": 0.0, "dispatches": 0}))
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    for f in glob.glob(f"/tmp/prof_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            # every kernel of the step is kept, rocPRIM's included; the synthetic generator's torch kernels (at::native, torch's own rocprim
            # build 400001) run outside the timed region
            if r["Counter_Name"] != c or "at::native" in k or "ROCPRIM_400001" in k or "rocclr" in k or "Cijk_" in k: continue
            if "rocprim" in k:       # template soup: keep the algorithm and the key/value types
                algo = next((a for a in ("radix_sort_onesweep", "radix_sort_histogram", "radix_sort", "merge_sort_block_merge", "merge_sort_block_sort", "merge_sort",
                                         "lookback_scan_state", "scan", "transform", "partition", "select") if a in k), "other")
                k = "rocprim::" + algo + ("<u64,u32>" if "unsigned long, unsigned int" in k else "<u64>" if "unsigned long" in k else "")
            else:
                k = k.split("(")[0]
            e = agg[k][c]; e["sum"] += float(r["Counter_Value"]); e["dispatches"] += 1
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU, one pass each, on `python bench.py --workload W --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-t-total` (the one-lane order: non-overlapped launches); "
               "FETCH/WRITE are in KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read stream); "
               "SQ_INSTS_VALU counts wave instructions (x64 lanes); sums over all dispatches of each kernel = per step",
       "workload": sys.argv[4], "kernels_sha": bench.kernels_sha(), "commit": sys.argv[3], "all_kernels": agg}
json.dump(out, open(sys.argv[1], "w"), indent=1)
PY
}

for PART in $PARTS; do
  case $PART in
    trace) kernel_trace lot-2x500-os1-64-3res "${TAG}" ;;
    pmc)
      pmc_passes lot-2x500-os1-64-3res "$OUT/pmc_latest.json"
      cp "$OUT/pmc_latest.json" "$OUT/${TAG}_pmc.json"
      cp "$OUT/pmc_latest.json" "$ROOT/profiles/pmc_latest.json" ;;
    line)
      # the default bench line exactly as the driver runs it, after the counter passes so that it carries `traffic` and the VALU figures of THIS kernel source
      ( cd "$ROOT" && python3 bench.py --gpus 1 --steps 20 --warmup 5 --extra-out "$OUT/${TAG}_bench_extra.json" ) > /tmp/bench_stdout.txt 2>/tmp/bench_stderr.txt
      tail -c 8000 /tmp/bench_stdout.txt | tail -1 > "$OUT/${TAG}_bench_as_driver_runs_it.json"
      python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read()); print('line bytes', len(json.dumps(d)), 'value', d['value'], 'frac', d['roofline']['frac'])" "$OUT/${TAG}_bench_as_driver_runs_it.json" ;;
    sq)
      (cd "$ROOT" && bash tools/pmc_sq.sh) > "$OUT/${TAG}_pmc_sq.txt" 2>&1
      [ -x "$ROOT/tools/ubench/valu_rate" ] && "$ROOT/tools/ubench/valu_rate" > "$OUT/${TAG}_valu_rate_ubench.txt" 2>&1 ;;
    w:*)
      W=${PART#w:}
      kernel_trace "$W" "${TAG}_${W}"
      pmc_passes "$W" "$OUT/pmc_${W}.json"
      cp "$OUT/pmc_${W}.json" "$ROOT/profiles/pmc_${W}.json"
      ( cd "$ROOT" && python3 bench.py --workload "$W" --steps 2 --warmup 1 --extra-out "$OUT/${TAG}_bench_${W}_extra.json" ) 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_${W}.json" ;;
  esac
done
ls -la "$OUT"
