#!/bin/bash
# Last pass of a round on the GPU box (run through gpurun from the repository root), at sources that will not change any more: the GPU suite (which writes the
# full-size parity record that bench.py checks against the source hashes), the default bench line exactly as the driver runs it, and the lines of configs[2..4]
# (their counter files come from tools/collect_profiles.sh "w:<workload>").  bash tools/final_lines.sh <round tag, e.g. r6>
# Results under gpurun_out/profiles_<tag>_final2/; copy what should be judged to profiles/.
TAG=${1:-rX}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/profiles_${TAG}_final2; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tee $OUT/${TAG}_pytest_gpu.log | grep -i "passed\|failed\|error" | tail -3
cp gpurun_out/parity_fullsize_2x500_3res.json profiles/parity_fullsize_2x500_3res.json 2>/dev/null
( python3 bench.py --gpus 1 --steps 20 --warmup 5 --extra-out "$OUT/${TAG}_final_bench_extra.json" ) > /tmp/bench_stdout.txt 2>/tmp/bench_stderr.txt
tail -c 8000 /tmp/bench_stdout.txt | tail -1 > "$OUT/${TAG}_final_bench_as_driver_runs_it.json"
python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read()); print('line bytes', len(json.dumps(d)), 'value', d['value'], 'frac', d['roofline']['frac'], 'parity', d['parity_fullsize_matches_sources'], d['t_total_s'])" "$OUT/${TAG}_final_bench_as_driver_runs_it.json"
for W in street-2x2000-hdl64e-1res street-2x2000-hdl64e-3res street-2x200-mls-knn lot-cascade-6x500; do
  python3 bench.py --workload $W --steps 2 --warmup 1 --extra-out "$OUT/${TAG}_final_bench_${W}_extra.json" 2>/dev/null | tail -1 > "$OUT/${TAG}_final_bench_${W}.json"
  python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['frac'], (d['cpu_baseline'] or {}).get('value'))" "$OUT/${TAG}_final_bench_${W}.json" $W
done
