#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp14; mkdir -p $OUT
python -m pytest tests/test_gpu_fullsize.py -x -q -k "two_phase_knn" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --lanes 1 --no-cpu-baseline --no-t-total --extra-out $OUT/extra.json 2>/dev/null | tail -1 | cut -c1-200
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
grep -E "k_scan_rimg|k_image_max|k_scan_qbound|k_knn|k_fill_u32" $f | cut -d, -f1-4 | sed 's/(.*"/"/' | cut -c1-120
cd $ROOT
for V in "LTM_KNN_COOP=1 LTM_KNN_QUEUE_FROM_PHASE1=1" "LTM_KNN_COOP=0 LTM_KNN_QUEUE_FROM_PHASE1=0" "LTM_KNN_COOP=1 LTM_KNN_QUEUE_FROM_PHASE1=0" "LTM_KNN_COOP=0 LTM_KNN_QUEUE_FROM_PHASE1=1"; do
  env $V python bench.py --steps 5 --warmup 2 --lanes 1 --no-cpu-baseline --no-t-total --extra-out $OUT/e.json 2>/dev/null | tail -1 > $OUT/l.json
  python3 -c "
import json
d=json.load(open('$OUT/l.json')); e=json.load(open('$OUT/e.json')); k=e['kernel_classes_ms_per_step']
print('$V', d['ms_per_step'], {x:k[x] for x in ('knn_build','knn_query','knn_query_p2','vote_scan')})"
done
