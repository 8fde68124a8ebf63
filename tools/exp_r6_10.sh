#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_exp10; mkdir -p $OUT
python - <<'PY'
import os, sys, tempfile
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from tools import synth
import fileproto as fp
n_kf = 500
sess = [synth.to_numpy(synth.make_session(s, n_kf, "os1-64", device="cuda")) for s in (1, 2)]
root = "/tmp/ltm_exp10"
os.makedirs(root, exist_ok=True)
dirs = fp.write_session_dirs(root, sess)
open(os.path.join(root, "p.yaml"), "w").write(fp.yaml_text(root, dirs, os.path.join(root, "out"), 0, n_kf - 1, res_list=(2.5, 2.0, 1.5), extra="  gpu_use_self_removert: true\n  gpu_lanes: LANES\n".replace("LANES", os.environ.get("EXP_LANES", "2"))))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
LTM_BENCH_NO_PROFILE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o bench -- $ROOT/lt-mapper_amd/host/ltm_run /tmp/ltm_exp10/p.yaml --bench 1 --warmup 2 > $OUT/cxx_stdout.txt 2>$OUT/cxx_stderr.txt
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$OUT/cxx_trace_min_l${EXP_LANES:-2}.csv" <<'PY'
import csv, sys, re
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Kernel_Name", "Queue_Id", "Start_Timestamp", "End_Timestamp"])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    n = re.sub(r"^void\s+", "", n)
    if "rocprim" in n:
        n = "rocprim::" + next((a for a in ("radix_sort_onesweep", "radix_sort_histogram", "radix_sort_block_sort", "merge_sort_block_merge", "merge_sort_block_sort", "lookback_scan_state", "scan", "transform", "partition", "select") if a in n), "other")
    else:
        n = n.split("(")[0]
    w.writerow([n, r.get("Queue_Id", "0"), r["Start_Timestamp"], r["End_Timestamp"]])
PY
gzip -f $OUT/cxx_trace_min_l${EXP_LANES:-2}.csv
grep "\[bench\]" $OUT/cxx_stdout.txt | cut -c1-200
cd $ROOT
rm -rf /tmp/ltm_exp10
