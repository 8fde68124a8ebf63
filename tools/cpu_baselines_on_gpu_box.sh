#!/bin/bash
# Round 6: the sampled single-thread CPU legs of configs[2..4] timed on the GPU box's own host (gpurun), with samples small enough for the
# GPU budget.  Results land in gpurun_out/cpu_baselines/ (copy into profiles/: bench.py quotes profiles/cpu_baseline_<workload>.json).
OUT=gpurun_out/cpu_baselines; mkdir -p $OUT
nproc > $OUT/host.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null
for WS in street-2x200-mls-knn:40 lot-cascade-6x500:50 street-2x2000-hdl64e-1res:100 street-2x2000-hdl64e-3res:100; do
  W=${WS%%:*}; S=${WS##*:}
  python tools/cpu_baseline_sampled.py --workload $W --stride $S --out $OUT/cpu_baseline_$W.json > $OUT/$W.log 2> $OUT/$W.err
  tail -c 400 $OUT/$W.log; echo
done
