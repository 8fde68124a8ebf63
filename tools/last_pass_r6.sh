#!/bin/bash
# The last GPU passes of round 6 (through gpurun, from the repository root), at sources that do not change any more.
#   bash tools/last_pass_r6.sh lines     the suite + the full-size parity record + the driver-style line + the lines of configs[2..4] (tools/final_lines.sh), then the parity passes
#   bash tools/last_pass_r6.sh counters  kernel trace, the three counter passes and the SQ counters of the default workload, trace + counters of configs[2..4] (one-lane order)
if [ "$1" = "lines" ]; then
  bash tools/final_lines.sh r6 > gpurun_out/r6_final_lines.log 2>&1; tail -8 gpurun_out/r6_final_lines.log
  bash tools/parity_pass_r6.sh 2>&1 | tail -6
  bash tools/parity_pass_r6_streets.sh 2>&1 | tail -4
else
  bash tools/collect_profiles.sh r6_final "trace pmc sq w:lot-cascade-6x500 w:street-2x2000-hdl64e-1res w:street-2x2000-hdl64e-3res w:street-2x200-mls-knn" > gpurun_out/r6_collect_final.log 2>&1
  tail -5 gpurun_out/r6_collect_final.log; ls gpurun_out/profiles_r6_final | wc -l
fi
