#!/bin/bash
# The last GPU pass of round 6 (through gpurun, from the repository root): the suite + the lines at the final sources (tools/final_lines.sh), then -- first pass only --
# the counter files of the street 3-res workload, sampled parity of configs[3] / [4] at their stated size, the cascade 01 -> 02 -> 03 at 500 keyframes on two lanes, a fuzz run;
# and the one-shot files -> files A/B of gpu_lanes 1 against 2.
bash tools/final_lines.sh r6 > gpurun_out/r6_final_lines.log 2>&1; tail -8 gpurun_out/r6_final_lines.log
if [ "$1" = "all" ]; then
  bash tools/collect_profiles.sh r6_final "w:street-2x2000-hdl64e-3res" > gpurun_out/r6_collect_final_c.log 2>&1
  python tools/parity_sampled.py --config 3 > gpurun_out/r6_parity_sampled_street_2x2000_hdl64e_32kf.json 2> gpurun_out/r6_parity_sampled_3.err
  python tools/parity_sampled.py --config 4 > gpurun_out/r6_parity_sampled_street_2x200_mls_32kf.json 2> gpurun_out/r6_parity_sampled_4.err
  python tools/parity_fullsize.py --config 2 --sessions 3 > gpurun_out/r6_parity_fullsize_lot_cascade_3x500_3res_two_lanes.json 2> gpurun_out/r6_parity_cascade.err
  python tools/fuzz_parity.py --n 24 --seed 66 > gpurun_out/r6_fuzz_parity_24_cases.json 2> gpurun_out/r6_fuzz.err
fi
python tools/t_total_lanes_ab.py > gpurun_out/r6_ttotal_one_shot_lanes_1_vs_2.json 2> gpurun_out/r6_ttotal_ab.err; tail -c 1500 gpurun_out/r6_ttotal_one_shot_lanes_1_vs_2.json
