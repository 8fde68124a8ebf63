#!/bin/bash
# End-to-end parity of the street workloads on the two-lane path (through gpurun): every output of run() against the oracle, bitwise --
# configs[3] geometry at 2x200 keyframes single-res and (config 33) 3-res, configs[4] (full MLS sensor) at 2x20 keyframes.
for C in 3 33 4; do
  python tools/parity_fullsize.py --config $C > gpurun_out/r6_parity_fullsize_config${C}_two_lanes.json 2> gpurun_out/r6_parity_fullsize_config${C}.err
  python - gpurun_out/r6_parity_fullsize_config${C}_two_lanes.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], {k: v for k, v in d.items() if not isinstance(v, (list, dict)) and k != "what"})
PY
done
