python tools/parity_sampled.py --config 3 > gpurun_out/r6_parity_sampled_street_2x2000_hdl64e_32kf.json 2> gpurun_out/r6_parity_sampled_3.err
python tools/parity_sampled.py --config 4 > gpurun_out/r6_parity_sampled_street_2x200_mls_32kf.json 2> gpurun_out/r6_parity_sampled_4.err
python tools/parity_fullsize.py --config 2 --sessions 6 --kf 120 > gpurun_out/r6_parity_lot_cascade_6x120_3res_two_lanes.json 2> gpurun_out/r6_parity_cascade6.err
python tools/fuzz_parity.py --n 24 --seed 67 > gpurun_out/r6_fuzz_parity_24_cases_final.json 2> gpurun_out/r6_fuzz.err
python tools/fuzz_parity.py --n 12 --seed 78 --se3 > gpurun_out/r6_fuzz_parity_se3_12_cases.json 2> gpurun_out/r6_fuzz_se3.err
for f in gpurun_out/r6_parity_sampled_street_2x2000_hdl64e_32kf.json gpurun_out/r6_parity_sampled_street_2x200_mls_32kf.json gpurun_out/r6_parity_lot_cascade_6x120_3res_two_lanes.json gpurun_out/r6_fuzz_parity_24_cases_final.json gpurun_out/r6_fuzz_parity_se3_12_cases.json; do python - $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], {k: v for k, v in d.items() if not isinstance(v, (list, dict)) and k != "what"})
PY
done
