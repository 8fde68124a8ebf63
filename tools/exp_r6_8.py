"""ltm_run --bench on two lanes / one lane, with and without the HIP-event profile (round-6 experiment)"""
import json, os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from tools import synth, t_total
import fileproto as fp
n_kf = 500
sess = [synth.to_numpy(synth.make_session(s, n_kf, "os1-64", device="cuda")) for s in (1, 2)]
root = tempfile.mkdtemp(prefix="ltm_exp8_")
dirs = fp.write_session_dirs(root, sess)
out = {}
for lanes in (1, 2):
    for noprof in (0, 1):
        if noprof: os.environ["LTM_BENCH_NO_PROFILE"] = "1"
        else: os.environ.pop("LTM_BENCH_NO_PROFILE", None)
        r = t_total.bench_cxx_host(root, dirs, n_kf, three_res=True, steps=5, warmup=2, lanes=lanes)
        out[f"lanes{lanes}_noprof{noprof}"] = r["ms_per_step"]
        print(f"lanes {lanes} profile {'off' if noprof else 'on'}: {r['ms_per_step']:.2f} ms", flush=True)
shutil.rmtree(root, ignore_errors=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r6_exp8.json"), "w"))
