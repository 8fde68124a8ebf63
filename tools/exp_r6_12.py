"""one-shot ltm_run files -> files on one and two lanes, with stage timing (round-6 experiment)"""
import json, os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from tools import synth
import fileproto as fp
n_kf = 500
sess = [synth.to_numpy(synth.make_session(s, n_kf, "os1-64", device="cuda")) for s in (1, 2)]
root = tempfile.mkdtemp(prefix="ltm_exp12_")
dirs = fp.write_session_dirs(root, sess)
exe = os.path.join(ROOT, "lt-mapper_amd", "host", "ltm_run")
for rep in range(2):
  for lanes in (1, 2):
    out = os.path.join(root, "out")
    yaml = os.path.join(root, "p.yaml")
    open(yaml, "w").write(fp.yaml_text(root, dirs, out, 0, n_kf - 1, res_list=(2.5, 2.0, 1.5), extra=f"  gpu_use_self_removert: true\n  gpu_lanes: {lanes}\n"))
    env = dict(os.environ, LTM_POOL_STATS="1", LTM_STAGE_TIMING="1", LTM_STEP0_TIMING="1")
    p = subprocess.run([exe, yaml], capture_output=True, text=True, env=env)
    t = [l for l in p.stdout.splitlines() if l.startswith("[timing]")]
    print("lanes", lanes, t[-1] if t else p.stderr[-500:])
    for l in p.stderr.splitlines():
        if "two lanes:" in l or "device pool" in l or "step 0" in l: print("   ", l)
    shutil.rmtree(out, ignore_errors=True)
shutil.rmtree(root, ignore_errors=True)
