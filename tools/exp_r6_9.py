"""pool growth per step on two lanes (round-6 experiment): hipMalloc counts of ltm_run --bench for 3 and 8 timed steps"""
import json, os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from tools import synth
import fileproto as fp
n_kf = 500
sess = [synth.to_numpy(synth.make_session(s, n_kf, "os1-64", device="cuda")) for s in (1, 2)]
root = tempfile.mkdtemp(prefix="ltm_exp9_")
dirs = fp.write_session_dirs(root, sess)
exe = os.path.join(ROOT, "lt-mapper_amd", "host", "ltm_run")
for lanes in (2,):
    for steps in (4,):
        yaml = os.path.join(root, "p.yaml")
        open(yaml, "w").write(fp.yaml_text(root, dirs, os.path.join(root, "out"), 0, n_kf - 1, res_list=(2.5, 2.0, 1.5), extra=f"  gpu_use_self_removert: true\n  gpu_lanes: {lanes}\n" + os.environ.get("EXP_EXTRA_YAML", "")))
        env = dict(os.environ, LTM_POOL_STATS="1", LTM_BENCH_NO_PROFILE="1", LTM_STAGE_TIMING="1")
        p = subprocess.run([exe, yaml, "--bench", str(steps), "--warmup", "2"], capture_output=True, text=True, env=env)
        line = [l for l in p.stdout.splitlines() if l.startswith("[bench] ")][-1]
        print("lanes", lanes, "steps", steps, "ms_per_step", json.loads(line[8:])["ms_per_step"])
        for l in p.stderr.splitlines():
            if "device pool" in l: print("   ", l)
        tl = [l for l in p.stderr.splitlines() if "two lanes:" in l]
        for l in tl[-5:]: print("   ", l)
shutil.rmtree(root, ignore_errors=True)
