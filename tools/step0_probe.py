"""Where Step 0 of a files -> files run goes: writes the 2 x 500 os1-64 lot sessions in the reference's on-disk format and runs `ltm_run` on them with
LTM_STEP0_TIMING=1 (per-stage laps of Step 0, the loader's wait / upload split and the thread time per file inside loadPCDFile and voxelGridFilter) and LTM_POOL_STATS=1,
once per setting of the swept variable (here: loader threads).  Run through gpurun; profiles/r5_step0_loader_threads.txt is its output, condensed."""
import os, sys, subprocess, tempfile, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import fileproto as fp
from tools import synth
n_kf = 500
sess = [synth.to_numpy(synth.make_session(s, n_kf, "os1-64", device="cuda:0")) for s in (1, 2)]
root = tempfile.mkdtemp(prefix="step0_")
dirs = fp.write_session_dirs(root, sess)
exe = os.path.join(os.getcwd(), "lt-mapper_amd", "host", "ltm_run")
import itertools
for r, mb in enumerate(("16", "64", "64", "32", "128")):
    out = os.path.join(root, f"out{r}"); shutil.rmtree(out, ignore_errors=True)
    y = os.path.join(root, "p.yaml")
    open(y, "w").write(fp.yaml_text(root, dirs, out, 0, n_kf - 1, res_list=(2.5, 2.0, 1.5), extra="  gpu_use_self_removert: true\n"))
    p = subprocess.run([exe, y], capture_output=True, text=True, env=dict(os.environ, LTM_STEP0_TIMING="1", LTM_POOL_STATS="1", LTM_LOADER_THREADS=str(mb)))
    print("run", r, "loader threads", mb, [l for l in p.stdout.splitlines() if l.startswith("[timing]")][-1][:200])
    print("\n".join(l for l in p.stderr.splitlines() if "step 0" in l or "pool" in l))
shutil.rmtree(root, ignore_errors=True)
