"""Real multi-GPU runs: these tests need TWO OR MORE devices and skip themselves on the one-GPU boxes this
repository is developed on, so the first node that has them exercises RCCL over xGMI without anyone editing the suite.

  * `ltm_run --gpus K` (C++ host, one host thread + context per GPU, RcclComm: all-reduce of label masks, all-gather-v, key-range all-to-all-v) must
    write byte for byte what the single-GPU run writes -- with the session groups (two extra communicators + pair swap, opt-in over RCCL:
    LTM_SESSION_GROUPS=1) and without;
  * `bench.py --gpus K` (Python host, torch.distributed "nccl" = RCCL, one process per GPU) must run and print a parseable line."""
import filecmp
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _device_count():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _worlds():
    n = _device_count()
    return [k for k in (2, 4, 8) if k <= n]


def _tree(root):
    return sorted(os.path.relpath(os.path.join(d, f), root) for d, _, files in os.walk(root) for f in files)


@pytest.mark.parametrize("three_res", [False, True])
def test_ltm_run_on_k_gpus_writes_what_one_gpu_writes(tmp_path, three_res):
    worlds = _worlds()
    if not worlds:
        pytest.skip(f"needs >= 2 GPUs, this box has {_device_count()}")
    import fileproto as fp
    from tools import synth
    exe = os.path.join(ROOT, "lt-mapper_amd", "host", "ltm_run")
    n_kf = 21                                     # not divisible by 2, 4 or 8: unequal keyframe blocks
    sess = [synth.to_numpy(synth.make_session(s, n_kf, "tiny")) for s in (1, 2)]
    dirs = fp.write_session_dirs(tmp_path, sess)
    extra = "  gpu_use_self_removert: true\n" if three_res else ""
    res = (2.5, 2.0, 1.5) if three_res else (2.5,)
    base = dict(os.environ, LTM_VOXEL_SHARD_MIN="0", HSA_ENABLE_IPC_MODE_LEGACY="0")     # shard every voxel grid, however small
    runs = {}
    cases = [("single", [], {})]
    for k in worlds:
        cases.append((f"rccl{k}", ["--gpus", str(k)], {}))                                   # unsplit sharding (the default over RCCL)
        cases.append((f"rccl{k}_groups", ["--gpus", str(k)], {"LTM_SESSION_GROUPS": "1"}))   # one rank group per session in Step 1 + pair swap
    for tag, args, env in cases:
        outdir = tmp_path / f"out_{tag}"
        yaml = tmp_path / f"params_{tag}.yaml"
        yaml.write_text(fp.yaml_text(tmp_path, dirs, outdir, 0, n_kf - 1, res_list=res, extra=extra))
        r = subprocess.run([exe, str(yaml)] + args, capture_output=True, text=True, timeout=900, env=dict(base, **env))
        assert r.returncode == 0, f"{tag}: " + r.stdout[-1500:] + r.stderr[-1500:]
        runs[tag] = outdir
    ref_files = _tree(runs["single"])
    assert len(ref_files) >= 14 + 5 * n_kf
    for tag, outdir in runs.items():
        if tag == "single":
            continue
        assert _tree(outdir) == ref_files, f"{tag}: different set of output files"
        for f in ref_files:
            assert filecmp.cmp(os.path.join(runs["single"], f), os.path.join(outdir, f), shallow=False), f"{tag}: {f} differs from the single-GPU run"


def test_bench_py_runs_on_two_gpus_over_rccl():
    if _device_count() < 2:
        pytest.skip(f"needs >= 2 GPUs, this box has {_device_count()}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    lines = {}
    for tag, extra_env in (("unsplit", {}), ("groups", {"LTM_SESSION_GROUPS": "1"})):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "lot-2x100-small-3res",
                            "--no-cpu-baseline", "--no-t-total", "--extra-out", os.devnull], capture_output=True, text=True, timeout=900, env=dict(env, **extra_env))
        assert r.returncode == 0, f"{tag}: " + r.stdout[-1500:] + r.stderr[-1500:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
        lines[tag] = d
    assert "rank group" in lines["groups"]["config"]["parallelism"] and "rank group" not in lines["unsplit"]["config"]["parallelism"]
