"""The C++ host's keyframe sharding (lt-mapper_amd/host/removert/Comm.h): `ltm_run --logical-ranks K`
runs K ranks -- one host thread and one device context each, all on the one GPU of the test box -- that vote / reproject / search
only their block of keyframes and exchange label masks, per-rank voxel-grid pieces and scan shards through the Comm interface.
Every output file must be BYTE-IDENTICAL to the single-rank run for K in {1, 2, 3, 4, 8} (even K: with and without the session groups), and `--gpus 1` must do the same through the
RCCL back end (ncclCommInitAll with one device: the plumbing the 8-GPU node uses).  The same comparison covers SURVEY 8f-1 / 8f-2: the
pipelined loader + background writer (default) must write exactly what the synchronous path writes."""
import filecmp
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _tree(root):
    out = []
    for d, _, files in os.walk(root):
        for f in files:
            out.append(os.path.relpath(os.path.join(d, f), root))
    return sorted(out)


@pytest.mark.parametrize("three_res", [False, True])
def test_sharded_host_outputs_do_not_depend_on_rank_count(tmp_path, three_res):
    import fileproto as fp
    from tools import synth
    exe = os.path.join(ROOT, "lt-mapper_amd", "host", "ltm_run")
    n_kf = 21                                     # not divisible by 2, 4 or 8: unequal keyframe blocks
    sess = [synth.to_numpy(synth.make_session(s, n_kf, "tiny")) for s in (1, 2)]
    dirs = fp.write_session_dirs(tmp_path, sess)
    extra = "  gpu_use_self_removert: true\n" if three_res else ""
    res = (2.5, 2.0, 1.5) if three_res else (2.5,)
    env = dict(os.environ, LTM_VOXEL_SHARD_MIN="0")     # shard every voxel grid, however small
    runs = {}
    # "sync": the synchronous loader / writer (removert/gpu_async_io: false) against the pipelined feeder + background writer (default)
    # even K: makeGlobalMap + Step 1 run on two session groups of K/2 ranks and rank pairs swap the results (Comm.h); k3 and the
    # LTM_SESSION_GROUPS=0 runs keep the unsplit sharding covered
    for tag, args in (("single", []), ("sync", []), ("k1", ["--logical-ranks", "1"]), ("k2", ["--logical-ranks", "2"]), ("k3", ["--logical-ranks", "3"]),
                      ("k4", ["--logical-ranks", "4"]), ("k8", ["--logical-ranks", "8"]), ("k2_unsplit", ["--logical-ranks", "2"]),
                      ("k4_unsplit", ["--logical-ranks", "4"]), ("rccl1", ["--gpus", "1"])):
        outdir = tmp_path / f"out_{tag}"
        yaml = tmp_path / f"params_{tag}.yaml"
        yaml.write_text(fp.yaml_text(tmp_path, dirs, outdir, 0, n_kf - 1, res_list=res, extra=extra + ("  gpu_async_io: false\n" if tag == "sync" else "")))
        r = subprocess.run([exe, str(yaml)] + args, capture_output=True, text=True, timeout=600, env=dict(env, LTM_SESSION_GROUPS="0") if tag.endswith("_unsplit") else env)
        assert r.returncode == 0, f"{tag}: " + r.stdout[-1500:] + r.stderr[-1500:]
        assert "T_total" in r.stdout
        runs[tag] = outdir
    ref_files = _tree(runs["single"])
    assert len(ref_files) >= 14 + 5 * n_kf
    for tag, outdir in runs.items():
        if tag == "single":
            continue
        assert _tree(outdir) == ref_files, f"{tag}: different set of output files"
        for f in ref_files:
            assert filecmp.cmp(os.path.join(runs["single"], f), os.path.join(outdir, f), shallow=False), f"{tag}: {f} differs from the single-rank run"
