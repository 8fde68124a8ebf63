"""Parity on the BASELINE.json configurations at their real sensor sizes ("next round" item 1): every output of
Removerter::run() on the GPU against the CPU oracle (8 threads over keyframes: same serial arg-min semantics per keyframe).

  configs[0]  ParkingLot 01 vs 02, 50 keyframes each, os1-64 (64 x 1024 rays), single-res, THROUGH THE FILE PROTOCOL (`ltm_run`)
  configs[1]  the same pair, 3-res selfRemovert (in memory; 2 x 50 keyframes so that the oracle finishes in about a minute)
  configs[3]  KITTI-scale hdl-64e (64 x 1900 rays) on the `street` scene, whole pipeline (not only the vote)
  configs[4]  dense MLS parameters: 0.1 m voxels, k = 2, thr = 0.04, 8192-column scans (rows reduced), `street` scene

Labels are implied exact by identical point sets; XYZ is compared bitwise in memory and within the north star's 1e-4 m through
the file protocol (the host inverts the poses itself)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_clouds_equal
from test_gpu_pipeline import _compare, _run_gpu

pytestmark = pytest.mark.gpu

ORACLE_THREADS = 8


def _gen_device():
    """the synthetic generator is device-agnostic torch code: ray casting 65 k - 200 k rays per keyframe is seconds on the GPU"""
    import torch
    return "cuda" if torch.cuda.is_available() else "cpu"


@pytest.fixture(scope="module")
def lot50():
    from tools import synth
    return [synth.to_numpy(synth.make_session(s, 50, "os1-64", device=_gen_device())) for s in (1, 2)]


def test_config0_2x50_os1_64_single_res_through_ltm_run(tmp_path, orc, lot50):
    import fileproto as fp
    exe = os.path.join(ROOT, "lt-mapper_amd", "host", "ltm_run")
    assert os.path.exists(exe), "build the host mirror first (make host)"
    n_kf = 50
    dirs = fp.write_session_dirs(tmp_path, lot50)
    outdir = tmp_path / "out"
    yaml = tmp_path / "params.yaml"
    yaml.write_text(fp.yaml_text(tmp_path, dirs, outdir, 0, n_kf - 1))
    r = subprocess.run([exe, str(yaml)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    c_kf = fp.parse_keyframes(n_kf, 0, n_kf - 1)
    assert c_kf == list(range(n_kf))
    q_kf = fp.query_keyframes_in_roi(lot50[0], c_kf, lot50[1], n_kf)
    assert len(q_kf) >= 20, "sessions 01 and 02 start 37 m apart on the same loop: about half of the query keyframes are within 10 m of a central pose"
    C, Q = fp.host_load(orc, lot50[0], c_kf), fp.host_load(orc, lot50[1], q_kf)
    assert int(C["offsets"][-1]) > 2_000_000, "os1-64 scans: the 0.05 m VoxelGrid takes its int32-overflow early-out (A.6)"
    ref = orc.pipeline_run(orc.make_params(k=2, knn_thr=0.01, threads=ORACLE_THREADS), C, Q)
    fp.compare_output_tree(outdir, ref, [lot50[0]["names"][k] for k in c_kf], assert_clouds_equal)
    n_map = len(fp.read_pcd(str(outdir / "OriginalNoisyCentralMapGlobal.pcd"))[1])
    assert n_map > 500_000
    assert "T_total" in r.stdout, "ltm_run reports files -> files wall time"


def test_config1_2x50_os1_64_three_res(ltm, orc, lot50):
    res = (2.5, 2.0, 1.5)
    C, Q = ({k: v for k, v in S.items()} for S in lot50)
    for S in (C, Q):           # Step 0 in memory: pre-clean only (the loader's VoxelGrid is a no-op at this size)
        pts, off = [], [0]
        for k in range(len(S["offsets"]) - 1):
            p = orc.preclean(S["scans"][int(S["offsets"][k]):int(S["offsets"][k + 1])], 2.5)
            pts.append(p); off.append(off[-1] + len(p))
        S["scans"], S["offsets"] = np.concatenate(pts), np.array(off, np.uint64)
    ref = orc.pipeline_run(orc.make_params(k=2, knn_thr=0.01, use_self_removert=True, res_list=res, threads=ORACLE_THREADS), C, Q)
    ctx, rmv = _run_gpu(ltm, C, Q, gpu_use_self_removert=True, remove_resolution_list=list(res))
    _compare(rmv, ref)
    assert len(rmv.outputs["central_map_dynamic"]) > 0 and len(rmv.outputs["weak_pd_map"]) > 0
    ctx.close()


def test_config3_hdl64e_street_pipeline(ltm, orc):
    from tools import synth
    # sessions 01 / 02 start 37 m apart on the same road: 40 keyframes at 1 m overlap only partly => large PD / ND sets as well
    C, Q = (synth.to_numpy(synth.make_session(s, 40, "hdl-64e", scene="street", kf_spacing=1.0, device=_gen_device())) for s in (1, 2))
    ref = orc.pipeline_run(orc.make_params(k=2, knn_thr=0.01, threads=ORACLE_THREADS), C, Q)
    ctx, rmv = _run_gpu(ltm, C, Q)
    _compare(rmv, ref)
    assert len(rmv.outputs["OriginalNoisyCentralMapGlobal"]) > 500_000
    ctx.close()


def test_config4_mls_parameters_pipeline(ltm, orc):
    """configs[4]: downsample_voxel_size 0.1, k = 2, thr = 0.04 (search radius 0.283 m ~ 2.8 voxels), 8192-column scans"""
    from tools import synth
    sensor = (48, 8192, -25.0, 25.0)          # mls-128x8192 with the rows reduced: 393 216 rays per scan
    C, Q = (synth.to_numpy(synth.make_session(s, 12, sensor, scene="street", kf_spacing=2.0, device=_gen_device())) for s in (1, 2))
    ref = orc.pipeline_run(orc.make_params(k=2, knn_thr=0.04, voxel=0.1, threads=ORACLE_THREADS), C, Q)
    ctx, rmv = _run_gpu(ltm, C, Q, num_nn_points_within=2, dist_nn_points_within=0.04, downsample_voxel_size=0.1)
    _compare(rmv, ref)
    ctx.close()
