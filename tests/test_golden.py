"""Committed golden fixtures (tests/golden/*.npz, made by tools/make_golden.py).  Since round 4 they are outputs of REFERENCE-COMPILED
code (oracle/_ref: the reference's unmodified sources built against stand-in ROS / Eigen / OpenCV / PCL headers; the fixture files name their
source), generated in the build container and committed because /root/reference does not exist on the GPU box.  On CPU the oracle must
reproduce them; on the GPU the HIP path must reproduce them with neither the oracle nor the reference in the loop."""
import os

import numpy as np
import pytest

from conftest import assert_clouds_equal

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["tiny_pair_1res", "tiny_pair_3res"]


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def _sessions(d):
    return (dict(scans=d["c_scans"], offsets=d["c_off"], poses=d["c_poses"], inv=d["c_inv"]),
            dict(scans=d["q_scans"], offsets=d["q_off"], poses=d["q_poses"], inv=d["q_inv"]))


def test_oracle_reproduces_primitive_golden(orc):
    d = _load("primitives")
    rimg, idx = orc.range_image(d["pts"], 50.0, 360.0, 125, 900, d["Tinv"], None)
    assert (rimg.view(np.uint32) == d["rimg"].view(np.uint32)).all() and (idx == d["idx"]).all()
    assert_clouds_equal(orc.voxel_centroid(d["pts"], 0.5), d["vox"], "voxel")
    assert (orc.knn_split(d["tgt"], d["pts"][:1000] * 0.1, 2, 0.05) == d["near"]).all()
    assert (orc.knn_split(d["tgt"], (d["pts"][:1000] * 0.5).astype(np.float32), 2, 1.0) == d["near_k2_thr1"]).all()   # Session.cpp:452-484, reference-compiled
    assert "reference-compiled" in str(d["source"])


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_pipeline_golden(orc, case):
    d = _load(case)
    n_kf, three_res, k = (int(x) for x in d["meta"])
    C, Q = _sessions(d)
    res = (2.5, 2.0, 1.5) if three_res else (2.5,)
    r = orc.pipeline_run(orc.make_params(k=k, knn_thr=float(d["thr"]), use_self_removert=bool(three_res), res_list=res), C, Q)
    for key in d.files:
        if key.startswith("map_"):
            assert_clouds_equal(r.cloud(key[4:]), d[key], key)
        elif key.startswith("scan_"):
            pts, off = r.scanset(key[5:])
            assert (off == d["off_" + key[5:]]).all()
            assert_clouds_equal(pts, d[key], key)


@pytest.mark.gpu
def test_gpu_reproduces_primitive_golden(gpu_ctx):
    d = _load("primitives")
    cloud = gpu_ctx.upload(d["pts"])
    rimg, idx = gpu_ctx.debug_range_image(cloud, 2.5, d["Tinv"], None)
    assert (rimg.view(np.uint32) == d["rimg"].view(np.uint32)).all() and (idx == d["idx"]).all()
    assert_clouds_equal(gpu_ctx.voxel_centroid(cloud, 0.5).download(), d["vox"], "voxel")
    q = d["pts"][:1000] * 0.1
    near, far = gpu_ctx.knn_split_cloud(gpu_ctx.upload(d["tgt"]), gpu_ctx.upload(q), 2, 0.05)
    assert_clouds_equal(near.download(), q[d["near"] == 1], "near")
    q2 = (d["pts"][:1000] * 0.5).astype(np.float32)
    near2, far2 = gpu_ctx.knn_split_cloud(gpu_ctx.upload(d["tgt"]), gpu_ctx.upload(q2), 2, 1.0)
    assert_clouds_equal(near2.download(), q2[d["near_k2_thr1"] == 1], "weak -> strong ND rule (reference-compiled fixture)")
    assert_clouds_equal(far2.download(), q2[d["near_k2_thr1"] == 0], "weak -> strong ND rule, kept weak")


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_reproduces_pipeline_golden(ltm, case):
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    d = _load(case)
    n_kf, three_res, k = (int(x) for x in d["meta"])
    C, Q = _sessions(d)
    ctx = ltm.Context(vfov=50.0, hfov=360.0, device=0)
    P = Params(num_nn_points_within=k, dist_nn_points_within=float(d["thr"]), gpu_use_self_removert=bool(three_res),
               remove_resolution_list=[2.5, 2.0, 1.5] if three_res else [2.5])
    sessions = [Session(n, ctx.upload_scans(S["scans"], S["offsets"]), ctx.poses(S["poses"], S["inv"])) for n, S in (("Central", C), ("Query", Q))]
    rm = Removerter(HipOps(ctx), P, *sessions)
    rm.run()
    for key in d.files:
        if key.startswith("map_"):
            assert_clouds_equal(rm.outputs[key[4:]].download(), d[key], key)
    for name, ss in rm.scan_outputs().items():
        pts, off = ss.download()
        assert (off == d["off_" + name]).all()
        assert_clouds_equal(pts, d["scan_" + name], name)
    ctx.close()
