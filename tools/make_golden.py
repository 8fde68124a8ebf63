#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small seeded inputs + the outputs of REFERENCE-COMPILED code for them.

Since round 4 the fixtures come from oracle/_ref/libltm_ref.so -- the reference's own, unmodified sources
(/root/reference/ltremovert/src/{utility,RosParamServer,Session,Removerter}.cpp) compiled against the stand-in ROS / Eigen /
OpenCV / PCL headers of oracle/refshim/include and driven through oracle/refshim/ref_capi.cpp -- not from the oracle: every array
written here is what the reference's code produced in this container (serial build; the PCL / Eigen leaves are the stand-ins'
restatements).  While generating, the oracle is required to agree bit for bit.  tests/test_golden.py checks the
oracle (CPU) and the HIP path (GPU box, where /root/reference does not exist) against these committed numbers.
    python tools/make_golden.py          # needs /root/reference
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as orc  # noqa: E402
from oracle import ref_py as ref  # noqa: E402
from tools import synth  # noqa: E402

MAPS = ["OriginalNoisyCentralMapGlobal", "OriginalNoisyQueryMapGlobal", "central_map_static", "central_map_dynamic", "query_map_static",
        "query_map_dynamic", "central_sess_high_dyn", "query_sess_high_dyn", "union_map_queryside", "union_map_centralside", "pd_map",
        "nd_map", "strong_nd_map", "weak_nd_map", "strong_pd_map", "weak_pd_map", "updated_map", "updated_map_strong"]
SCANS = ["scans_updated", "scans_updated_strong", "scans_pd", "scans_pd_strong", "scans_nd_strong"]


def _same(a, b, what):
    a = np.asarray(a, np.float32).reshape(-1, 4); b = np.asarray(b, np.float32).reshape(-1, 4)
    assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all(), f"oracle and reference-compiled code disagree on {what}"


def pipeline_case(name, n_kf, sensor, three_res, k, thr):
    C = synth.to_numpy(synth.make_session(1, n_kf, sensor))
    Q = synth.to_numpy(synth.make_session(2, n_kf, sensor))
    for S in (C, Q):          # Session.cpp:110: the inverse poses are Eigen's Matrix4d::inverse() of the parsed poses (stand-in Eigen == oracle, bitwise)
        S["inv"] = ref.inverse4x4(S["poses"].reshape(-1, 16)).reshape(S["poses"].shape)
        assert (S["inv"].view(np.uint64) == orc.inverse_poses(S["poses"]).view(np.uint64)).all(), "inverse poses: oracle and stand-in Eigen disagree"
    res = (2.5, 2.0, 1.5) if three_res else (2.5,)
    R = ref.Removerter(ref.make_params(k=k, knn_thr=thr, use_self_removert=three_res, res_list=res)).pipeline_run(C, Q)
    r = orc.pipeline_run(orc.make_params(k=k, knn_thr=thr, use_self_removert=three_res, res_list=res), C, Q)
    d = {"meta": np.array([n_kf, int(three_res), k], dtype=np.int64), "thr": np.float32(thr),
         "source": np.array("oracle/_ref/libltm_ref.so: the reference's sources compiled against oracle/refshim (tools/make_golden.py)")}
    for tag, S in (("c", C), ("q", Q)):
        d[f"{tag}_scans"], d[f"{tag}_off"], d[f"{tag}_poses"], d[f"{tag}_inv"] = S["scans"], S["offsets"], S["poses"], S["inv"]
    state = {"central_map_static": (0, "static"), "central_map_dynamic": (0, "dynamic"), "query_map_static": (1, "static"), "query_map_dynamic": (1, "dynamic")}
    for m in MAPS:
        c = R.session_map(*state[m]) if m in state else R.cloud(m)
        o = r.cloud(m)
        assert (c is None) == (o is None), m
        if c is not None:
            _same(c, o, m)
            d["map_" + m] = c
    for s in SCANS:
        pts, off = R.scanset(s)
        o_pts, o_off = r.scanset(s)
        assert (off == o_off).all(), s
        _same(pts, o_pts, s)
        d["scan_" + s] = pts
        d["off_" + s] = off
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **d)
    print(name, {k: v.shape for k, v in d.items() if k.startswith("map_")})
    R.close()


def primitive_case():
    rng = np.random.default_rng(20250224)
    pts = rng.normal(0, 20, (4000, 4)).astype(np.float32)
    pts[:, 2] *= 0.1
    T = np.eye(4); T[:3, 3] = [3.5, -2.25, 0.5]
    c, s = np.cos(0.3), np.sin(0.3)
    T[:2, :2] = [[c, -s], [s, c]]
    Tinv = np.linalg.inv(T)
    # utility.cpp:64-72 + :92-142, :204-219 and the k-NN rule of Session.cpp:452-484 (k = 2, thr = 1.0 there), all reference-compiled
    rimg, idx = ref.map2range_img(ref.transform_global_map_to_local(pts, Tinv, np.eye(4)), 50.0, 360.0, 125, 900)
    o_rimg, o_idx = orc.range_image(pts, 50.0, 360.0, 125, 900, Tinv, None)
    assert (rimg.view(np.uint32) == o_rimg.view(np.uint32)).all() and (idx == o_idx).all()
    vox = ref.octree_downsampling(pts, 0.5)
    _same(vox, orc.voxel_centroid(pts, 0.5), "voxel")
    tgt = rng.normal(0, 2, (3000, 4)).astype(np.float32)
    q12 = (pts[:1000] * 0.5).astype(np.float32)
    R = ref.Removerter()
    near_k2_thr1 = R.weak_strong_split(tgt, q12)
    assert (near_k2_thr1 == orc.knn_split(tgt, q12, 2, 1.0)).all() and 0 < near_k2_thr1.sum() < len(near_k2_thr1)
    R.close()
    near = orc.knn_split(tgt, pts[:1000] * 0.1, 2, 0.05)       # other (k, thr): the oracle's rule (the reference text hard-codes 2 / 1.0 in the callable form)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "primitives.npz"), pts=pts, Tinv=Tinv, rimg=rimg, idx=idx, vox=vox, tgt=tgt, near=near,
                        near_k2_thr1=near_k2_thr1, source=np.array("oracle/_ref/libltm_ref.so (reference-compiled) except `near` (oracle)"))
    print("primitives", rimg.shape, vox.shape, int(near.sum()), int(near_k2_thr1.sum()))


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    primitive_case()
    pipeline_case("tiny_pair_1res", 3, "tiny", False, 2, 0.01)
    pipeline_case("tiny_pair_3res", 3, "tiny", True, 3, 0.1)
