#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small seeded inputs + the CPU oracle's outputs for them.

The reference ships no golden vectors and cannot be built or imported here (C++/ROS/PCL), so these fixtures are produced
by the oracle (oracle/ltm_oracle.cpp), whose pins are: exhaustive atan2f vs glibc (oracle/pin_atan2f.c) and the hand-derived
known-answer tests (tests/test_oracle_kat.py).  They freeze the oracle's behaviour (tests/test_golden.py re-derives them on
CPU) and let the GPU path be checked against committed numbers.  Re-run after any intentional oracle change:
    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as orc  # noqa: E402
from tools import synth  # noqa: E402

MAPS = ["OriginalNoisyCentralMapGlobal", "OriginalNoisyQueryMapGlobal", "central_map_static", "central_map_dynamic", "query_map_static",
        "query_map_dynamic", "central_sess_high_dyn", "query_sess_high_dyn", "union_map_queryside", "union_map_centralside", "pd_map",
        "nd_map", "strong_nd_map", "weak_nd_map", "strong_pd_map", "weak_pd_map", "updated_map", "updated_map_strong"]
SCANS = ["scans_updated", "scans_updated_strong", "scans_pd", "scans_pd_strong", "scans_nd_strong"]


def pipeline_case(name, n_kf, sensor, three_res, k, thr):
    C = synth.to_numpy(synth.make_session(1, n_kf, sensor))
    Q = synth.to_numpy(synth.make_session(2, n_kf, sensor))
    res = (2.5, 2.0, 1.5) if three_res else (2.5,)
    r = orc.pipeline_run(orc.make_params(k=k, knn_thr=thr, use_self_removert=three_res, res_list=res), C, Q)
    d = {"meta": np.array([n_kf, int(three_res), k], dtype=np.int64), "thr": np.float32(thr)}
    for tag, S in (("c", C), ("q", Q)):
        d[f"{tag}_scans"], d[f"{tag}_off"], d[f"{tag}_poses"], d[f"{tag}_inv"] = S["scans"], S["offsets"], S["poses"], S["inv"]
    for m in MAPS:
        c = r.cloud(m)
        if c is not None:
            d["map_" + m] = c
    for s in SCANS:
        pts, off = r.scanset(s)
        d["scan_" + s] = pts
        d["off_" + s] = off
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **d)
    print(name, {k: v.shape for k, v in d.items() if k.startswith("map_")})


def primitive_case():
    rng = np.random.default_rng(20250224)
    pts = rng.normal(0, 20, (4000, 4)).astype(np.float32)
    pts[:, 2] *= 0.1
    T = np.eye(4); T[:3, 3] = [3.5, -2.25, 0.5]
    c, s = np.cos(0.3), np.sin(0.3)
    T[:2, :2] = [[c, -s], [s, c]]
    Tinv = np.linalg.inv(T)
    rimg, idx = orc.range_image(pts, 50.0, 360.0, 125, 900, Tinv, None)
    vox = orc.voxel_centroid(pts, 0.5)
    tgt = rng.normal(0, 2, (3000, 4)).astype(np.float32)
    near = orc.knn_split(tgt, pts[:1000] * 0.1, 2, 0.05)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "primitives.npz"), pts=pts, Tinv=Tinv, rimg=rimg, idx=idx, vox=vox, tgt=tgt, near=near)
    print("primitives", rimg.shape, vox.shape, int(near.sum()))


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    primitive_case()
    pipeline_case("tiny_pair_1res", 3, "tiny", False, 2, 0.01)
    pipeline_case("tiny_pair_3res", 3, "tiny", True, 3, 0.1)
