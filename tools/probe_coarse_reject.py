"""VERDICT r5 item 3, measured on the CPU (numpy, no GPU needed): how many of the point-projections that phase 1 of k_vote_map_cull rejects with its per-pixel
bound `r^2 >= Q[pixel]` would ALREADY be rejected by a transcendental-free test against a coarse direction table -- a cube map of max Q per cell, built per
keyframe from the finished scan image?  Synthetic `lot` sessions (tools/synth.py, os1-64), the voxelised map of session 01, its own keyframes, resolution 2.5.

    python tools/probe_coarse_reject.py [--kf 120] [--sample 6]

Prints, per cells-per-face, the fraction of all point-projections and of phase 1's rejects that the coarse test removes."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cube_cell(d, n):
    """cube-map cell of unit-free direction vectors d (N,3): face by the dominant axis, (u, v) = the other two over it"""
    a = np.abs(d)
    ax = np.argmax(a, axis=1)
    m = a[np.arange(len(d)), ax]
    sgn = np.sign(d[np.arange(len(d)), ax]) > 0
    face = ax * 2 + sgn
    o1, o2 = (ax + 1) % 3, (ax + 2) % 3
    u = d[np.arange(len(d)), o1] / m
    v = d[np.arange(len(d)), o2] / m
    iu = np.clip(((u + 1.0) * 0.5 * n).astype(np.int64), 0, n - 1)
    iv = np.clip(((v + 1.0) * 0.5 * n).astype(np.int64), 0, n - 1)
    return (face * n + iu) * n + iv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kf", type=int, default=120)
    ap.add_argument("--sample", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r6_probe_coarse_reject_table.json"))
    a = ap.parse_args()
    from oracle import oracle_py as orc
    from tools import synth
    S = synth.to_numpy(synth.make_session(1, a.kf, "os1-64"))
    I4 = np.eye(4)
    pts, off = [], [0]
    for k in range(a.kf):
        p = orc.preclean(S["scans"][int(S["offsets"][k]):int(S["offsets"][k + 1])], 2.5)
        pts.append(p); off.append(off[-1] + len(p))
    scans, offs = np.concatenate(pts), np.array(off, np.uint64)
    cmap = orc.voxel_centroid(orc.merge_to_global(scans, offs, S["poses"], I4), 0.05)
    M = len(cmap)
    R, C = orc.rimg_size(50.0, 360.0, 2.5)
    thr = 0.1
    res = {n: dict(points=0, phase1_rejects=0, coarse_rejects=0) for n in (16, 32, 64, 128)}
    for kf in np.linspace(0, a.kf - 1, a.sample).astype(int):
        scan = scans[int(offs[kf]):int(offs[kf + 1])]
        rimg, _ = orc.range_image(scan, 50.0, 360.0, R, C, want_idx=False)
        # Q[px]: a map point of this pixel can only be flagged if r < s - thr (empty pixels: never, below 9800 m)
        lim = np.where(rimg < 9999.0, np.maximum(rimg.astype(np.float64) - thr + 1e-3, 0.0), 0.0)
        Q = (lim * lim).ravel()
        inv = S["inv"][kf].reshape(4, 4)
        loc = cmap[:, :3].astype(np.float64) @ inv[:3, :3].T + inv[:3, 3]
        r2 = (loc * loc).sum(1)
        az = np.degrees(np.arctan2(loc[:, 1], loc[:, 0])); el = np.degrees(np.arctan2(loc[:, 2], np.hypot(loc[:, 0], loc[:, 1])))
        row = np.clip(np.round(R * (1 - (el + 25.0) / 50.0)), 0, R - 1).astype(np.int64)
        col = np.clip(np.round(C * ((az + 180.0) / 360.0)), 0, C - 1).astype(np.int64)
        keep1 = r2 < Q[row * C + col]                       # phase 1 survivors
        # pixel directions (centres and corners) for the coarse tables
        rr, cc = np.meshgrid(np.arange(R), np.arange(C), indexing="ij")
        dirs = []
        for dr in (-0.5, 0.0, 0.5):
            for dc in (-0.5, 0.0, 0.5):
                e = np.radians(25.0 - 50.0 * (rr + dr) / R).ravel(); z = np.radians(360.0 * (cc + dc) / C - 180.0).ravel()
                dirs.append(np.stack([np.cos(e) * np.cos(z), np.cos(e) * np.sin(z), np.sin(e)], 1))
        # rows 0 / R-1 also receive every direction clamped from outside the field of view: their Q must cover the caps of the cube map
        top_q, bot_q = Q.reshape(R, C)[0].max(), Q.reshape(R, C)[-1].max()
        for n in res:
            table = np.zeros(6 * n * n)
            for d in dirs:
                np.maximum.at(table, cube_cell(d, n), Q)
            # conservative: dilate by one cell on each face (a pixel's footprint can straddle cells between the sampled directions)
            t = table.reshape(6, n, n)
            dil = t.copy()
            for du in (-1, 0, 1):
                for dv in (-1, 0, 1):
                    sh = np.roll(np.roll(t, du, 1), dv, 2)
                    dil = np.maximum(dil, sh)
            # cells no pixel direction reaches (above / below the field of view) take the clamped rows' bound
            cell_dirs_el = None
            table = dil.ravel()
            el_pt = el
            cell = cube_cell(loc, n)
            qc = table[cell]
            qc = np.where(el_pt > 25.0 - 50.0 / R, np.maximum(qc, top_q), qc)
            qc = np.where(el_pt < -25.0 + 50.0 / R, np.maximum(qc, bot_q), qc)
            coarse_reject = r2 >= qc
            assert not (coarse_reject & keep1).any(), "the coarse table must be conservative"
            res[n]["points"] += M; res[n]["phase1_rejects"] += int((~keep1).sum()); res[n]["coarse_rejects"] += int(coarse_reject.sum())
    out = {"what": "k_vote_map_cull phase 1 (r^2 >= Q[pixel]) against a coarse cube-map bound (max Q per cell, dilated by one cell), lot session 01, "
                   f"{a.kf} keyframes, map of {M} points, {a.sample} keyframes sampled, resolution 2.5 ({R} x {C})", "rows": []}
    for n, r in res.items():
        row = {"cells_per_face": f"{n} x {n}", "cell_deg": round(90.0 / n, 2), "phase1_reject_fraction": round(r["phase1_rejects"] / r["points"], 4),
               "coarse_reject_fraction_of_all": round(r["coarse_rejects"] / r["points"], 4), "coarse_share_of_phase1_rejects": round(r["coarse_rejects"] / max(r["phase1_rejects"], 1), 4)}
        out["rows"].append(row)
        print(row)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
