#!/usr/bin/env python3
"""Where does the bounded-error projection start to miss the exact pixel?  (margin of Geom::cull_eps_px)

    python tools/eps_sweep.py [--n 2000000]

For every image resolution alpha and every LTM_CULL_EPS_SCALE (band = scale * pixels-per-degree, the floor switched off) it throws
points that sit on pixel-rounding boundaries (+- a noise comparable with the band) at ltm_debug_cull_check, which
recomputes the reference's arithmetic per point and counts the points whose exact pixel is outside the candidate set.
The shipped scale must sit well above the first scale that shows a violation.  Prints one JSON line."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def boundary_points(rng, n, alpha, vfov, hfov, sigma_rad, rmin, rmax):
    import numpy as np
    rows, cols = int(round(vfov * alpha)), int(round(hfov * alpha))
    kc = rng.integers(0, cols, n)
    kr = rng.integers(0, rows, n)
    # reference pixel = round(cols * (deg(az) + hfov/2) / hfov): boundaries at (k + 0.5) * hfov / cols - hfov / 2
    az = np.deg2rad((kc + 0.5) * hfov / cols - hfov / 2.0) + rng.normal(0, sigma_rad, n)
    el = np.deg2rad(vfov / 2.0 - (kr + 0.5) * vfov / rows) + rng.normal(0, sigma_rad, n)
    which = rng.integers(0, 3, n)                     # 0: both on a boundary, 1: only azimuth, 2: only elevation
    az = np.where(which == 2, rng.uniform(-np.pi, np.pi, n), az)
    el = np.where(which == 1, np.deg2rad(rng.uniform(-vfov / 2, vfov / 2, n)), el)
    r = np.exp(rng.uniform(np.log(rmin), np.log(rmax), n))
    return np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--scales", default="6e-4,3e-4,2e-4,1.5e-4,1e-4,7e-5,5e-5,3e-5,1e-5")
    ap.add_argument("--alphas", default="1.5,2.5,4.0,6.0")
    args = ap.parse_args()
    import numpy as np
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd import capi
    vfov, hfov = 50.0, 360.0
    out = {}
    prng = np.random.default_rng(5)
    q, _ = np.linalg.qr(prng.normal(size=(3, 3)))
    T = np.eye(4); T[:3, :3] = q * np.sign(np.linalg.det(q)); T[:3, 3] = [310.0, -120.0, 4.0]
    Tinv = np.linalg.inv(T)
    for scale in args.scales.split(","):
        os.environ["LTM_CULL_EPS_SCALE"] = scale
        os.environ["LTM_CULL_EPS_FLOOR"] = "0"
        ctx = capi.Context(vfov=vfov, hfov=hfov, device=0)
        row = {}
        for alpha in (float(a) for a in args.alphas.split(",")):
            rng = np.random.default_rng(int(alpha * 1000))
            bad = 0
            for sigma in (1e-6, 4e-6, 1.5e-5):
                for (rmin, rmax) in ((0.3, 3.0), (3.0, 150.0), (150.0, 9000.0)):
                    pts = boundary_points(rng, args.n, alpha, vfov, hfov, sigma, rmin, rmax)
                    bad += ctx.cull_check(pts, alpha)
                    if rmax <= 150.0:
                        glob = ((T[:3, :3] @ pts.astype(np.float64).T).T + T[:3, 3]).astype(np.float32)
                        bad += ctx.cull_check(glob, alpha, Tinv)
            row[str(alpha)] = bad
        out[scale] = row
        ctx.close()
        print(scale, row, file=sys.stderr, flush=True)
    print(json.dumps({"what": "cull_check violations per LTM_CULL_EPS_SCALE and alpha", "n_per_set": args.n, "violations": out}))


if __name__ == "__main__":
    main()
