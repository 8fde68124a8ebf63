"""synth-v1: synthetic LiDAR sessions in the reference's in-memory shape.

No dataset is reachable offline (README.md:104 points to a download), so every config of BASELINE.json
is restated as a seeded synthetic scene.  This generator is device agnostic torch code: on the GPU box it
produces the 2x500-keyframe benchmark sessions in about a second, on CPU it produces the small parity
fixtures.  It is tooling, not product: nothing in lt-mapper_amd/ imports it.

Scene `lot` (ParkingLot-like): ground z=0, perimeter walls of a 120 x 80 m lot (10 m high), 6 rows x 40
parking bays; a bay is occupied in session s with probability 0.6 and a quarter of the bays are re-drawn
between consecutive sessions (low-dynamic change => ND/PD); 10 movers (pedestrians, cars) whose position is
a function of the keyframe index (high-dynamic points).  The sensor drives a rounded-rectangle loop at 1 m
keyframe spacing, 1.9 m above ground; session s starts 37*s m further along the loop.
Sensors: `os1-64` 64 x 1024, `hdl-64e` 64 x 1900, `mls` 128 x 8192, or any (rings, az, el_lo, el_hi).
Analytic ray casting (slab tests), max range 120 m, range noise N(0, 0.02 m), 5 % dropout, intensity U[0,255).
Poses are reported in the shared frame with a small SE(2) residual and 6 significant digits, like the text
files LT-SLAM writes (ltslam/src/utility.cpp:190-200).
"""
import math

import numpy as np
import torch

SENSORS = {
    "os1-64": (64, 1024, -22.5, 22.5),
    "hdl-64e": (64, 1900, -24.9, 2.0),
    "mls": (128, 8192, -25.0, 25.0),
    "tiny": (16, 180, -22.5, 22.5),
    "small": (32, 512, -22.5, 22.5),
}

MASTER_SEED = 20250224


def _hash01(*keys):
    """deterministic uniform [0,1) from integer keys (splitmix64)"""
    z = 0x9E3779B97F4A7C15
    for k in keys:
        z = (z ^ (int(k) & 0xFFFFFFFFFFFFFFFF)) * 0xBF58476D1CE4E5B9 & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 30
        z = z * 0x94D049BB133111EB & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
    return (z >> 11) / float(1 << 53)


def _bay_boxes(session, seed):
    """AABBs [x0,y0,z0,x1,y1,z1] of the cars parked in `session`"""
    rows_y = [-30.0, -18.0, -6.0, 6.0, 18.0, 30.0]
    boxes = []
    for r, y in enumerate(rows_y):
        for b in range(40):
            bay = r * 40 + b
            epoch = 0
            for s in range(1, session + 1):
                if _hash01(seed, 7, bay, s) < 0.25:
                    epoch = s
            if _hash01(seed, 11, bay, epoch) < 0.6:
                cx = -48.75 + 2.5 * b
                boxes.append([cx - 0.9, y - 2.25, 0.0, cx + 0.9, y + 2.25, 1.5])
    return boxes


def _mover_boxes(kf_global, seed):
    """10 movers; kf_global = arc position in metres (1 keyframe per metre, ~1 s per keyframe)"""
    t = float(kf_global)
    boxes = []
    for m in range(10):
        lane_y = [-24.0, -12.0, 0.0, 12.0, 24.0][m % 5] + (1.5 if m < 5 else -1.5)
        x0 = -50.0 + 100.0 * _hash01(seed, 13, m)
        if m < 6:   # pedestrian 0.6 x 0.6 x 1.7 at 1.4 m/s
            v, hx, hy, hz = 1.4 * (1 if m % 2 == 0 else -1), 0.3, 0.3, 1.7
        else:       # car 4.5 x 1.8 x 1.5 at 4 m/s
            v, hx, hy, hz = 4.0 * (1 if m % 2 == 0 else -1), 2.25, 0.9, 1.5
        x = ((x0 + v * t + 50.0) % 100.0) - 50.0
        boxes.append([x - hx, lane_y - hy, 0.0, x + hx, lane_y + hy, hz])
    return boxes


def _loop_pose(s):
    """rounded-rectangle loop, half extents (45, 12), corner radius 4; s = arc length"""
    hx, hy, r = 45.0, 12.0, 4.0
    sx, sy = 2 * (hx - r), 2 * (hy - r)
    arc = 0.5 * math.pi * r
    per = 2 * sx + 2 * sy + 4 * arc
    s = s % per
    segs = [
        ("l", sx, (-(hx - r), -hy), 0.0), ("a", arc, (hx - r, -hy + r), -0.5 * math.pi),
        ("l", sy, (hx, -(hy - r)), 0.5 * math.pi), ("a", arc, (hx - r, hy - r), 0.0),
        ("l", sx, (hx - r, hy), math.pi), ("a", arc, (-(hx - r), hy - r), 0.5 * math.pi),
        ("l", sy, (-hx, hy - r), -0.5 * math.pi), ("a", arc, (-(hx - r), -hy + r), math.pi),
    ]
    for kind, length, p, ang in segs:
        if s <= length:
            if kind == "l":
                return p[0] + s * math.cos(ang), p[1] + s * math.sin(ang), ang
            a = ang + s / r
            return p[0] + r * math.cos(a), p[1] + r * math.sin(a), a + 0.5 * math.pi
        s -= length
    return -(hx - r), -hy, 0.0


# ---------------------------------------------------------------------------------------- scene `street`
STREET_N = 6            # blocks per side
STREET_PITCH = 120.0    # 100 m block + 20 m road


def _street_static_boxes(session, seed):
    """buildings (one per block, 80 x 80 m footprint, 8-30 m high) + cars parked along the kerbs of the east-west roads"""
    boxes = []
    for i in range(STREET_N):
        for j in range(STREET_N):
            cx, cy = 60.0 + STREET_PITCH * i + 10.0, 60.0 + STREET_PITCH * j + 10.0
            h = 8.0 + 22.0 * _hash01(seed, 31, i, j)
            boxes.append([cx - 40.0, cy - 40.0, 0.0, cx + 40.0, cy + 40.0, h])
    for j in range(STREET_N + 1):                       # east-west roads centred on y = j*120
        for side in (-1, 1):
            for b in range(int(STREET_N * STREET_PITCH / 6.0)):
                bay = (j * 2 + (side > 0)) * 1000 + b
                epoch = 0
                for s in range(1, session + 1):
                    if _hash01(seed, 37, bay, s) < 0.25:
                        epoch = s
                if _hash01(seed, 41, bay, epoch) < 0.35:
                    x0, y0 = 3.0 + 6.0 * b, STREET_PITCH * j + side * 7.5
                    boxes.append([x0 - 2.25, y0 - 0.9, 0.0, x0 + 2.25, y0 + 0.9, 1.5])
    return boxes


def _street_movers(t, seed):
    boxes = []
    for m in range(30):
        j = m % (STREET_N + 1)
        y = STREET_PITCH * j + (2.0 if m % 2 else -2.0)
        span = STREET_N * STREET_PITCH
        x0 = span * _hash01(seed, 43, m)
        if m < 18:
            v, hx, hy, hz = 1.4 * (1 if m % 2 == 0 else -1), 0.3, 0.3, 1.7
            y += 5.0 if m % 4 < 2 else -5.0
        else:
            v, hx, hy, hz = 8.0 * (1 if m % 2 == 0 else -1), 2.25, 0.9, 1.5
        x = (x0 + v * t) % span
        boxes.append([x - hx, y - hy, 0.0, x + hx, y + hy, hz])
    return boxes


def _street_pose(s):
    """lawn-mower path along the east-west roads, connected on the outer north-south roads"""
    span = STREET_N * STREET_PITCH
    leg = span + STREET_PITCH
    total = leg * (STREET_N + 1)
    s = s % total
    j = int(s // leg)
    u = s - j * leg
    east = (j % 2 == 0)
    if u <= span:
        x = u if east else span - u
        return x, STREET_PITCH * j, 0.0 if east else math.pi
    x = span if east else 0.0
    return x, STREET_PITCH * j + (u - span), 0.5 * math.pi


def _round_sig(x, sig=6):
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    nz = x != 0
    mag = np.floor(np.log10(np.abs(x[nz])))
    scale = 10.0 ** (sig - 1 - mag)
    out[nz] = np.round(x[nz] * scale) / scale
    return out


def _cast(o, d, boxes, chunk=131072):
    """o (3,), d (R,3) float64, boxes (B,6): nearest positive slab hit per ray -> (R,) range (inf if none)"""
    out = []
    for a in range(0, d.shape[0], chunk):
        inv = 1.0 / d[a:a + chunk]                  # (R,3); zeros give inf, handled by min/max
        t0 = (boxes[None, :, 0:3] - o[None, None, :]) * inv[:, None, :]
        t1 = (boxes[None, :, 3:6] - o[None, None, :]) * inv[:, None, :]
        tn = torch.minimum(t0, t1).amax(dim=2)
        tf = torch.maximum(t0, t1).amin(dim=2)
        hit = (tn <= tf) & (tn > 0.5)
        tn = torch.where(hit, tn, torch.full_like(tn, float("inf")))
        out.append(tn.amin(dim=1))
    return torch.cat(out) if len(out) > 1 else out[0]


def make_session(session, n_kf, sensor="os1-64", seed=MASTER_SEED, device="cpu", scene="lot", kf_spacing=1.0,
                 max_range=120.0, noise=0.02, dropout=0.05, pose_noise=True, tilt_deg=0.0, z_drift=0.0, origin=(0.0, 0.0, 0.0)):
    """returns dict(scans (P,4) f32, offsets (n_kf+1) u64, poses (n_kf,16) f64, inv (n_kf,16) f64, names)
    with scans/offsets as torch tensors on `device` and poses as numpy.

    tilt_deg / z_drift / origin (round 4, full SE(3) poses as LT-SLAM output has them): every keyframe's sensor is rolled and pitched by
    N(0, tilt_deg) degrees (hash-seeded) and lifted by N(0, z_drift) m -- the rays are cast from that attitude, so the scans stay consistent
    with the scene -- and the whole session is reported `origin` metres away from where it was generated (poses only; at 40 km the 6 significant
    digits of the pose text quantise translations to 0.1 m, exactly as the reference's writer would, ltslam/src/utility.cpp:190-200)."""
    assert scene in ("lot", "street")
    rings, n_az, el_lo, el_hi = SENSORS[sensor] if isinstance(sensor, str) else sensor
    dev = torch.device(device)
    f64 = torch.float64
    el = torch.deg2rad(torch.linspace(el_lo, el_hi, rings, dtype=f64, device=dev))
    az0 = torch.arange(n_az, dtype=f64, device=dev) * (2 * math.pi / n_az) - math.pi
    room = torch.tensor([[-60.0, -40.0, 0.0, 60.0, 40.0, 10.0]], dtype=f64, device=dev)
    static_boxes = _bay_boxes(session, seed) if scene == "lot" else _street_static_boxes(session, seed)
    static_t = torch.tensor(static_boxes, dtype=f64, device=dev)
    gen = torch.Generator(device=dev)
    scans, offsets, poses = [], [0], []
    for kf in range(n_kf):
        gen.manual_seed((seed * 1000003 + session * 65537 + kf) & 0x7FFFFFFFFFFF)
        s_arc = 37.0 * session + kf_spacing * kf
        x, y, yaw = _loop_pose(s_arc) if scene == "lot" else _street_pose(s_arc)
        if tilt_deg or z_drift:       # Box-Muller on the hash stream: deterministic per (seed, session, keyframe)
            def gauss(a, b):
                u1, u2 = max(_hash01(seed, a, session, kf), 1e-12), _hash01(seed, b, session, kf)
                return math.sqrt(-2.0 * math.log(u1)) * math.cos(2 * math.pi * u2)
            roll, pitch = math.radians(tilt_deg) * gauss(31, 37), math.radians(tilt_deg) * gauss(41, 43)
            dz = z_drift * gauss(47, 53)
        else:
            roll = pitch = dz = 0.0
        o = torch.tensor([x, y, 1.9 + dz], dtype=f64, device=dev)
        phase = float(_hash01(seed, 17, session, kf)) * (2 * math.pi / n_az)
        az = az0 + phase
        ce, se = torch.cos(el)[:, None], torch.sin(el)[:, None]
        dl = torch.stack([(ce * torch.cos(az)[None, :]), (ce * torch.sin(az)[None, :]), se.expand(rings, n_az)], dim=2).reshape(-1, 3)
        cy, sy = math.cos(yaw), math.sin(yaw)
        if roll or pitch:
            cp, sp, cr, sr = math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
            Rt = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]]) @ np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])      # attitude about the sensor
            R = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]]) @ Rt
            dw = dl @ torch.tensor(R.T, dtype=f64, device=dev)
        else:
            Rt = None
            dw = torch.stack([cy * dl[:, 0] - sy * dl[:, 1], sy * dl[:, 0] + cy * dl[:, 1], dl[:, 2]], dim=1)
        if scene == "lot":
            # room: exit distance of the enclosing box; a ceiling exit is sky (no return)
            inv = 1.0 / dw
            t0 = (room[0, 0:3][None, :] - o[None, :]) * inv
            t1 = (room[0, 3:6][None, :] - o[None, :]) * inv
            tmax = torch.maximum(t0, t1)
            t_exit, axis = tmax.min(dim=1)
            sky = (axis == 2) & (dw[:, 2] > 0)
            rng = torch.where(sky, torch.full_like(t_exit, float("inf")), t_exit)
            movers = _mover_boxes(s_arc, seed)
        else:
            # open scene: ground plane z = 0 below the horizon, sky above
            tg = -o[2] / dw[:, 2]
            rng = torch.where(dw[:, 2] < 0, tg, torch.full_like(tg, float("inf")))
            movers = _street_movers(s_arc, seed)
        boxes = torch.cat([static_t, torch.tensor(movers, dtype=f64, device=dev)], dim=0)
        if scene == "street":   # only boxes within reach of this pose can be hit: keeps the ray x box tensor small
            c = 0.5 * (boxes[:, 0:2] + boxes[:, 3:5]); h = 0.5 * (boxes[:, 3:5] - boxes[:, 0:2])
            near = (torch.clamp((c - o[None, 0:2]).abs() - h, min=0.0).norm(dim=1) < max_range)
            boxes = boxes[near]
        rng = torch.minimum(rng, _cast(o, dw, boxes))
        u = torch.rand(rng.shape[0], 3, generator=gen, device=dev, dtype=f64)
        g4 = torch.rand(rng.shape[0], 4, generator=gen, device=dev, dtype=f64).sum(dim=1)   # Irwin-Hall(4): var 1/3
        rng = rng + noise * (g4 - 2.0) * math.sqrt(3.0)
        keep = torch.isfinite(rng) & (rng < max_range) & (rng > 0.3) & (u[:, 0] >= dropout)
        pts = (dl * rng[:, None])[keep]
        inten = (u[:, 1] * 255.0)[keep]
        scans.append(torch.cat([pts, inten[:, None]], dim=1).to(torch.float32))
        offsets.append(offsets[-1] + int(scans[-1].shape[0]))
        # reported pose: truth + small SE(2) residual, 6 significant digits
        if pose_noise:
            ex = 0.02 * (2 * _hash01(seed, 19, session, kf) - 1)
            ey = 0.02 * (2 * _hash01(seed, 23, session, kf) - 1)
            eyaw = math.radians(0.05) * (2 * _hash01(seed, 29, session, kf) - 1)
        else:
            ex = ey = eyaw = 0.0
        c2, s2 = math.cos(yaw + eyaw), math.sin(yaw + eyaw)
        T = np.array([[c2, -s2, 0, x + ex], [s2, c2, 0, y + ey], [0, 0, 1, 1.9 + dz], [0, 0, 0, 1]], dtype=np.float64)
        if Rt is not None:
            T[:3, :3] = T[:3, :3] @ Rt
        T[:3, 3] += np.asarray(origin, dtype=np.float64)
        T[:3, :] = _round_sig(T[:3, :], 6)
        poses.append(T)
    poses = np.stack(poses) if poses else np.zeros((0, 4, 4))
    inv_p = np.linalg.inv(poses) if n_kf else poses.copy()
    scans_t = torch.cat(scans, dim=0) if scans else torch.zeros((0, 4), dtype=torch.float32, device=dev)
    return dict(scans=scans_t.contiguous(), offsets=torch.tensor(offsets, dtype=torch.int64), poses=poses.reshape(-1, 16),
                inv=inv_p.reshape(-1, 16), names=[f"{i:06d}.pcd" for i in range(n_kf)])


def to_numpy(sess):
    return dict(scans=sess["scans"].cpu().numpy(), offsets=sess["offsets"].cpu().numpy().astype(np.uint64),
                poses=sess["poses"], inv=sess["inv"], names=sess["names"])


if __name__ == "__main__":
    import time
    t = time.time()
    s = make_session(1, 4, "small")
    print(s["scans"].shape, s["offsets"], time.time() - t)
