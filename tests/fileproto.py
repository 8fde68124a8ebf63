"""TEST INFRASTRUCTURE: the reference's file protocol for the drop-in tests -- session directories of PCD scans + pose
text files + params_ltmapper.yaml in, the 23-item output tree out -- and the host-side Step 0 (Session.cpp:80-302, 506-533) restated
with the oracle so that oracle and `ltm_run` start from the same loaded data."""
import os

import numpy as np

HDR = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
       "WIDTH {w}\nHEIGHT {h}\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")

MAP_FILES = ["OriginalNoisyCentralMapGlobal", "OriginalNoisyQueryMapGlobal", "central_sess_high_dyn", "query_sess_high_dyn",
             "union_map_queryside", "union_map_centralside", "pd_map", "nd_map", "strong_nd_map", "weak_nd_map", "strong_pd_map", "weak_pd_map",
             "updated_map", "updated_map_strong"]
SCAN_DIRS = [("scans_updated", True), ("scans_updated_strong", False), ("scans_pd", False), ("scans_pd_strong", False), ("scans_nd_strong", False)]


def lzf_compress(data):
    """greedy LZF compressor (liblzf's stream format, what pcl::io::savePCDFileBinaryCompressed writes): literal runs and back references of 3..264
    bytes at distances 1..8192.  Slow, for small test clouds only."""
    d = bytes(data)
    n = len(d)
    out, lit, last, i = bytearray(), bytearray(), {}, 0

    def flush():
        for a in range(0, len(lit), 32):
            piece = lit[a:a + 32]
            out.append(len(piece) - 1)
            out.extend(piece)
        lit.clear()
    while i < n:
        best = dist = 0
        if i + 3 <= n:
            key = d[i:i + 3]
            c = last.get(key, -1)
            last[key] = i
            if c >= 0 and i - c <= 8192:
                ln = 0
                while i + ln < n and ln < 264 and d[c + ln] == d[i + ln]:
                    ln += 1
                if ln >= 3:
                    best, dist = ln, i - c
        if not best:
            lit.append(d[i])
            i += 1
            continue
        flush()
        off, L = dist - 1, best - 2
        if L < 7:
            out.append((L << 5) | (off >> 8))
        else:
            out.append((7 << 5) | (off >> 8))
            out.append(L - 7)
        out.append(off & 0xff)
        i += best
    flush()
    return bytes(out)


def write_pcd(path, pts, ascii_=False, compressed=False):
    n = len(pts)
    with open(path, "wb") as f:
        if compressed:      # DATA binary_compressed: u32 compressed size, u32 raw size, LZF stream of the structure-of-arrays payload (all x, all y, ...)
            soa = np.ascontiguousarray(np.asarray(pts, dtype=np.float32).T).tobytes()
            comp = lzf_compress(soa)
            f.write(HDR.format(w=n, h=1, n=n).replace("DATA binary", "DATA binary_compressed").encode())
            f.write(np.array([len(comp), len(soa)], dtype=np.uint32).tobytes())
            f.write(comp)
        elif ascii_:
            f.write(HDR.format(w=n, h=1, n=n).replace("DATA binary", "DATA ascii").encode())
            for p in pts:
                f.write((" ".join(repr(float(v)) for v in p) + "\n").encode())
        else:
            f.write(HDR.format(w=n, h=1, n=n).encode())
            f.write(np.ascontiguousarray(pts, dtype=np.float32).tobytes())


def read_pcd(path):
    raw = open(path, "rb").read()
    i = raw.index(b"DATA binary\n") + len(b"DATA binary\n")
    hdr = raw[:i].decode()
    n = int([l for l in hdr.splitlines() if l.startswith("POINTS")][0].split()[1])
    return hdr, np.frombuffer(raw[i:], dtype=np.float32).reshape(n, 4)


def parse_keyframes(n, start, end):       # Session.cpp:138-173 incl. the double increment (quirk Q6), gap 1
    out, i = [], 0
    while i < n:
        if i > end or i < start:
            i += 2
            continue
        out.append(i)
        i += 1
    return out


def write_session_dirs(root, sessions, tags=("01", "02"), ascii_scans=(), compressed=False):
    """sessions: tools.synth.to_numpy dicts.  Returns the scan directories."""
    dirs = []
    for tag, S in zip(tags, sessions):
        d = os.path.join(str(root), tag, "Scans")
        os.makedirs(d)
        n_kf = len(S["offsets"]) - 1
        for k in range(n_kf):
            a, b = int(S["offsets"][k]), int(S["offsets"][k + 1])
            write_pcd(os.path.join(d, S["names"][k]), S["scans"][a:b], ascii_=(k in ascii_scans), compressed=compressed)
        with open(os.path.join(str(root), tag, "poses.txt"), "w") as f:
            for k in range(n_kf):
                f.write(" ".join(repr(float(v)) for v in S["poses"][k][:12]) + "\n")
        dirs.append(d)
    return dirs


def yaml_text(root, dirs, outdir, start_idx, end_idx, res_list=(2.5,), voxel=0.05, k=2, thr=0.01, extra=""):
    return f"""removert:
  isScanFileKITTIFormat: false
  saveMapPCD: true   # also writes OriginalNoisy*MapGlobal.pcd
  save_pcd_directory: "{outdir}"   # no trailing slash on purpose
  central_sess_scan_dir: "{dirs[0]}/"
  central_sess_pose_path: "{root}/01/poses.txt"
  query_sess_scan_dir: "{dirs[1]}/"
  query_sess_pose_path: "{root}/02/poses.txt"
  sequence_vfov: 50
  sequence_hfov: 360
  ExtrinsicLiDARtoPoseBase: [1.0, 0.0, 0.0, 0.0,
                             0.0, 1.0, 0.0, 0.0,
                             0.0, 0.0, 1.0, 0.0,
                             0.0, 0.0, 0.0, 1.0]
  use_keyframe_gap: true
  keyframe_gap: 1
  start_idx: {start_idx}
  end_idx: {end_idx}
  remove_resolution_list: [{", ".join(str(r) for r in res_list)}]
  downsample_voxel_size: {voxel}
  num_nn_points_within: {k}
  dist_nn_points_within: {thr}
  num_omp_cores: 16
  rimg_color_max: 20.0
{extra}"""


def host_load(orc, S, kfs, voxel=0.05, roundtrip_ascii=()):
    """Session::loadKeyframes + precleaningKeyframes on the selected keyframes: per-scan pcl::VoxelGrid (oracle restatement, A.6),
    near-range pre-clean; inverse poses by the oracle's restatement of Eigen's Matrix4d::inverse() (the host uses the library's: same bits)"""
    pts, off = [], [0]
    for k in kfs:
        a, b = int(S["offsets"][k]), int(S["offsets"][k + 1])
        raw = S["scans"][a:b]
        if k in roundtrip_ascii:   # went through the ascii writer: repr() round-trips float32 exactly
            raw = np.array([[np.float32(float(repr(float(v)))) for v in p] for p in raw], np.float32)
        p = orc.preclean(orc.voxel_grid(raw, voxel), 2.5)
        pts.append(p); off.append(off[-1] + len(p))
    poses = S["poses"].reshape(-1, 16)[kfs].copy()
    inv = orc.inverse_poses(poses)      # Session.cpp:109-110: the oracle's own Eigen restatement; the host uses the library's
    return dict(scans=np.concatenate(pts), offsets=np.array(off, np.uint64), poses=poses, inv=inv)


def query_keyframes_in_roi(central, c_kf, query, n_q):
    """Session::parseKeyframesInROI (Session.cpp:230-263): query keyframes within 10 m of any selected central pose"""
    c_pos = central["poses"].reshape(-1, 4, 4)[c_kf][:, :3, 3]
    q_pos = query["poses"].reshape(-1, 4, 4)[:, :3, 3]
    return [k for k in range(n_q) if np.sqrt(((c_pos - q_pos[k]) ** 2).sum(1)).min() <= 10.0]


def compare_output_tree(outdir, ref, central_names, assert_clouds_equal, xyz_tol=0.0):
    """every map file and the five per-keyframe directories of `outdir` against an oracle PipelineResult"""
    for fname in MAP_FILES:
        want = ref.cloud(fname)
        path = os.path.join(str(outdir), fname + ".pcd")
        if want is None:
            assert not os.path.exists(path), f"{fname}: written although the oracle has none"
            continue
        hdr, got = read_pcd(path)
        assert hdr == HDR.format(w=1, h=len(got), n=len(got)), f"{fname}: header is not what pcl::io::savePCDFileBinary writes"
        assert_clouds_equal(got, want, fname, xyz_tol=xyz_tol)
    for d, octree in SCAN_DIRS:
        w_pts, w_off = ref.scanset(d)
        names = sorted(os.listdir(os.path.join(str(outdir), d)))
        assert names == list(central_names), f"{d}: one file per central keyframe, named like the input scan"
        for j, nm in enumerate(names):
            hdr, got = read_pcd(os.path.join(str(outdir), d, nm))
            n = len(got)
            assert hdr == (HDR.format(w=1, h=n, n=n) if octree else HDR.format(w=n, h=1, n=n))
            assert_clouds_equal(got, w_pts[int(w_off[j]):int(w_off[j + 1])], f"{d}/{nm}", xyz_tol=xyz_tol)
