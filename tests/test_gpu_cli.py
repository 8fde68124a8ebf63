"""File-protocol drop-in test: the ROS-free `ltm_run` (C++ mirror of Removerter/Session over the C ABI) reads the reference's
on-disk inputs (flat PCD scan directories, pose text files, params_ltmapper.yaml keys) and writes the reference's output
tree.  Every output file is compared with the CPU oracle run on the same loaded data."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_clouds_equal

pytestmark = pytest.mark.gpu

from fileproto import HDR, parse_keyframes, read_pcd, write_pcd   # noqa: E402


# the three ways the host gets its outputs to disk: background writer fed through the library's ring of pinned chunks (default),
# background writer on whole page-locked buffers (gpu_fetch_chunked: false), and the synchronous path (gpu_async_io: false)
WRITERS = {"chunked": "", "whole_buffer": "  gpu_fetch_chunked: false\n", "synchronous": "  gpu_async_io: false\n"}


@pytest.mark.parametrize("writer", list(WRITERS))
def test_ltm_run_file_protocol(tmp_path, orc, writer):
    from tools import synth
    exe = os.path.join(ROOT, "lt-mapper_amd", "host", "ltm_run")
    assert os.path.exists(exe), "build the host mirror first (make host)"
    n_kf = 40
    sess = [synth.to_numpy(synth.make_session(s, n_kf, "tiny")) for s in (1, 2)]
    dirs = []
    for tag, S in zip(("01", "02"), sess):
        d = tmp_path / tag / "Scans"
        d.mkdir(parents=True)
        for k in range(n_kf):
            a, b = int(S["offsets"][k]), int(S["offsets"][k + 1])
            write_pcd(str(d / S["names"][k]), S["scans"][a:b], ascii_=(k == 3))
        with open(tmp_path / tag / "poses.txt", "w") as f:
            for k in range(n_kf):
                f.write(" ".join(repr(float(v)) for v in S["poses"][k][:12]) + "\n")
        dirs.append(d)
    outdir = tmp_path / "out"
    start_idx, end_idx = 11, 39         # odd start: the first in-range scan is skipped in the reference (Q6)
    yaml = tmp_path / "params.yaml"
    yaml.write_text(f"""removert:
  isScanFileKITTIFormat: false
  saveMapPCD: true   # also writes OriginalNoisy*MapGlobal.pcd
  save_pcd_directory: "{outdir}"   # no trailing slash on purpose
  central_sess_scan_dir: "{dirs[0]}/"
  central_sess_pose_path: "{tmp_path}/01/poses.txt"
  query_sess_scan_dir: "{dirs[1]}/"
  query_sess_pose_path: "{tmp_path}/02/poses.txt"
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
  remove_resolution_list: [2.5]
  downsample_voxel_size: 0.05
  num_nn_points_within: 2
  dist_nn_points_within: 0.01
  num_omp_cores: 16
  rimg_color_max: 20.0
  gpu_viz_every: 7      # RViz images of every 7th source keyframe of each vote pass -> <out>/viz/*.ppm
""" + WRITERS[writer])
    r = subprocess.run([exe, str(yaml)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]

    # ---- the same Step 0 on the host side of the oracle
    c_kf = parse_keyframes(n_kf, start_idx, end_idx)
    assert c_kf[0] == 12, "quirk Q6: an odd start_idx skips the first in-range scan"
    c_pos = sess[0]["poses"].reshape(-1, 4, 4)[c_kf][:, :3, 3]
    q_pos = sess[1]["poses"].reshape(-1, 4, 4)[:, :3, 3]
    q_kf = [k for k in range(n_kf) if np.sqrt(((c_pos - q_pos[k]) ** 2).sum(1)).min() <= 10.0]
    assert len(q_kf) > 3

    def load(S, kfs):
        pts, off = [], [0]
        for k in kfs:
            a, b = int(S["offsets"][k]), int(S["offsets"][k + 1])
            raw = S["scans"][a:b]
            if k == 3:   # went through the ascii writer: repr() round-trips float32 exactly
                raw = np.array([[np.float32(float(repr(float(v)))) for v in p] for p in raw], np.float32)
            p = orc.preclean(orc.voxel_grid(raw, 0.05), 2.5)   # pcl::VoxelGrid of the loader: oracle restatement (A.6)
            pts.append(p); off.append(off[-1] + len(p))
        poses = S["poses"].reshape(-1, 16)[kfs].copy()
        # the pose file holds 12 numbers per line; the inverse is computed by the host (ltm_inverse4x4) and by the oracle, each with its own restatement of Eigen's kernel
        inv = orc.inverse_poses(poses)      # Session.cpp:109-110: the oracle's own Eigen restatement; the host uses the library's
        return dict(scans=np.concatenate(pts), offsets=np.array(off, np.uint64), poses=poses, inv=inv)

    C, Q = load(sess[0], c_kf), load(sess[1], q_kf)
    ref = orc.pipeline_run(orc.make_params(k=2, knn_thr=0.01), C, Q)

    def close(a, b, what):   # host (ltm_inverse4x4) and oracle (orc_inverse4x4) restate the same Eigen kernel: the file path is bitwise
        assert_clouds_equal(a, b, what, xyz_tol=0.0)

    files = {"OriginalNoisyCentralMapGlobal": "OriginalNoisyCentralMapGlobal", "OriginalNoisyQueryMapGlobal": "OriginalNoisyQueryMapGlobal",
             "central_sess_high_dyn": "central_sess_high_dyn", "query_sess_high_dyn": "query_sess_high_dyn",
             "union_map_queryside": "union_map_queryside", "union_map_centralside": "union_map_centralside", "pd_map": "pd_map", "nd_map": "nd_map",
             "strong_nd_map": "strong_nd_map", "weak_nd_map": "weak_nd_map", "strong_pd_map": "strong_pd_map", "weak_pd_map": "weak_pd_map",
             "updated_map": "updated_map", "updated_map_strong": "updated_map_strong"}
    for fname, key in files.items():
        want = ref.cloud(key)
        path = outdir / (fname + ".pcd")
        if want is None:
            assert not path.exists()
            continue
        hdr, got = read_pcd(str(path))
        assert hdr == HDR.format(w=1, h=len(got), n=len(got)), f"{fname}: header is not what pcl::io::savePCDFileBinary writes"
        close(got, want, fname)
    for d, key, octree in (("scans_updated", "scans_updated", True), ("scans_updated_strong", "scans_updated_strong", False), ("scans_pd", "scans_pd", False),
                           ("scans_pd_strong", "scans_pd_strong", False), ("scans_nd_strong", "scans_nd_strong", False)):
        w_pts, w_off = ref.scanset(key)
        names = sorted(os.listdir(outdir / d))
        assert names == [sess[0]["names"][k] for k in c_kf], f"{d}: one file per central keyframe, named like the input scan"
        for j, nm in enumerate(names):
            hdr, got = read_pcd(str(outdir / d / nm))
            n = len(got)
            assert hdr == (HDR.format(w=1, h=n, n=n) if octree else HDR.format(w=n, h=1, n=n))
            close(got, w_pts[int(w_off[j]):int(w_off[j + 1])], f"{d}/{nm}")
    assert (outdir / "map_static").is_dir() and (outdir / "map_dynamic").is_dir()

    # ---- the device-rendered RViz images of pass 0 (central self-removal at 2.5), keyframe 0
    viz = sorted(os.listdir(outdir / "viz"))
    n_src = len(c_kf)
    assert sum(1 for v in viz if v.startswith("000_")) == 4 * ((n_src + 6) // 7), viz[:8]
    rows, cols = orc.rimg_size(50.0, 360.0, 2.5)

    def read_ppm(path):
        raw = open(path, "rb").read()
        head = f"P6\n{cols} {rows}\n255\n".encode()
        assert raw.startswith(head) and len(raw) == len(head) + rows * cols * 3
        return np.frombuffer(raw[len(head):], np.uint8).reshape(rows, cols, 3)[:, :, ::-1]      # RGB file -> BGR

    _, cmap0 = read_pcd(str(outdir / "OriginalNoisyCentralMapGlobal.pcd"))
    map_r, map_i = orc.range_image(cmap0, 50.0, 360.0, rows, cols, T1=C["inv"][0], T2=np.eye(4))
    scan_r, _ = orc.range_image(C["scans"][int(C["offsets"][0]):int(C["offsets"][1])], 50.0, 360.0, rows, cols, want_idx=False)
    for name, want in (("scan", orc.colormap(scan_r, 0.0, 20.0)), ("map", orc.colormap(map_r, 0.0, 20.0)),
                       ("diff", orc.colormap(scan_r - map_r, 0.0, 0.5)), ("ptidx", orc.colormap(map_i, 0.0, float(len(cmap0))))):
        got = read_ppm(str(outdir / "viz" / f"000_000000_{name}.ppm"))
        bad = (got != want).any(axis=2).mean()
        assert bad == 0.0, f"viz {name}: {bad:.4%} of the pixels differ"

    # ---- the reference's fine-grained methods (calcDescrepancyAndParseDynamicPointIdxForEachScan, getStaticIdxFromDynamicIdx,
    # parsePointcloudSubsetUsingPtIdx, scan2RangeImg) kept as thin wrappers must reproduce the batch partition
    r = subprocess.run([exe, str(yaml), "--check-wrappers"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "check-wrappers: OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]

    # ---- lifelong hand-over through the file protocol.  Run 2: central := scans_updated/ of run 1 with the pose
    # subset ltm_run wrote next to it, query := session 02 again (the synthetic sessions 01 and 03 do not overlap within 40 keyframes).  The loader re-applies VoxelGrid + pre-clean, as the reference would.
    n_c = len(c_kf)
    out2 = tmp_path / "out2"
    yaml2 = tmp_path / "params2.yaml"
    yaml2.write_text(yaml.read_text().replace(f'save_pcd_directory: "{outdir}"', f'save_pcd_directory: "{out2}/"')
                     .replace(f'central_sess_scan_dir: "{dirs[0]}/"', f'central_sess_scan_dir: "{outdir}/scans_updated/"')
                     .replace(f'central_sess_pose_path: "{tmp_path}/01/poses.txt"', f'central_sess_pose_path: "{outdir}/scans_updated_poses.txt"')
                     .replace(f"start_idx: {start_idx}", "start_idx: 0").replace(f"end_idx: {end_idx}", f"end_idx: {n_c - 1}")
                     .replace("gpu_viz_every: 7", "gpu_viz_every: 0"))
    pose_lines = open(outdir / "scans_updated_poses.txt").read().split("\n")[:-1]
    assert len(pose_lines) == n_c
    got_poses = np.array([[float(v) for v in ln.split()] for ln in pose_lines])
    assert (got_poses == sess[0]["poses"].reshape(-1, 16)[c_kf][:, :12]).all(), "pose subset must round-trip exactly"
    r = subprocess.run([exe, str(yaml2)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    upd = [read_pcd(str(outdir / "scans_updated" / sess[0]["names"][k]))[1] for k in c_kf]
    pts2, off2 = [], [0]
    for p in upd:
        q = orc.preclean(orc.voxel_grid(p, 0.05), 2.5)
        pts2.append(q); off2.append(off2[-1] + len(q))
    poses2 = sess[0]["poses"].reshape(-1, 16)[c_kf].copy()
    C2 = dict(scans=np.concatenate(pts2), offsets=np.array(off2, np.uint64), poses=poses2,
              inv=orc.inverse_poses(poses2))
    c2_pos = poses2.reshape(-1, 4, 4)[:, :3, 3]
    q3_pos = sess[1]["poses"].reshape(-1, 4, 4)[:, :3, 3]
    q3_kf = [k for k in range(n_kf) if np.sqrt(((c2_pos - q3_pos[k]) ** 2).sum(1)).min() <= 10.0]
    Q3 = load(sess[1], q3_kf)
    ref2 = orc.pipeline_run(orc.make_params(k=2, knn_thr=0.01), C2, Q3)
    for fname in ("updated_map", "pd_map", "nd_map"):
        want = ref2.cloud(fname)
        if want is None:
            assert not (out2 / (fname + ".pcd")).exists()
            continue
        close(read_pcd(str(out2 / (fname + ".pcd")))[1], want, "cascade run 2 " + fname)
    w_pts, w_off = ref2.scanset("scans_updated")
    for j, k in enumerate(c_kf):
        close(read_pcd(str(out2 / "scans_updated" / sess[0]["names"][k]))[1], w_pts[int(w_off[j]):int(w_off[j + 1])], f"cascade run 2 scans_updated/{k}")


def test_cxx_host_and_python_host_drive_the_same_pipeline(tmp_path):
    """`Removerter::run()` exists twice, in lt-mapper_amd/host/src/Removerter.cpp (the north-star C++ host) and in
    lt-mapper_amd/removerter.py (tests, bench, torch.distributed sharding).  Both only sequence C-ABI calls: on the same loaded sessions they must
    launch every kernel class the same number of times over the same number of work units.  `ltm_run --bench` reports its per-class launch counts
    and units (ltm_profile_read); the Python host is profiled the same way on the same data."""
    import fileproto as fp
    from ltmapper_amd import capi
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    from tools import synth, t_total
    n_kf = 44      # the two sessions start 37 m apart on the same loop: the query's first keyframes fall into the central ROI from ~40 central keyframes on
    sess = [synth.to_numpy(synth.make_session(s, n_kf, "small")) for s in (1, 2)]
    dirs = fp.write_session_dirs(tmp_path, sess)
    cxx = t_total.bench_cxx_host(str(tmp_path), dirs, n_kf, three_res=True, steps=1, warmup=1, lanes=1)
    cxx2 = t_total.bench_cxx_host(str(tmp_path), dirs, n_kf, three_res=True, steps=1, warmup=1, lanes=2)      # the same stages on two lanes
    assert cxx["lanes"] == 1 and cxx2["lanes"] == 2
    # the Python host on what the C++ loader makes of those files: keyframes 0..n_kf-1 of the central session (even start: no Q6 skip), the query
    # keyframes inside the 10 m ROI; per-scan VoxelGrid + pre-clean on the device
    c_kf = fp.parse_keyframes(n_kf, 0, n_kf - 1)
    q_kf = fp.query_keyframes_in_roi(sess[0], c_kf, sess[1], n_kf)
    assert len(q_kf) >= 3, "the test needs an overlapping query session"
    assert cxx["keyframes"] == [len(c_kf), len(q_kf)]
    ctx = capi.Context(vfov=50.0, hfov=360.0, device=0)
    loaded = []
    for S, kfs in ((sess[0], c_kf), (sess[1], q_kf)):
        pts = [S["scans"][int(S["offsets"][k]):int(S["offsets"][k + 1])] for k in kfs]
        off = np.cumsum([0] + [len(p) for p in pts]).astype(np.uint64)
        scans = ctx.preclean(ctx.voxel_grid_scanset(ctx.upload_scans(np.concatenate(pts), off), 0.05), 2.5)
        loaded.append((scans, ctx.poses(S["poses"].reshape(-1, 16)[kfs])))
    P = Params(gpu_use_self_removert=True, remove_resolution_list=[2.5, 2.0, 1.5])
    ctx.profile_enable(True); ctx.profile_reset()
    rm = Removerter(HipOps(ctx), P, Session("Central", *loaded[0]), Session("Query", *loaded[1]))
    rm.run()
    ctx.synchronize()
    py = ctx.profile_read()
    ctx.close()
    names = sorted(set(cxx["classes"]) | set(py))
    diff = []
    for nme in names:
        a, b = cxx["classes"].get(nme), py.get(nme)
        la, ua = (a["launches_per_step"], a["units_per_step"]) if a else (0, 0)
        lb, ub = (b["launches"], b["units"]) if b else (0, 0)
        if nme == "knn_query_p2":
            # its units are the queries phase 1 left undecided: which nine points of a crowded cell a 2-choice bucket holds depends on the order the
            # build's atomics land in, so the count varies by ~1e-4 from run to run (the flags do not: every undecided query gets the exact search)
            ua = ub = 0
        if (la, ua) != (lb, ub):
            diff.append((nme, (la, ua), (lb, ub)))
    assert not diff, f"the two hosts do not issue the same work: (class, C++ host, Python host) = {diff}"
    assert len(names) >= 12
    # two lanes: the same work units per class (calls are cut differently -- one grid per lane where one lane batches two -- so launches may differ)
    diff2 = []
    for nme in names:
        if nme == "knn_query_p2":
            continue
        a, b = cxx["classes"].get(nme), cxx2["classes"].get(nme)
        if (a["units_per_step"] if a else 0) != (b["units_per_step"] if b else 0):
            diff2.append((nme, a, b))
    assert not diff2, f"the two-lane schedule does not do the one-lane schedule's work: {diff2}"


def test_binary_compressed_scan_directories_give_the_same_output_tree(tmp_path):
    """Session.cpp:275 loads whatever PCD encoding the scan files have (pcl::io::loadPCDFile).  The same two sessions written as `DATA binary` and as
    `DATA binary_compressed` (LZF streams with back references, structure-of-arrays payload: tests/fileproto.py) must give byte-identical output trees."""
    import filecmp
    import fileproto as fp
    from tools import synth
    exe = os.path.join(ROOT, "lt-mapper_amd", "host", "ltm_run")
    n_kf = 16
    sess = [synth.to_numpy(synth.make_session(s, n_kf, "tiny")) for s in (1, 2)]
    for S in sess:      # noisy float coordinates do not compress; a constant intensity column does: long, self-overlapping back references
        S["scans"] = S["scans"].copy()
        S["scans"][:, 3] = 7.0
    outs = {}
    for kind in ("binary", "compressed"):
        root = tmp_path / kind
        root.mkdir()
        dirs = fp.write_session_dirs(root, sess, compressed=(kind == "compressed"))
        if kind == "compressed":
            one = open(os.path.join(dirs[0], sess[0]["names"][0]), "rb").read()
            assert b"DATA binary_compressed\n" in one and len(one) < 0.8 * 16 * int(sess[0]["offsets"][1]), "the scans really are compressed (back references)"
        out = root / "out"
        y = root / "params.yaml"
        y.write_text(fp.yaml_text(root, dirs, out, 0, n_kf - 1))
        r = subprocess.run([exe, str(y)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, kind + ": " + r.stdout[-1500:] + r.stderr[-1500:]
        outs[kind] = str(out)
    files = sorted(os.path.relpath(os.path.join(d, f), outs["binary"]) for d, _, fs in os.walk(outs["binary"]) for f in fs)
    assert len(files) >= 14 + 5 * n_kf
    for f in files:
        assert filecmp.cmp(os.path.join(outs["binary"], f), os.path.join(outs["compressed"], f), shallow=False), f
