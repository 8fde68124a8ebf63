"""CPU tests of the C++ host mirror (lt-mapper_amd/host): file formats, parameter reader, and that the CLI fails loudly
(no CPU fallback) when no GPU is present."""
import os
import subprocess

from conftest import ROOT

HOST = os.path.join(ROOT, "lt-mapper_amd", "host")


def _build():
    subprocess.check_call(["make", "-s", "-C", ROOT, "hip"])
    subprocess.check_call(["make", "-s", "-C", HOST])


def test_host_selftest_pcd_yaml_pose_voxelgrid(tmp_path):
    _build()
    r = subprocess.run([os.path.join(HOST, "host_selftest"), str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


def test_local_comm_exchanges_session_groups_and_failure_release():
    """lt-mapper_amd/host/src/comm_selftest.cpp: LocalComm on CPU threads (host memory as "device" memory): label MAX all-reduce, all-gather(-v),
    all-to-all-v, the session groups of even worlds with their pair swap, LTM_SESSION_GROUPS=0, and a failing rank releasing the ranks parked in
    the world's and in either group's barrier"""
    _build()
    r = subprocess.run([os.path.join(HOST, "comm_selftest")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


def test_cli_usage_and_loud_failure_without_gpu(tmp_path):
    import torch
    _build()
    exe = os.path.join(HOST, "ltm_run")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([exe, str(tmp_path / "missing.yaml")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot open parameter file" in r.stderr
    if not torch.cuda.is_available():
        y = tmp_path / "p.yaml"
        y.write_text(f"removert:\n  save_pcd_directory: \"{tmp_path}/out/\"\n")
        r = subprocess.run([exe, str(y)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr, r.stderr


def test_cascade_yaml_hand_over(tmp_path):
    """tools/cascade_yaml.py: the YAML of run j+1 points the central session at scans_updated/ + the pose subset of run j"""
    from tools.cascade_yaml import next_yaml
    save = tmp_path / "out1"
    save.mkdir()
    (save / "scans_updated_poses.txt").write_text("1 0 0 0 0 1 0 0 0 0 1 0\n" * 7)
    y1 = f"""removert:
  save_pcd_directory: "{save}"   # no trailing slash
  central_sess_scan_dir: "/data/01/Scans/"
  central_sess_pose_path: "/data/01/poses.txt"
  query_sess_scan_dir: "/data/02/Scans/"
  query_sess_pose_path: "/data/02/poses.txt"
  start_idx: 11
  end_idx: 39
  keyframe_gap: 5
  num_nn_points_within: 2
"""
    y2 = next_yaml(y1, "/data/03/Scans/", "/data/03/poses.txt", str(tmp_path / "out2") + "/")
    assert f'central_sess_scan_dir: "{save}/scans_updated/"' in y2
    assert f'central_sess_pose_path: "{save}/scans_updated_poses.txt"' in y2
    assert 'query_sess_scan_dir: "/data/03/Scans/"' in y2 and 'query_sess_pose_path: "/data/03/poses.txt"' in y2
    assert "start_idx: 0" in y2 and "end_idx: 6" in y2 and "keyframe_gap: 1" in y2 and "use_keyframe_gap: true" in y2
    assert "num_nn_points_within: 2" in y2 and f'save_pcd_directory: "{tmp_path}/out2/"' in y2


def test_loader_voxelgrid_equals_oracle_restatement(tmp_path, orc):
    """Session::loadKeyframes' per-scan pcl::VoxelGrid (host/src/utility.cpp) against the oracle's restatement (orc_voxel_grid, SURVEY
    A.6) on the files themselves: a dense cloud that is really down-sampled (several leaf sizes) and a wide one that takes PCL's
    int32-overflow early-out (output = input) -- bitwise, including the output order by voxel index"""
    import numpy as np
    import fileproto as fp
    _build()
    exe = os.path.join(HOST, "host_selftest")
    rng = np.random.default_rng(3)
    dense = np.concatenate([rng.uniform(-4, 4, (30000, 3)), rng.uniform(0, 255, (30000, 1))], 1).astype(np.float32)
    wide = dense.copy()
    wide[:, :2] *= 30.0
    for name, pts, leafs in (("dense", dense, (0.05, 0.2, 0.5)), ("wide", wide, (0.05,))):
        src = tmp_path / f"{name}.pcd"
        fp.write_pcd(str(src), pts)
        for leaf in leafs:
            dst = tmp_path / f"{name}_{leaf}.pcd"
            r = subprocess.run([exe, "--voxelgrid", str(src), repr(leaf), str(dst)], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr
            got = fp.read_pcd(str(dst))[1]
            want = orc.voxel_grid(pts, leaf)
            assert got.shape == want.shape, f"{name} leaf {leaf}: {got.shape[0]} vs {want.shape[0]} points"
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), f"{name} leaf {leaf}: values / order differ"
            assert (len(got) == len(pts)) == (name == "wide")


def test_ros_wrapper_compiles_and_links_against_api_stubs(tmp_path):
    """lt-mapper_amd/host/ros/src/removert_main_ros.cpp (the catkin node `removert_removert`) cannot be built for real here -- no ROS in the
    image -- but it must at least stay in step with the host mirror it derives from: compile it against minimal stand-ins for the roscpp /
    image_transport / sensor_msgs declarations it uses (tests/ros_stubs) and link it with the same sources CMakeLists.txt lists"""
    import subprocess
    host = os.path.join(ROOT, "lt-mapper_amd", "host")
    src = [os.path.join(host, "ros", "src", "removert_main_ros.cpp")] + [os.path.join(host, "src", f) for f in
          ("utility.cpp", "RosParamServer.cpp", "Session.cpp", "Removerter.cpp", "Comm.cpp")]
    cm = open(os.path.join(host, "ros", "CMakeLists.txt")).read()
    for f in ("utility.cpp", "RosParamServer.cpp", "Session.cpp", "Removerter.cpp", "Comm.cpp", "removert_main_ros.cpp"):
        assert f in cm, f"{f} missing from the catkin target"
    exe = str(tmp_path / "removert_removert")
    cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-pthread", "-I", os.path.join(ROOT, "tests", "ros_stubs"), "-I", host, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "lt-mapper_amd", "csrc"),
           "-o", exe] + src + ["-L", os.path.join(ROOT, "lt-mapper_amd"), "-lltm_hip", "-Wl,-rpath," + os.path.join(ROOT, "lt-mapper_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.exists(exe)
