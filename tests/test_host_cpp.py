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
