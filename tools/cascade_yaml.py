#!/usr/bin/env python3
"""Next link of the lifelong cascade through the file protocol (reference README.md:115-118 leaves it to
the user).  Given the YAML of run j and the scan directory / pose file of the next query session, writes the YAML of run
j+1: central session := <save_pcd_directory of run j>/scans_updated/ with the pose subset `ltm_run` wrote next to it
(scans_updated_poses.txt), every keyframe used (start_idx 0, end_idx n-1, keyframe_gap 1), outputs under a new directory.

    tools/cascade_yaml.py run1.yaml --query-scans 03/Scans/ --query-poses 03/poses.txt --save out3/ > run2.yaml
"""
import argparse
import os
import re
import sys


def _get(text, key):
    m = re.search(r'^\s*' + re.escape(key) + r'\s*:\s*"?([^"\n#]*?)"?\s*(#.*)?$', text, re.M)
    return m.group(1).strip() if m else None


def _set(text, key, value, quote=True):
    line = f'  {key}: "{value}"' if quote else f"  {key}: {value}"
    pat = re.compile(r'^\s*' + re.escape(key) + r'\s*:.*$', re.M)
    return pat.sub(line, text, count=1) if pat.search(text) else text.rstrip("\n") + "\n" + line + "\n"


def next_yaml(prev_yaml_text, query_scans, query_poses, save_dir, n_keyframes=None):
    prev_save = _get(prev_yaml_text, "save_pcd_directory")
    if not prev_save:
        raise ValueError("the previous YAML has no save_pcd_directory")
    prev_save = prev_save if prev_save.endswith("/") else prev_save + "/"       # Removerter.cpp:26-27 enforces the slash
    if n_keyframes is None:
        poses = prev_save + "scans_updated_poses.txt"
        if not os.path.exists(poses):
            raise FileNotFoundError(poses + " (written by ltm_run at the end of the previous run)")
        n_keyframes = sum(1 for ln in open(poses) if ln.strip())
    out = prev_yaml_text
    out = _set(out, "central_sess_scan_dir", prev_save + "scans_updated/")
    out = _set(out, "central_sess_pose_path", prev_save + "scans_updated_poses.txt")
    out = _set(out, "query_sess_scan_dir", query_scans)
    out = _set(out, "query_sess_pose_path", query_poses)
    out = _set(out, "save_pcd_directory", save_dir)
    out = _set(out, "start_idx", 0, quote=False)
    out = _set(out, "end_idx", n_keyframes - 1, quote=False)
    out = _set(out, "use_keyframe_gap", "true", quote=False)
    out = _set(out, "keyframe_gap", 1, quote=False)
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("prev_yaml")
    ap.add_argument("--query-scans", required=True)
    ap.add_argument("--query-poses", required=True)
    ap.add_argument("--save", required=True)
    a = ap.parse_args()
    sys.stdout.write(next_yaml(open(a.prev_yaml).read(), a.query_scans, a.query_poses, a.save))


if __name__ == "__main__":
    main()
