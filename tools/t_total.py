#!/usr/bin/env python3
"""T_total: files -> files wall time of the drop-in `ltm_run` on a BASELINE configuration.

    python tools/t_total.py --kf 500 [--three-res] [--ranks K] [--runs 3] [--keep DIR]

Writes the two synthetic `lot` sessions in the reference's on-disk format (flat directories of binary PCD scans + pose text files,
tools/synth.py), writes a params_ltmapper.yaml, runs lt-mapper_amd/host/ltm_run and reports its own timing line: T_total, the
Step 0 part (directory scan, PCD decode, per-scan VoxelGrid, upload, pre-clean, makeGlobalMap), Steps 1-3 (which include the 16
map PCD writes) and the 5 x N_c scan-file writes.  One JSON line per invocation; the first run warms the page cache."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def measure(sess, kf, three_res=False, ranks=1, runs=3, root=None, extra_yaml="", dirs=None):
    """files -> files of `ltm_run` on keyframes 0..kf-1 of the two sessions `sess` (tools.synth.to_numpy dicts, written in the reference's
    on-disk format under `root` unless `dirs` already holds them).  Returns (result dict, dirs)."""
    import fileproto as fp
    root = root or tempfile.mkdtemp(prefix="ltm_ttotal_")      # the caller removes it
    os.makedirs(root, exist_ok=True)
    t0 = time.perf_counter()
    if dirs is None:
        dirs = fp.write_session_dirs(root, sess)
    t_write_inputs = time.perf_counter() - t0
    in_bytes = sum(int(S["offsets"][min(kf, len(S["offsets"]) - 1)]) * 16 for S in sess)
    exe = os.path.join(ROOT, "lt-mapper_amd", "host", "ltm_run")
    extra = "".join(f"  {kv.strip()}\n" for kv in extra_yaml.split(";") if kv.strip())
    if three_res:
        extra += "  gpu_use_self_removert: true\n"
    out_runs = []
    for r in range(runs):
        outdir = os.path.join(root, f"out{r}")
        shutil.rmtree(outdir, ignore_errors=True)
        yaml = os.path.join(root, "params.yaml")
        with open(yaml, "w") as f:
            f.write(fp.yaml_text(root, dirs, outdir, 0, kf - 1, res_list=(2.5, 2.0, 1.5) if three_res else (2.5,), extra=extra))
        cmd = [exe, yaml] + (["--logical-ranks", str(ranks)] if ranks > 1 else [])
        t0 = time.perf_counter()
        p = subprocess.run(cmd, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            raise RuntimeError(p.stdout[-2000:] + p.stderr[-2000:])
        line = [l for l in p.stdout.splitlines() if l.startswith("[timing]")][-1].split()
        t = {line[i]: float(line[i + 1]) for i in range(len(line) - 1) if line[i].startswith("T_")}
        kfs = [int(x) for x in line[line.index("keyframes") + 1: line.index("keyframes") + 3]]
        out_bytes = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(outdir) for f in fs)
        diag = [l for l in p.stderr.splitlines() if l.startswith("[ltm]")]         # LTM_POOL_STATS=1: allocator statistics of the run
        out_runs.append(dict(process_wall_s=round(wall, 3), **{k: round(v, 3) for k, v in t.items()}, keyframes=kfs, output_bytes=out_bytes,
                             **({"diagnostics": diag} if diag else {})))
        shutil.rmtree(outdir, ignore_errors=True)
    best = min(out_runs, key=lambda x: x["T_total"])
    res = {"what": "ltm_run files -> files", "keyframes_per_session": kf, "three_res": three_res, "ranks": ranks, "input_bytes": in_bytes,
           "runs": out_runs, "best": best, "keyframe_pairs_per_s_incl_io": round(min(best["keyframes"]) / best["T_total"], 2),
           "T_total_minus_steps_s": round(best["T_total"] - best["T_steps123"], 3), "input_write_s": round(t_write_inputs, 1)}
    return res, dirs


def bench_cxx_host(root, dirs, kf, three_res=True, steps=3, warmup=1, lanes=None):
    """`ltm_run <yaml> --bench steps`: makeGlobalMap + Steps 1-3 timed by the C++ host itself, the loaded sessions resident on the device, no
    output files -- the same timed region as bench.py's, driven by the north-star host.  Returns its JSON line as a dict."""
    import fileproto as fp
    exe = os.path.join(ROOT, "lt-mapper_amd", "host", "ltm_run")
    outdir = os.path.join(root, "out_bench")
    yaml = os.path.join(root, "params_bench.yaml")
    with open(yaml, "w") as f:
        f.write(fp.yaml_text(root, dirs, outdir, 0, kf - 1, res_list=(2.5, 2.0, 1.5) if three_res else (2.5,),
                             extra=("  gpu_use_self_removert: true\n" if three_res else "") + (f"  gpu_lanes: {int(lanes)}\n" if lanes else "")))
    p = subprocess.run([exe, yaml, "--bench", str(steps), "--warmup", str(warmup)], capture_output=True, text=True)
    shutil.rmtree(outdir, ignore_errors=True)
    if p.returncode != 0:
        raise RuntimeError(p.stdout[-1500:] + p.stderr[-1500:])
    line = [l for l in p.stdout.splitlines() if l.startswith("[bench] ")][-1]
    return json.loads(line[len("[bench] "):])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kf", type=int, default=50)
    ap.add_argument("--sensor", default="os1-64")
    ap.add_argument("--three-res", action="store_true")
    ap.add_argument("--ranks", type=int, default=1, help="--logical-ranks K of ltm_run (1 = plain single-GPU run)")
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--keep", default=None)
    ap.add_argument("--extra-yaml", default="", help="additional `key: value` lines, ';'-separated")
    args = ap.parse_args()
    import torch
    from tools import synth
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    root = args.keep or tempfile.mkdtemp(prefix="ltm_ttotal_")
    t0 = time.perf_counter()
    sess = [synth.to_numpy(synth.make_session(s, args.kf, args.sensor, device=dev)) for s in (1, 2)]
    t_gen = time.perf_counter() - t0
    try:
        res, _ = measure(sess, args.kf, args.three_res, args.ranks, args.runs, root=root, extra_yaml=args.extra_yaml)
    except RuntimeError as e:
        sys.exit(str(e))
    res.update(sensor=args.sensor, input_generation_s=round(t_gen, 1))
    print(json.dumps(res))
    if not args.keep:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
