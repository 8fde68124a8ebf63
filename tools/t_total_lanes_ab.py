"""files -> files of `ltm_run` on configs[1] (one-shot process: every pool cold) with gpu_lanes 1 against 2, alternating, same box; one run each with LTM_POOL_STATS=1.
    python tools/t_total_lanes_ab.py [--kf 500] [--rounds 3] > gpurun_out/<name>.json        (on the GPU box; test / measurement helper)"""
import argparse
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kf", type=int, default=500)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    from tools import synth, t_total
    sess = [synth.to_numpy(synth.make_session(s, a.kf, "os1-64", device="cuda")) for s in (1, 2)]
    root = tempfile.mkdtemp(prefix="ltm_lanes_ab_")
    out = {"what": __doc__.splitlines()[0], "keyframes_per_session": a.kf, "runs": []}
    try:
        dirs = None
        _, dirs = t_total.measure(sess, a.kf, three_res=True, runs=1, root=root)      # writes the inputs, warms the page cache and the binary
        for r in range(a.rounds):
            for lanes in (1, 2):
                res, _ = t_total.measure(sess, a.kf, three_res=True, runs=1, root=root, dirs=dirs, extra_yaml=f"gpu_lanes: {lanes}")
                b = res["best"]
                out["runs"].append({"round": r, "lanes": lanes, **{k: b[k] for k in b if k.startswith("T_") or k == "process_wall_s"}})
        for lanes in (1, 2):
            os.environ["LTM_POOL_STATS"] = "1"
            res, _ = t_total.measure(sess, a.kf, three_res=True, runs=1, root=root, dirs=dirs, extra_yaml=f"gpu_lanes: {lanes}")
            os.environ.pop("LTM_POOL_STATS", None)
            out[f"pool_stats_lanes_{lanes}"] = res["best"].get("diagnostics")
        for lanes in (1, 2):
            v = [x for x in out["runs"] if x["lanes"] == lanes]
            out[f"median_lanes_{lanes}"] = {k: sorted(x[k] for x in v)[len(v) // 2] for k in v[0] if k.startswith("T_")}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
