"""The printed bench line must be parseable from the tail of stdout the driver keeps (round 4's 21 KB line was not, BENCH_r04.parsed = null)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _synthetic_full():
    """a `full` record as bench.py builds it, with every list at its real length and prose in every string field"""
    prose = "x" * 400
    classes = ["vote_map_cull", "reproject_map", "voxel", "merge", "knn_query", "knn_query_p2", "partition", "vote_scan", "vote_compare", "vote_fill", "vote_map_exact",
               "knn_build", "reproject_gather", "voxel_scanset", "voxel_grid_scanset"]
    rl = [{"class": c, "kernels": prose, "bound": "hbm", "ms_per_step": 12.345, "launches_per_step": 21.0, "units_per_step": 1.0e9, "algorithmic_bytes_per_step": 6.2e11,
           "compulsory_bytes_per_step": 1.0e11, "hbm_algorithmic_frac": 1.0071, "hbm_compulsory_frac": 0.1712, "valu_frac": 0.6123, "traffic": 1.03e11, "traffic_scope": prose,
           "hbm_measured_frac": 0.1673, "traffic_over_compulsory": 1.023, "frac": 0.6123, "class_kernels_counted": [prose] * 4} for c in classes]
    return {
        "metric": "keyframe-pairs/sec (removert+diff)", "value": 2921.123, "unit": "keyframe-pairs/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 171.123,
        "cxx_host_ms_per_step": 170.63, "cxx_host_one_shot_steps123_ms": 200.1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "dtype_detail": prose,
        "data": "synthetic (tools/synth.py synth-v1, seed 20250224)",
        "config": {"workload": "lot-2x500-os1-64-3res", "sessions": prose, "keyframes_per_session": 500, "keyframe_pairs_per_step": 500, "sensor": "os1-64",
                   "remove_resolution_list": [2.5, 2.0, 1.5], "self_removert": True, "knn": {"k": 2, "thr": 0.01}, "voxel": 0.05, "map_points_last_pair": [6831259, 6789012],
                   "scan_points": [27000000, 27000000], "parallelism": prose, "step": prose},
        "roofline": dict(rl[0], kernel="k_vote_map_cull", achieved=48.1, peak=78.6, unit="T VALU lane-inst/s", avg_launch_ms=3.672, share_of_step=0.45, pmc=True, pmc_source=prose,
                         definitions=prose, algorithmic_bytes_per_launch=2.9e10, compulsory_bytes_per_launch=5.0e9),
        "cpu_baseline": {"value": 0.3314, "unit": "keyframe-pairs/s", "cores": 1, "kind": "port", "reference_compiled": {"what": prose, "port_speedup_over_reference_compiled": 1.41},
                         "sample": prose * 2, "sample_short": prose, "measured_s": 52.1, "keyframe_stride": 50, "extrapolated_step_s": 1509.0, "host_cores": 256, "cgroup_cpu_quota_cpus": 16.0,
                         "all_cores": {"value": 2.7171, "cores": 256, "quota_note": prose * 2, "cgroup_cpu_quota_cpus": 16.0, "measured_s": 187.0,
                                       "quoted_from": "profiles/cpu_allcore_latest.json (same oracle sources, " + prose + ")"},
                         "full_unsampled_runs_committed": {"1thread": {"cpu": prose}, "allcore": {"cpu": prose}}},
        "rooflines": rl, "traffic_groups": [{"group": prose, "classes": classes}] * 5,
        "t_total": {"what": prose, "configs[1] 2x500 3-res": {"T_total_s": 0.416, "T_step0_s": 0.189, "T_steps123_s": 0.2, "x": prose}, "cxx_host_bench": {"classes": {c: 1.0 for c in classes}}},
        "parity_fullsize": {"record": prose, "matches_sources": True}, "scaling_model": {"what": prose * 5}, "stage_ms": {c: 1.0 for c in classes},
        "kernel_classes_ms_per_step": {c: 1.0 for c in classes}, "voxel_grids": {"what": prose}, "vote_cull": {"fraction": 0.1}, "synth_generation_s": 9.9,
    }


def test_printed_line_is_short_and_parses_from_an_8000_byte_tail():
    import bench
    full = _synthetic_full()
    assert len(json.dumps(full)) > 20000          # the record itself is as large as round 4's line
    line = json.dumps(bench.slim_line(full, "profiles/bench_extra_latest.json"))
    assert len(line) < 6000, len(line)
    # what a driver that keeps the last 8000 bytes of stdout sees: warnings before, the line last
    stdout = ("some runtime warning\n" * 2000 + line + "\n").encode()
    tail = stdout[-8000:].decode()
    got = json.loads(tail.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline", "extra"):
        assert k in got, k
    assert got["config"]["workload"] == "lot-2x500-os1-64-3res"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in got["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k
    assert all(isinstance(v, (int, float, str, bool, type(None))) for v in got["roofline"].values())      # numbers and names, no prose objects
    # a number that was not measured in this run says so in the printed line itself, with the file it comes from
    assert got["cpu_baseline"]["all_cores"]["quoted_from"] == "profiles/cpu_allcore_latest.json" and got["cpu_baseline"]["all_cores"]["measured_now"] is False


def test_the_committed_round4_record_also_slims_down():
    """round 4's actual 21 KB record through the same function"""
    import bench
    path = os.path.join(ROOT, "profiles", "r4_final_bench_default.json")
    full = json.loads(open(path).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 15000
    line = json.dumps(bench.slim_line(full, None))
    assert len(line) < 6000 and json.loads(line)["value"] == full["value"]
