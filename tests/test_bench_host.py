"""CPU tests of bench.py's host-side logic (no GPU): the distribution statistics of the parity report, the paired
dense / needed-rows comparison, usable-core detection, and the launcher's refusal of a mismatched world size."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BB      # noqa: E402


def test_parity_stats_against_a_known_distribution():
    rng = np.random.RandomState(0)
    r32 = 50 + 100 * rng.rand(64)
    r64 = r32 * (1 - 0.02 + 0.01 * rng.randn(64))           # the reference against itself: ~2 % lower
    ours = r32 * (1 - 0.015 + 0.01 * rng.randn(64))
    st = BB.parity_stats(ours, r32, r64)
    assert st["frames"] == 64
    assert abs(st["final_loss_rel_delta_signed_mean"] - np.mean((ours - r32) / r32)) < 1e-12
    assert abs(st["reference_f64_minus_f32_signed_mean"] - np.mean((r64 - r32) / r32)) < 1e-12
    assert st["frames_below_reference_f32"] + st["frames_above_reference_f32"] == 64
    assert st["final_loss_rel_delta_median"] <= st["final_loss_rel_delta_p90"] <= st["final_loss_rel_delta_max"]
    assert st["paired_wilcoxon_p_vs_reference_f32"] < 0.05                  # a 1.5 % shift over 64 frames is visible
    # a frame the reference disagrees with itself on is listed, not averaged
    st32 = np.stack([r32, r32], 1); st64 = np.stack([r64, r64], 1); st64[5, 0] = 7 * st32[5, 0]
    ok = BB.self_consistent_frames(st32, st64)
    assert ok.sum() == 63 and not ok[5]
    part = BB.parity_stats(ours, r32, r64, scored=ok)
    assert part["frames"] == 63 and part["frames_not_scored"][0]["frame"] == 5
    # identical runs: nothing to report
    same = BB.parity_stats(r32, r32, r64)
    assert same["final_loss_rel_delta_mean"] == 0.0 and same["final_loss_rel_delta_max"] == 0.0


def test_loss_distribution_and_paired_stats_tolerate_non_finite_frames():
    a = np.array([10.0, 12.0, np.nan, 400.0, 11.0])
    b = np.array([10.5, 11.0, 13.0, 9.0, np.inf])
    d = BB.loss_distribution(a)
    assert d["non_finite"] == 1 and d["final_loss_median"] == 11.5 and d["outliers_gt_3x_median"] == 1
    p = BB.paired_stats(a, b, "x", "y")
    assert p["frames"] == 3 and p["frames_without_a_finite_pair"] == 2
    assert p["frames_x_lower"] == 1 and p["frames_y_lower"] == 2 and np.isfinite(p["signed_rel_delta_median"])


def test_usable_cores_is_bounded_by_affinity():
    n, info = BB._usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and info["usable"] == n
    if "affinity" in info:
        assert n <= info["affinity"]


def test_bench_refuses_a_launcher_world_size_that_differs_from_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "must agree" in out.stderr


def test_record_widths_follow_the_configuration():
    """dist.pack_records / unpack_records with num_betas 16, 6 hand components, 20 expression coefficients (the cmd_parser
    default of num_pca_comps is 6): widths come from the result dict, frame index and evaluation count stay exact."""
    from smplifyx_amd import dist as sd
    B = 5
    rng = np.random.RandomState(1)
    res = dict(cam_translation=rng.randn(B, 3), global_orient=rng.randn(B, 3), betas=rng.randn(B, 16),
               left_hand_pose=rng.randn(B, 6), right_hand_pose=rng.randn(B, 6), expression=rng.randn(B, 20),
               jaw_pose=rng.randn(B, 3), leye_pose=rng.randn(B, 3), reye_pose=rng.randn(B, 3), body_pose=rng.randn(B, 63),
               final_loss=rng.rand(B), stage_evals=np.full((B, 6), 3_000_000, np.int64))
    first = 2 ** 24 + 3                                             # beyond float32's exact integers
    rec = sd.pack_records(res, first)
    fields = sd.record_fields(res)
    assert rec.dtype == np.float64 and rec.shape == (B, sum(n for _, n in fields))
    u = sd.unpack_records(rec, fields)
    assert np.array_equal(u["frame"][:, 0], first + np.arange(B)) and np.all(u["evals"][:, 0] == 18_000_000)
    assert u["betas"].shape == (B, 16) and u["left_hand_pose"].shape == (B, 6) and np.array_equal(u["betas"], res["betas"])


def test_benchmark_detector_keeps_three_camera_keypoints_and_matches_the_golden_set():
    """synthetic.make_frames(min_camera_keypoints=3): no frame loses more than one of the camera-initialisation keypoints,
    every other keypoint is what the raw detector gives, shards regenerate the same frames, and the first 64 frames are
    the ones the reference fitted for tests/golden/e2e_bench.npz (bench.py's reference_parity leg relies on that)."""
    from smplifyx_amd import synthetic
    K, cam = 25, (9, 12, 2, 5)
    joints_fn = lambda P: np.tile(np.linspace(-0.5, 0.5, K * 3).reshape(1, K, 3), (len(P["betas"]), 1, 1))   # (any fixed joints)
    raw = synthetic.make_frames(300, joints_fn, K)
    kept = synthetic.make_frames(300, joints_fn, K, min_camera_keypoints=BB.MIN_CAMERA_KEYPOINTS, camera_keypoints=cam)
    present = (kept["keypoints"][:, list(cam), 2] > 0).sum(1)
    assert present.min() >= 3 and (raw["keypoints"][:, list(cam), 2] > 0).sum(1).min() < 3
    other = [j for j in range(K) if j not in cam]
    assert np.array_equal(raw["keypoints"][:, other], kept["keypoints"][:, other])
    changed = np.flatnonzero((raw["keypoints"] != kept["keypoints"]).any((1, 2)))
    assert 0 < len(changed) < 30                                  # ~5 % of the frames
    shard = synthetic.make_frames(40, joints_fn, K, start=100, min_camera_keypoints=3, camera_keypoints=cam)
    assert np.array_equal(shard["keypoints"], kept["keypoints"][100:140])
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_bench.npz"))
    assert (g["keypoints"][:, list(cam), 2] > 0).sum(1).min() >= 3
    assert np.array_equal(g["keypoints"][..., 2] > 0, kept["keypoints"][:64, :, 2] > 0)       # same dropout pattern, frame by frame


def test_raw_sequence_golden_is_the_filtered_one_with_two_frames_replaced():
    """bench.load_bench_golden(raw=True): frames 23 and 51 (two camera-initialisation keypoints dropped on the SURVEY 8(d)
    sequence) carry their raw keypoints and the reference's fits of THOSE; the other 62 frames are shared."""
    import bench
    from smplifyx_amd import synthetic
    f, r = bench.load_bench_golden(False), bench.load_bench_golden(True)
    d = np.abs(f["keypoints"] - r["keypoints"]).reshape(64, -1).max(1)
    assert np.flatnonzero(d > 0).tolist() == [23, 51]
    for i in (23, 51):
        assert not np.array_equal(f["f%d_f32_losses" % i], r["f%d_f32_losses" % i])
        assert (r["keypoints"][i][[9, 12, 2, 5], 2] == 0).sum() == 2      # two of the four camera keypoints missing
    assert np.array_equal(f["f7_f32_losses"], r["f7_f32_losses"])


def _fat_report():
    """A report shaped like bench.py's full one, with everything that once made the line 20 KB: per-frame arrays, per-stage
    lists, paragraph-long notes, NaN / inf values, numpy scalars."""
    rng = np.random.RandomState(0)
    arr = [float(x) for x in rng.rand(64)]
    prose = "x" * 900
    roof = {"kernel": "k_lbs_dense16", "bound": "mfma", "achieved": np.float64(120.9244523), "peak": 157.3, "unit": "TFLOP/s",
            "frac": 0.768750491, "traffic": 89751775.5, "frac_executed": 0.56, "mfma_busy_frac": 0.527, "avg_launch_us": 58.05,
            "launches": np.int64(1665), "frames_per_launch": 153.1, "share_of_step": 0.51, "note": prose,
            "traffic_detail": {"source": prose}}
    return {"metric": "fitted frames/sec", "value": 512.3456789, "unit": "frames/s", "n_gpus": 1, "steps": 3, "warmup": 1,
            "ms_per_step": 499.7123, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 256 synthetic frames/GPU, synthetic SMPL-X, body-only K=25", "keypoints": prose,
                       "arithmetic": prose, "frames_per_gpu": 256, "lbs_mode": "dense", "gemm_columns_per_gpu": 256,
                       "parallelism": "frames sharded, dp1", "closure_evals_per_frame_mean": 2448.1, "closure_evals_per_frame_max": 4429,
                       "closure_evals_per_s": 1254177.1, "final_loss_mean": float("nan"), "final_loss_median": 73.4},
            "roofline": roof, "roofline_tick": dict(roof, kernel="k_tick_dense", bound="hbm", rows_by_stage_class=[11, 53, 225]),
            "roofline_pen": dict(roof, kernel="k_pen_* + k_adj_*"),
            "cpu_baseline": {"value": 0.469, "unit": "frames/s", "cores": 16, "kind": "port", "sample": "16 procs x 14 s",
                             "host": {"a": 1}, "sample_detail": {"t": prose}},
            "reference_parity": {"frames": 63, "final_loss_rel_delta_mean": 0.0161, "final_loss_rel_delta_signed_mean": -0.0027,
                                 "reference_f32_vs_f64_rel_delta_mean": 0.0284, "camera_stage_loss_rel_delta_max": 7.5e-5,
                                 "final_loss": arr, "reference_final_loss_f32": arr, "reference_final_loss_f64": arr, "note": prose,
                                 "per_stage_loss_rel_delta_mean": arr[:6]},
            "closure_parity": {"loss_rel_err_max": 1.18e-7, "grad_rel_err_max": float("inf"), "loss_rel_err_max_per_stage": arr[:6]},
            "value_min3_camera_keypoints": 553.36, "min3_camera_keypoints": {"reference_parity": {"final_loss": arr}},
            "alt": {"lbs_mode": "rows", "value": 1394.27, "unit": "frames/s", "note": prose, "paired_vs_dense": {"x": 1}},
            "kernels_ms_avg": {"lbs_dense": 0.058, "tick_dense": 0.0568, "fit_rows": 0.0},
            "host": {"enqueue_us_per_round": 7.3, "loop_us_per_round": 112.2, "kernels_us_per_round": 115.3, "queue_dry_frac": 0.0, "wait_frac": 0.94},
            "ranks": [{"rank": r, "frames_per_s": 600.0 + r} for r in range(8)], "detail": "gpurun_out/bench_detail_body.json"}


def test_bench_line_is_compact_and_strict_json():
    """The contract line: < 4 KB whatever the report holds (the driver keeps 8 000 bytes of stdout: round 3's 20-KB line was
    cut and counted as unmeasured), strict JSON (no NaN / Infinity tokens), scalars only inside its objects, every
    contract key present, the roofline and cpu_baseline objects with their required members."""
    import json
    full = _fat_report()
    assert len(json.dumps(BB.sanitize(full))) > 8000                       # the long form IS long: it goes to the detail file
    line = BB.compact_line(full)
    txt = json.dumps(line, allow_nan=False)                                  # raises on NaN / inf
    assert len(txt) < BB.LINE_LIMIT, len(txt)
    back = json.loads(txt, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["config"]["workload"].startswith("configs[1]") and back["config"]["final_loss_mean"] is None
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert back["closure_parity"]["grad_rel_err_max"] is None              # inf -> null
    assert back["host"]["loop_us_per_round"] == 112.2
    for name, obj in back.items():                                          # numbers and short tags only
        if isinstance(obj, dict):
            for k, v in obj.items():
                assert not isinstance(v, (list, dict)), (name, k)
                assert not isinstance(v, str) or len(v) <= 160, (name, k)
    assert back["value"] == 512.346 and back["roofline"]["launches"] == 1665
    # a report that is too long even after reduction sheds optional objects, never the contract's
    full["config"]["workload"] = "w" * 5000
    for k in ("roofline", "roofline_tick", "roofline_pen"):
        full[k]["kernel"] = "k" * 5000
    txt = json.dumps(BB.compact_line(full), allow_nan=False)
    assert len(txt) < BB.LINE_LIMIT and "roofline" in json.loads(txt) and "cpu_baseline" in json.loads(txt)


def test_detail_file_is_strict_json(tmp_path, monkeypatch):
    import json
    monkeypatch.setattr(BB, "ROOT", str(tmp_path))
    rel = BB.write_detail(_fat_report(), "unit")
    d = json.load(open(os.path.join(str(tmp_path), rel)), parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert len(d["reference_parity"]["final_loss"]) == 64 and d["config"]["final_loss_mean"] is None


def test_tick_byte_model_follows_the_workload():
    """roofline_tick's algorithmic bytes come from the workload's own vertex items and optimiser width (round 3 divided the
    body-only 11 rows by the full model's kernel time): 11 rows body-only; 11 / 53 / 225 by stage class for coco25 +
    hands + face + contour, 51 of them dynamic-contour items; constants that every frame of a launch shares (static rows,
    VPoser weights twice) count once per launch, like the GEMM's matrix."""
    from smplifyx_amd import utils as U
    jm_body = U.smpl_to_annotation("smplx", use_hands=False, use_face=False, use_face_contour=False, format="coco25")
    jm_full = U.smpl_to_annotation("smplx", use_hands=True, use_face=True, use_face_contour=True, format="coco25")
    assert BB.item_rows_by_class(jm_body, 21, 51, 25) == ([11, 11, 11], [0, 0, 0])
    rows, dyn = BB.item_rows_by_class(jm_full, 21, 51, 25)
    assert rows[0] == 11 and rows[2] == 225 and rows[0] < rows[1] < rows[2] and dyn == [0, 0, 51]
    sh, pf = BB.tick_bytes(11, 0, 119, False)
    assert abs(sh - 11 * (3 * 506 + 16) * 4) < 1e-6 and abs(pf - (2 * 100 * 119 * 4 + 8 * 119 * 4)) < 1e-6
    vp = 4.0 * (512 * 32 + 512 * 512 + 126 * 512 + 512 + 512 + 126)
    sh2, pf2 = BB.tick_bytes(174, 51, 88, True, vp)
    assert sh2 > 2 * vp and abs(sh2 - 2 * vp - 174 * (3 * 506 + 16) * 4) < 1e-3 and pf2 > 51 * 3 * 506 * 4
