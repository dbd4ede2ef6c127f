"""Edge cases of the HIP path (run with -m gpu): ragged batch sizes, batch-composition
independence of the dense (MFMA) path, degenerate frames, the halpe format."""
import numpy as np
import pytest
import torch

import helpers as H
import test_gpu_parity as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg_body():
    return H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)


@pytest.mark.parametrize("B", [33, 100, 130, 250])
def test_dense_path_is_batch_composition_independent(synth_model, cfg_body, B):
    """Column position in the GEMM operands, frame-block split (1 .. 4 blocks of 64 frames, partial slices) and compaction
    must not change a frame's numbers: frame 2 alone == frame 2 as the LAST of B."""
    cfg = dict(cfg_body); cfg["use_camera_prior"] = False
    dm = T._dm(synth_model, cfg)
    frames = T.synth_frames(synth_model, cfg, 3)
    idx = [i % 2 for i in range(B - 1)] + [2]
    out = []
    for ids in ([2], idx):
        fb = H.engine_batch_from_frames(dm, cfg, frames, ids, lbs_mode="dense", reuse=True)
        fb.guess_init(cfg["body_tri_idxs"])
        l, g = fb.closure(0)
        fb.fit(first_stage=-1, last_stage=1)
        out.append((l[-1], g[-1].copy(), {k: v[-1].copy() for k, v in fb.get_params().items()}, fb.stats()))
        fb.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    for k in out[0][2]:
        assert np.array_equal(out[0][2][k], out[1][2][k]), k
    assert np.array_equal(out[0][3]["stage_evals"][-1], out[1][3]["stage_evals"][-1])


@H.requires_lab()
def test_kernel_shapes_of_the_dense_loop_give_the_same_bits():
    """Round 4 added launch shapes to both kernels of the dense loop, each chosen by the number of active frames: k_lbs_dense16<W>
    with W = 3 / 4 / 5 wavefronts per workgroup, k_tick_dense on eight wavefronts (<= 256 frames) or four.  A frame's fit must not
    depend on the shape: the same 90-frame job (6 slices: W = 3 by the launcher's rule; then 5, 4, ... as frames finish) with the
    launcher's choices, with one width forced for every launch (SFX_LBS_W = 3, 4, 5) and with the four-wavefront tick kernel
    (SFX_TICK_THREADS = 256) -- the switches are read once per process, hence the subprocesses -- bit for bit."""
    import hashlib, json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import hashlib, json, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import helpers as H, test_gpu_parity as T
from smplifyx_amd import synthetic
model = synthetic.make_synthetic_model(0)
cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False); cfg["use_camera_prior"] = False
dm = T._dm(model, cfg)
frames = T.synth_frames(model, cfg, 5)
fb = H.engine_batch_from_frames(dm, cfg, frames, [i %% 5 for i in range(90)], lbs_mode="dense", reuse=True)
fb.guess_init(cfg["body_tri_idxs"])
fb.fit(first_stage=-1, last_stage=1)
P = fb.get_params(); st = fb.stats()
h = hashlib.sha256()
for k in sorted(P): h.update(np.ascontiguousarray(P[k]).tobytes())
h.update(np.ascontiguousarray(st["stage_evals"]).tobytes()); h.update(np.ascontiguousarray(st["stage_loss"]).tobytes())
v, j = fb.forward()
h.update(v.cpu().numpy().tobytes())
print(json.dumps({"sha": h.hexdigest(), "evals": int(np.asarray(st["stage_evals"]).sum())}))
''' % (root, os.path.join(root, "tests"))
    outs = {}
    for name, env_extra in (("default", {}), ("w3", {"SFX_LBS_W": "3"}), ("w4", {"SFX_LBS_W": "4"}), ("w5", {"SFX_LBS_W": "5"}),
                            ("tick256", {"SFX_TICK_THREADS": "256"})):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert outs["default"]["evals"] > 90 * 50, outs
    for name, o in outs.items():
        assert o == outs["default"], (name, outs)


@H.requires_lab()
def test_the_two_dense_kernels_are_interchangeable_bit_for_bit(synth_model, cfg_body):
    """k_lbs_dense16 (16 frames per wavefront, the product kernel) and k_lbs_dense (32 per wavefront, kept for A/B
    measurements) put every (vertex, frame) through the same chain of fp32 operations: all vertices of a 100-frame launch
    and a whole two-stage fit must agree bit for bit."""
    from smplifyx_amd import _capi
    lib = _capi.load()
    cfg = dict(cfg_body); cfg["use_camera_prior"] = False
    dm = T._dm(synth_model, cfg)
    frames = T.synth_frames(synth_model, cfg, 3)
    idx = [i % 3 for i in range(100)]
    out = []
    prev = lib.sfx_debug_lbs_dense_form(0)
    try:
        for form in (16, 32):
            lib.sfx_debug_lbs_dense_form(form)
            assert lib.sfx_debug_lbs_dense_form(0) == form
            fb = H.engine_batch_from_frames(dm, cfg, frames, idx, lbs_mode="dense", reuse=True)
            fb.guess_init(cfg["body_tri_idxs"])
            l, g = fb.closure(0)
            verts = fb.debug_read("verts").copy()
            fb.fit(first_stage=-1, last_stage=1)
            out.append((l.copy(), g.copy(), verts, {k: v.copy() for k, v in fb.get_params().items()}, fb.stats()["stage_evals"].copy()))
            fb.close()
    finally:
        lib.sfx_debug_lbs_dense_form(prev)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert np.abs(out[0][2]).max() > 0.1
    for k in out[0][3]:
        assert np.array_equal(out[0][3][k], out[1][3][k]), k
    assert np.array_equal(out[0][4], out[1][4])


@H.requires_lab()
@pytest.mark.parametrize("n", [1, 9, 16, 17, 32])
def test_the_few_frames_dense_kernel_is_interchangeable_bit_for_bit(synth_model, cfg_body, n):
    """k_lbs_dense16c (round 5: at <= 32 active frames the three coordinates of a 16-frame slice go to three wavefronts -- the
    launch floor) against k_lbs_dense16 (form 17 = without it) and k_lbs_dense (form 32): vertices, exported keypoint offsets
    (through the closure's loss and gradient) and a two-stage fit, bit for bit, at 1, 9, 16, 17 and 32 frames."""
    from smplifyx_amd import _capi
    lib = _capi.load()
    cfg = dict(cfg_body); cfg["use_camera_prior"] = False
    dm = T._dm(synth_model, cfg)
    frames = T.synth_frames(synth_model, cfg, 3)
    idx = [i % 3 for i in range(n)]
    out = []
    prev = lib.sfx_debug_lbs_dense_form(0)
    try:
        for form in (16, 17, 32):
            lib.sfx_debug_lbs_dense_form(form)
            assert lib.sfx_debug_lbs_dense_form(0) == form
            fb = H.engine_batch_from_frames(dm, cfg, frames, idx, lbs_mode="dense", reuse=True)
            fb.guess_init(cfg["body_tri_idxs"])
            l, g = fb.closure(0)
            verts = fb.debug_read("verts").copy()
            fb.fit(first_stage=-1, last_stage=1)
            out.append((l.copy(), g.copy(), verts, {k: v.copy() for k, v in fb.get_params().items()}, fb.stats()["stage_evals"].copy()))
            fb.close()
    finally:
        lib.sfx_debug_lbs_dense_form(prev)
    assert np.abs(out[0][2]).max() > 0.1
    for other in (1, 2):
        assert np.array_equal(out[0][0], out[other][0]) and np.array_equal(out[0][1], out[other][1]) and np.array_equal(out[0][2], out[other][2]), other
        for k in out[0][3]:
            assert np.array_equal(out[0][3][k], out[other][3][k]), (other, k)
        assert np.array_equal(out[0][4], out[other][4])


def test_dense_and_rows_agree_per_closure(synth_model, cfg_body):
    """The needed-rows kernel and the all-vertices MFMA kernel evaluate the same objective: loss
    and gradient agree to fp32 rounding (both are compared with the oracle elsewhere)."""
    cfg = dict(cfg_body); cfg["use_camera_prior"] = False
    dm = T._dm(synth_model, cfg)
    frames = T.synth_frames(synth_model, cfg, 3)
    res = {}
    for mode in ("rows", "dense"):
        fb = H.engine_batch_from_frames(dm, cfg, frames, range(3), lbs_mode=mode)
        fb.guess_init(cfg["body_tri_idxs"])
        res[mode] = [fb.closure(s) for s in (-1, 0, 2)]
        fb.close()
    for (lr, gr), (ld, gd) in zip(res["rows"], res["dense"]):
        assert np.allclose(lr, ld, rtol=1e-5)          # different fp32 summation orders (MFMA k-chain vs wave scan)
        assert np.linalg.norm(gr - gd) <= 1e-4 * np.linalg.norm(gr)


@pytest.mark.parametrize("mode", ["rows", "dense"])
def test_degenerate_frames_do_not_poison_the_batch(synth_model, cfg_body, mode):
    """A frame without any detection (all keypoints and confidences zero): guess_init divides by a
    zero 2-D limb length as the reference does (fitting.py:61-67 -> t_z = inf), the camera stage
    sees (inf - inf)^2 = NaN and run_fitting stops it at once returning None (fitting.py:177-183,216);
    the body stages then only see the priors (every keypoint weight is zero).  The frame must
    terminate and leave its neighbours untouched."""
    from smplifyx_amd import driver
    cfg = dict(cfg_body); cfg["use_camera_prior"] = False
    dm = T._dm(synth_model, cfg)
    g = T._golden("e2e_synth")
    kp = np.concatenate([g["keypoints"][:2], np.zeros((1, 25, 3), np.float32)])
    rp = np.concatenate([g["reg_pose"][:2], g["reg_pose"][:1]])
    rg = np.concatenate([g["reg_global"][:2], g["reg_global"][:1]])
    jw = H.base_joint_weights(cfg, 25)
    res = driver.fit_frames(dm, cfg, kp, jw, 600, 800, 5000.0, reg_pose=rp, reg_global=rg, lbs_mode=mode)
    ref = driver.fit_frames(dm, cfg, kp[:2], jw, 600, 800, 5000.0, reg_pose=rp[:2], reg_global=rg[:2], lbs_mode=mode)
    for k in ("cam_translation", "global_orient", "betas", "pose_embedding", "final_loss"):
        assert np.array_equal(res[k][:2], ref[k]), k
    assert np.isnan(res["stage_loss"][2, 0]) and res["stage_evals"][2, 0] <= 2     # camera stage: None in the reference
    assert np.isinf(res["cam_translation"][2, 2])
    assert np.all(np.isfinite(res["stage_loss"][2, 1:])) and np.all(np.isfinite(res["pose_embedding"][2]))
    assert np.all(res["stage_loss"][2, 1:] < 1e5)                                  # priors only


def test_halpe_closure_matches_oracle(synth_model):
    """cfg_files/fit_smplx_combined_halpe.yaml (K = 26 body joints, 3 stages, confidences used in
    the camera initialisation) without its interpenetration term: loss and gradient vs oracle."""
    cfg = H.load_cfg("fit_smplx_combined_halpe.yaml", use_hands=False, use_face=False, interpenetration=False)
    assert cfg["format"] == "halpe" and len(H.joint_map_for(cfg)) == 26
    dm = T._dm(synth_model, cfg)
    B = 2
    frames = T.synth_frames(synth_model, cfg, B) if False else None
    from smplifyx_amd import synthetic
    K = len(H.joint_map_for(cfg))
    frames = synthetic.make_frames(B, H.oracle_joints_fn(synth_model, cfg), K, focal=5000.0)
    fb = H.engine_batch_from_frames(dm, cfg, frames, range(B), lbs_mode="rows")
    rng = np.random.RandomState(3)
    P = H.random_params(rng, B, scale=0.5)
    P["pose_embedding"] = frames["reg_pose"] + 0.1 * rng.normal(size=(B, 63)).astype(np.float32)
    P["global_orient"] = frames["reg_global"] + 0.1 * rng.normal(size=(B, 3)).astype(np.float32)
    P["cam_translation"] = (frames["cam_t"] + 0.3 * rng.normal(size=(B, 3))).astype(np.float32)
    est = (frames["cam_t"][:, 2] + 1.0).astype(np.float32)
    kp = frames["keypoints"]
    thr = np.array([cfg.get("confidence_threshold", 0)] * 26 + [0] * 110)[:K]
    jw = np.tile(H.base_joint_weights(cfg, K), (B, 1)); jw[kp[:, :, 2] < thr[None]] = 0
    cm = np.zeros((B, K), np.float32)
    for b in range(B):
        for j in cfg["init_joints_idxs"]:
            if kp[b, j, 0] != 0 and kp[b, j, 1] != 0 and not kp[b, j, 2] < thr[j]:
                cm[b, j] = 1
    fb.set_frames(kp, jw, cm, frames["focal"], np.tile([frames["W"] * 0.5, frames["H"] * 0.5], (B, 1)),
                  1000.0 / frames["H"], est_tz=est)
    fb.set_params(regression_pose=frames["reg_pose"], **P)
    P["est_tz"] = est
    assert fb.n_stages == 3
    for stage in (-1, 0, 2):
        loss, grad = fb.closure(stage)
        for i in range(B):
            lo, go = T._oracle_closure(synth_model, cfg, frames, i, P, stage)
            H.check_closure("halpe-full-rows", stage, loss[i], lo, grad[i], go)


@pytest.mark.parametrize("n,extra", [(2, ["--lbs", "rows"]), (2, ["--workload", "pen"]), (4, [])], ids=["body-rows-2", "pen-dense-2", "body-dense-4"])
def test_bench_two_rank_control_flow_rehearsal(n, extra):
    """`python bench.py --gpus 2` (it launches its own 2 ranks; both on GPU 0, gloo, in this rehearsal): the N > 1 control
    flow -- per-rank frame blocks, barriers, max-over-ranks time, the record gather -- produces one
    JSON line whose frame count is the whole job's.  Also for BASELINE configs[4] (the halpe cfg with the interpenetration
    term, which the 8-GPU run shards exactly like this: per-rank camera priors, per-rank collision buffers)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SFX_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    # exactly what the driver types, no launcher: bench.py starts its ranks itself
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
           "--frames", "32", "--no-cpu", "--no-alt"] + extra
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["frames_per_gpu"] == 32 and d["scaling"] == "weak"
    assert abs(d["value"] - n * 32 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-4 * d["value"]      # (both figures are printed to 6 digits)
    assert len(lines[0]) < 4096
    # per-rank records live in the detail file; the line carries their summary and the single-GPU rate of the SAME job size
    # (rank 0's own frames over its own time), which is what 1 -> N efficiency is computed against
    ranks = json.load(open(os.path.join(root, d["detail"])))["ranks"]
    assert [r["rank"] for r in ranks] == list(range(n)) and all(r["closure_evals_total"] > 0 for r in ranks)
    c = d["config"]
    assert c["per_gpu_frames_per_s_min"] <= c["per_gpu_frames_per_s_mean"] and c["single_gpu_same_job_frames_per_s"] == pytest.approx(ranks[0]["frames_per_s"], rel=1e-4)
    assert d["value"] <= n * c["per_gpu_frames_per_s_mean"] * 1.001          # the job's rate is bound by its slowest rank
    if "pen" in extra:
        assert d["config"]["workload"].startswith("configs[4]") and d["roofline_pen"]["pairs_per_column"] >= 0
        return
    if n != 2:
        return
    # ... and under an external launcher whose world size disagrees with --gpus it refuses instead of mis-reporting
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=root, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "must agree" in bad.stderr


def test_bench_rccl_calls_single_rank():
    """bench.py under torch.distributed.run with ONE rank and SFX_FORCE_COLLECTIVE=1: the RCCL calls of the
    N > 1 path (init_process_group('nccl', device_id=...), barrier, all_reduce(MAX) of the float64 time on the
    device, the all_gather of the result records as device tensors) against the real backend."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SFX_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1")
    env.pop("SFX_BENCH_REHEARSAL", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
           "--frames", "32", "--no-cpu", "--no-alt", "--no-parity"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["frames_per_gpu"] == 32 and d["value"] > 0


@pytest.mark.parametrize("workload", ["body", "full", "pen"])
def test_bench_line_contract(workload):
    """bench.py on a small batch: one JSON line with the contract's keys, the roofline object of the dense
    kernel, and (headline workload) the loss delta against the reference's own fits of the golden frames."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--frames", "32", "--no-cpu",
           "--no-alt", "--workload", workload]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    # the driver parses this line out of the last 8 000 bytes of stdout: it must stay small and strict (round 3's was 20 KB)
    assert len(lines[0]) < 4096 and len(out.stdout) < 8000, (len(lines[0]), len(out.stdout))
    d = json.loads(lines[0], parse_constant=lambda c: (_ for _ in ()).throw(ValueError("non-strict JSON: " + c)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    for name, obj in d.items():
        if isinstance(obj, dict):
            assert all(not isinstance(v, (list, dict)) for v in obj.values()), name
    detail = json.load(open(os.path.join(root, d["detail"])))
    assert detail["value"] == pytest.approx(d["value"], rel=1e-5) and "note" in detail["roofline"]
    assert d["unit"] == "frames/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) < 1e-4 * d["value"]       # (both figures are printed to 6 digits)
    r = d["roofline"]
    assert r["kernel"] == "k_lbs_dense16" and r["bound"] == "mfma" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert np.isfinite(d["config"]["final_loss_mean"])
    h = d["host"]        # the host thread's side of the loop: enqueue time and the loop's wall time per round next to the kernels'
    assert 0 < h["enqueue_us_per_round"] < h["loop_us_per_round"] and 0 <= h["queue_dry_frac"] < 1 and h["kernels_us_per_round"] > 0
    assert 0 <= h["outside_loop_ms_per_step"] < d["ms_per_step"]          # batch set-up, collection, gather: what a busy host inflates
    if workload == "body":
        rp = d["reference_parity"]
        assert rp["frames"] >= 32 and rp["camera_stage_loss_rel_delta_max"] < 2e-4
        assert abs(rp["final_loss_rel_delta_signed_mean"]) <= rp["reference_f32_vs_f64_rel_delta_mean"]
        assert rp["final_loss_rel_delta_median"] <= 1.5 * rp["reference_f32_vs_f64_rel_delta_median"]
        assert "roofline_tick" in d and d["roofline_tick"]["kernel"] == "k_tick_dense"
        assert d["roofline_tick"]["rows_per_frame_launch"] == pytest.approx(11.0)
        # configs[3]'s per-GPU job (1 024 frames through the default 512-column pool) behind the headline: the equal-job denominator
        c3 = detail["config"]["configs3_single_gpu"]
        assert c3["frames"] == 1024 and c3["gemm_columns"] == 512 and d["config"]["configs3_single_gpu_frames_per_s"] == pytest.approx(c3["frames_per_s"], rel=1e-4)
        assert c3["frames_per_s"] > d["value"]             # (1 024 frames fill the GEMM's columns better than 32)
    if workload == "full":       # the byte model of the per-frame kernel is this workload's: live items by stage, VPoser weights twice
        rt = d["roofline_tick"]
        assert 11.0 < rt["rows_per_frame_launch"] <= 225.0 and rt["shared_bytes_per_launch"] > 2 * 1.3e6
        assert rt["l2_stream_bytes_per_frame"] > rt["shared_bytes_per_launch"] and 0 < rt["frac"] < 0.2
    if workload == "pen":        # grid entries and pairs are COUNTED over the timed region, not typical constants
        rp = d["roofline_pen"]
        assert rp["grid_entries_per_column"] > 0 and rp["pairs_per_column"] >= 0 and rp["columns_per_launch"] > 0
        assert rp["frac"] == pytest.approx(rp["achieved"] / rp["peak"], rel=1e-4)


@pytest.mark.parametrize("mode", ["rows", "dense"])
def test_vertices_with_many_skinning_weights(synth_model, cfg_body, mode):
    """lbs_weights rows with more nonzeros than the packed 8-entry form (learned weights of a real
    model may have them): the needed-rows path falls back to the full row, the dense path's tile
    lists cover any count.  Loss and gradient vs oracle on a model whose keypoint vertices carry 12
    nonzero weights."""
    cfg = dict(cfg_body); cfg["use_camera_prior"] = False
    model = dict(synth_model)
    W = np.array(model["weights"], np.float32).copy()
    rng = np.random.RandomState(4)
    vids = [int(v) for v in np.asarray(list(model["extra_vertex_ids"].values()) if isinstance(model["extra_vertex_ids"], dict)
                                       else model["extra_vertex_ids"]).reshape(-1)]
    for v in vids:
        js = rng.choice(55, 12, replace=False)
        w = rng.uniform(0.2, 1.0, 12).astype(np.float32)
        W[v] = 0
        W[v, js] = w / w.sum()
    model["weights"] = W
    assert (np.count_nonzero(W[vids], axis=1) == 12).all()
    dm = T._dm(model, cfg)
    B = 2
    from smplifyx_amd import synthetic
    frames = synthetic.make_frames(B, H.oracle_joints_fn(model, cfg), 25, focal=5000.0)
    fb = H.engine_batch_from_frames(dm, cfg, frames, range(B), lbs_mode=mode)
    rng = np.random.RandomState(12)
    P = H.random_params(rng, B, scale=0.5)
    P["pose_embedding"] = frames["reg_pose"] + 0.1 * rng.normal(size=(B, 63)).astype(np.float32)
    P["global_orient"] = frames["reg_global"] + 0.1 * rng.normal(size=(B, 3)).astype(np.float32)
    P["cam_translation"] = (frames["cam_t"] + 0.3 * rng.normal(size=(B, 3))).astype(np.float32)
    est = (frames["cam_t"][:, 2] + 1.0).astype(np.float32)
    fb.set_frames(frames["keypoints"], T._jw(cfg, frames), T._cmask(cfg, frames), frames["focal"],
                  np.tile([frames["W"] * 0.5, frames["H"] * 0.5], (B, 1)), 1000.0 / frames["H"], est_tz=est)
    fb.set_params(regression_pose=frames["reg_pose"], **P)
    P["est_tz"] = est
    for stage in (-1, 1):
        loss, grad = fb.closure(stage)
        for i in range(B):
            lo, go = T._oracle_closure(model, cfg, frames, i, P, stage)
            H.check_closure("12-weights-%s" % mode, stage, loss[i], lo, grad[i], go)


@pytest.mark.parametrize("mode", ["rows", "dense"])
def test_use_pca_false_closure_matches_oracle(synth_model, mode):
    """use_pca=False (cmd_parser.py:127; smplx.SMPLX): 45 + 45 hand pose variables.  248 optimisation variables with the dead
    body_pose parameter, which the device leaves out of its 192-wide vectors (185): closure against the oracle built with
    use_pca=False, gradient compared on the live variables."""
    from oracle.body_model import SMPLXRef
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml")
    cfg["use_camera_prior"] = False
    from smplifyx_amd import engine
    dm = engine.DeviceModel(synth_model, joint_map=H.joint_map_for(cfg), num_betas=10, num_expression_coeffs=10,
                            use_face_contour=cfg["use_face_contour"], use_pca=False)
    assert dm.num_pca == 45
    B = 2
    frames = T.synth_frames(synth_model, cfg, 3)
    frames = {k: (v[:B] if isinstance(v, np.ndarray) else v) for k, v in frames.items()}
    fb = H.engine_batch_from_frames(dm, cfg, frames, range(B), lbs_mode=mode)
    assert fb.num_vars(0) == 248 - 63
    rng = np.random.RandomState(21)
    P = H.random_params(rng, B, scale=0.5, npca=45)
    P["left_hand_pose"] *= 0.2; P["right_hand_pose"] *= 0.2
    P["pose_embedding"] = frames["reg_pose"] + 0.1 * rng.normal(size=(B, 63)).astype(np.float32)
    P["global_orient"] = frames["reg_global"] + 0.1 * rng.normal(size=(B, 3)).astype(np.float32)
    P["cam_translation"] = (frames["cam_t"] + 0.3 * rng.normal(size=(B, 3))).astype(np.float32)
    est = (frames["cam_t"][:, 2] + 1.0).astype(np.float32)
    fb.set_frames(frames["keypoints"], T._jw(cfg, frames), T._cmask(cfg, frames), frames["focal"],
                  np.tile([frames["W"] * 0.5, frames["H"] * 0.5], (B, 1)), 1000.0 / frames["H"], est_tz=est)
    fb.set_params(regression_pose=frames["reg_pose"], **P)
    P["est_tz"] = est
    orig = H.oracle_model
    H.oracle_model = lambda model, cfg_, dtype=torch.float32: SMPLXRef(
        model, joint_map=H.joint_map_for(cfg_), num_betas=cfg_["num_betas"], num_expression_coeffs=cfg_["num_expression_coeffs"],
        use_pca=False, use_face_contour=cfg_["use_face_contour"], create_body_pose=True, dtype=dtype)
    try:
        for stage in (-1, 0, 2):
            loss, grad = fb.closure(stage)
            for i in range(B):
                lo, go = T._oracle_closure(synth_model, cfg, frames, i, P, stage)
                if stage >= 0:
                    assert go.size == 248 and np.all(go[13:76] == 0)
                    go = np.concatenate([go[:13], go[76:]])          # (the dead body_pose parameter is not a device variable here)
                H.check_closure("pca-off-full-%s" % mode, stage, loss[i], lo, grad[i], go)
    finally:
        H.oracle_model = orig


def test_high_precision_mode_closure_matches_oracle(synth_model):
    """cfg float_dtype: float64 -> sfx_batch_cfg.high_precision: projection in fp64 in every stage.  Same objective, so the
    same bounds against fp64 autograd of the oracle; and the loss moves by less than fp32 pixel rounding allows."""
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)
    cfg["use_camera_prior"] = False
    cfg64 = dict(cfg, float_dtype="float64")
    out = {}
    for tag, c in (("f32", cfg), ("hp", cfg64)):
        out[tag] = T.closure_probe(synth_model, c, "dense", "high-precision-" + tag)
    for st in out["hp"]:
        assert out["hp"][st][0] <= 2e-6 and out["hp"][st][1] <= 6e-6


def test_standalone_loss_forward_matches_the_closure(synth_model):
    """SMPLifyLoss.forward(body_model_output, camera=..., gt_joints=..., ...) and SMPLifyCameraInitLoss.forward, called the
    way fitting.py:248-259 calls them, return the device objective at the parameters the ModelOutput was made from:
    equal to the fitting closure's value bit for bit, and to the oracle within the closure bounds."""
    import test_gpu_dropin as DI
    from smplifyx_amd import fitting, prior
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)
    cfg["use_camera_prior"] = False
    bm, camera = DI._setup(synth_model, cfg)
    dev = torch.device("cuda")
    frames = T.synth_frames(synth_model, cfg, 3)
    i = 1
    kd = torch.tensor(frames["keypoints"][i:i + 1], device=dev)
    gt_joints, joints_conf = kd[:, :, :2], kd[:, :, 2].reshape(1, -1)
    joint_weights = torch.tensor(T._jw(cfg, frames)[i:i + 1], device=dev)
    pose_embedding = torch.tensor(frames["reg_pose"][i:i + 1] + 0.05, device=dev, requires_grad=True)
    rng = np.random.RandomState(5)
    bm.reset_params(global_orient=frames["reg_global"][i:i + 1], body_pose=pose_embedding.detach(),
                    betas=0.5 * rng.normal(size=(1, 10)).astype(np.float32))
    with torch.no_grad():
        camera.translation[:] = torch.tensor(frames["cam_t"][i:i + 1] + 0.1, device=dev)
        camera.center[:] = torch.tensor([frames["W"] * 0.5, frames["H"] * 0.5], device=dev)
    mk = lambda t: prior.create_prior(prior_type=t, dtype=torch.float32)
    loss = fitting.create_loss(loss_type="smplify", joint_weights=joint_weights, rho=cfg["rho"], use_joints_conf=True,
                               use_face=False, use_hands=False, body_pose_prior=mk("l2"), shape_prior=mk("l2"),
                               angle_prior=mk("angle"), interpenetration=False, dtype=torch.float32,
                               regression_pose=torch.tensor(frames["reg_pose"][i:i + 1], device=dev), num_stages=3).to(dev)
    w = {"data_weight": 1000.0 / frames["H"], "body_pose_weight": torch.tensor(cfg["body_pose_prior_weights"][1], device=dev),
         "shape_weight": torch.tensor(cfg["shape_weights"][1], device=dev)}
    w["bending_prior_weight"] = 3.17 * w["body_pose_weight"]
    loss.reset_loss_weights(w)
    out = bm(return_verts=True, body_pose=pose_embedding, return_full_pose=True)
    total = loss(out, camera=camera, gt_joints=gt_joints, body_model_faces=bm.faces_tensor, joints_conf=joints_conf,
                 joint_weights=joint_weights, pose_embedding=pose_embedding, use_vposer=False)
    assert total.dim() == 0 and torch.isfinite(total)
    with fitting.FittingMonitor(**cfg) as monitor:
        closure = monitor.create_fitting_closure(None, bm, camera=camera, gt_joints=gt_joints, joints_conf=joints_conf,
                                                 joint_weights=joint_weights, loss=loss, use_vposer=False,
                                                 pose_embedding=pose_embedding, return_verts=True, return_full_pose=True)
        via_closure = closure(stage=1, backward=False)
    assert float(total) == float(via_closure)
    P = dict(global_orient=frames["reg_global"], pose_embedding=np.tile(pose_embedding.detach().cpu().numpy(), (3, 1)),
             betas=np.tile(bm.betas.detach().cpu().numpy(), (3, 1)), cam_translation=np.tile(camera.translation.detach().cpu().numpy(), (3, 1)),
             est_tz=np.zeros(3, np.float32))
    for k in ("expression", "jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose"):
        P[k] = np.tile(getattr(bm, k).detach().cpu().numpy(), (3, 1))
    lo, _ = T._oracle_closure(synth_model, cfg, frames, i, P, 1)
    assert abs(float(total) - lo) <= H.CLOSURE_LOSS_TOL * abs(lo), (float(total), lo)
    # camera-initialisation loss (fitting.py:499-520)
    init_idxs = [k for k in cfg["init_joints_idxs"] if float(gt_joints[0, k, 0]) != 0]
    closs = fitting.create_loss("camera_init", joints_conf=joints_conf, use_conf=False, trans_estimation=camera.translation.detach().clone(),
                                init_joints_idxs=torch.tensor(init_idxs, device=dev), depth_loss_weight=1e2, dtype=torch.float32).to(dev)
    closs.reset_loss_weights({"data_weight": 1000.0 / frames["H"]})
    cval = closs(out, camera=camera, gt_joints=gt_joints, body_model=bm)
    assert cval.dim() == 0 and torch.isfinite(cval) and float(cval) >= 0
    with pytest.raises(RuntimeError):
        from smplifyx_amd.smplx import ModelOutput
        loss(ModelOutput(joints=out.joints), camera=camera, gt_joints=gt_joints, joints_conf=joints_conf, joint_weights=joint_weights)


def test_long_queue_through_a_small_column_pool(synth_model):
    """cfg.slots with a queue 16 waves deep (B / slots = 16): the bound on the polled rounds scales with the queue, and
    the pooled job equals the resident one frame for frame (camera stage only: cheap)."""
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)
    cfg["use_camera_prior"] = False
    cfg["maxiters"] = 4
    from smplifyx_amd import engine
    dm = T._dm(synth_model, cfg)
    B = 512
    frames = T.synth_frames(synth_model, cfg, 3)
    idx = [i % 3 for i in range(B)]
    res = {}
    for slots in (0, 32):
        fb = H.engine_batch_from_frames(dm, cfg, frames, idx, lbs_mode="dense")
        fb.close()
        kp = frames["keypoints"][idx]
        fb = engine.FrameBatch(dm, B, cfg, lbs_mode="dense", reuse_entry_eval=True, has_regression_pose=True, slots=slots)
        K = kp.shape[1]
        jw = np.tile(H.base_joint_weights(cfg, K), (B, 1))
        cm = np.zeros((B, K), np.float32); cm[:, cfg["init_joints_idxs"]] = 1
        fb.set_frames(kp, jw, cm, frames["focal"], np.tile([frames["W"] * 0.5, frames["H"] * 0.5], (B, 1)), 1000.0 / frames["H"])
        fb.set_params(regression_pose=frames["reg_pose"][idx], global_orient=frames["reg_global"][idx],
                      pose_embedding=frames["reg_pose"][idx], cam_translation=np.zeros((B, 3), np.float32))
        fb.guess_init(cfg["body_tri_idxs"])
        fb.fit(first_stage=-1, last_stage=0)
        res[slots] = fb.stats()["stage_loss"][:, :2].copy()
        fb.close()
    assert np.all(np.isfinite(res[32])) and np.array_equal(res[0], res[32])
