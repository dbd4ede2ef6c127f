"""GPU parity tests (run with `-m gpu` on the MI355X box): the HIP path, called through the
C ABI (ctypes), against the CPU oracle on identical seeded inputs.

Tolerances (fp32 path; north_star asks 1e-4 relative):
  LBS vertices / joints     abs 5e-6 m on ~1 m geometry (vs oracle fp32), 1e-5 vs oracle fp64
  closure loss              rel 1e-5  (helpers.CLOSURE_LOSS_TOL; observed maxima are printed at session end)
  closure gradient          ||g - g_ref|| / ||g_ref|| < 1e-4 (helpers.CLOSURE_GRAD_TOL; vs oracle fp64 autograd)
  end-to-end stage losses   rel 2e-3 on well-posed synthetic frames (the reference's own
                            fp32-vs-fp64 spread on such frames is 6e-5 .. 1e-3)
"""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path has no CPU fallback")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def cfg_full():
    return H.load_cfg("fit_smplx_combined_coco25.yaml")


@pytest.fixture(scope="module")
def cfg_body():
    return H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)


def _dm(model, cfg, **kw):
    from smplifyx_amd import engine
    return engine.DeviceModel(model, joint_map=H.joint_map_for(cfg), num_betas=cfg["num_betas"],
                              num_expression_coeffs=cfg["num_expression_coeffs"],
                              num_pca_comps=cfg["num_pca_comps"], use_face_contour=cfg["use_face_contour"], **kw)


def _oracle_forward(model, cfg, P, dtype):
    bm = H.oracle_model(model, cfg, dtype)
    B = P["global_orient"].shape[0]
    V, J, F = [], [], []
    for i in range(B):
        bm.reset_params(**{k: v[i:i + 1] for k, v in P.items() if k != "pose_embedding"})
        with torch.no_grad():
            o = bm(return_verts=True, body_pose=torch.tensor(P["pose_embedding"][i:i + 1], dtype=dtype),
                   return_full_pose=True)
        V.append(o.vertices[0].numpy()); J.append(o.joints[0].numpy()); F.append(o.full_pose[0].numpy())
    return np.stack(V), np.stack(J), np.stack(F)


def test_lbs_forward_dense_matches_oracle(gpu, synth_model, cfg_full):
    dm = _dm(synth_model, cfg_full)
    rng = np.random.RandomState(3)
    B = 37      # not a multiple of the 32-frame MFMA tile
    P = H.random_params(rng, B)
    P["global_orient"][0] = 0; P["pose_embedding"][0] = 0          # exactly-zero pose (eps path)
    t = lambda a: torch.tensor(a, device=gpu)
    verts, joints, fp = dm.lbs_forward(t(P["global_orient"]), t(P["pose_embedding"]), t(P["betas"]),
                                       t(P["expression"]), t(P["jaw_pose"]), t(P["leye_pose"]), t(P["reye_pose"]),
                                       t(P["left_hand_pose"]), t(P["right_hand_pose"]))
    v32, j32, f32 = _oracle_forward(synth_model, cfg_full, P, torch.float32)
    v64, j64, f64 = _oracle_forward(synth_model, cfg_full, P, torch.float64)
    assert np.abs(fp.cpu().numpy() - f32).max() < 1e-6
    assert np.abs(verts.cpu().numpy() - v32).max() < 5e-6
    assert np.abs(verts.cpu().numpy() - v64).max() < 1e-5
    assert np.abs(joints.cpu().numpy() - j32).max() < 5e-6
    assert np.abs(joints.cpu().numpy() - j64).max() < 1e-5
    assert joints.shape == (B, 135, 3)


def _oracle_closure(model, cfg, frames, i, params, stage, dtype=torch.float64):
    """loss + flat gradient (reference variable order) of the oracle objective at `params`."""
    ff = H.oracle_frame_fit(model, cfg, frames, i, dtype=dtype)
    bm = ff.bm
    with torch.no_grad():
        for k, v in params.items():
            if k == "est_tz":
                continue
            if k == "pose_embedding":
                ff.pose_embedding.copy_(torch.tensor(v[i:i + 1], dtype=dtype))
            elif k == "cam_translation":
                ff.cam_t.copy_(torch.tensor(v[i:i + 1], dtype=dtype))
            else:
                getattr(bm, k).copy_(torch.tensor(v[i:i + 1], dtype=dtype))
        ff.init_t[:, 2] = float(params["est_tz"][i])
    if stage < 0:
        ps = [ff.cam_t, bm.global_orient]
        fn = ff.camera_objective
    else:
        ps = [p for p in bm.parameters() if p.requires_grad] + [ff.pose_embedding]
        w = dict(ff.stages[stage]); w["data_weight"] = ff.data_weight
        w["bending_prior_weight"] = 3.17 * w["body_pose_weight"]
        jw = ff.jw.clone()
        if ff.use_hands: jw[:, ff.nb:ff.nb + 42] = w["hand_weight"]
        if ff.use_face: jw[:, ff.nb + 42:] = w["face_weight"]
        jw[:, ff.low] = 0
        fn = lambda: ff.body_terms(stage, w, jw)["total"]
    for p in ps:
        p.grad = None
    loss = fn()
    loss.backward()
    g = torch.cat([(p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), dtype=dtype)) for p in ps])
    return loss.item(), g.numpy()


def closure_probe(model, cfg, mode, label, B=3, seed=11, stages=None, vposer=None, check=True):
    """HIP closure (through the C ABI) against fp64 autograd of the oracle at B seeded random points of B synthetic
    frames, camera stage + body stages.  Returns {stage: (max loss rel err, max gradient rel err)}; with check=True every
    comparison also goes through helpers.check_closure (bounds 1e-5 / 1e-4).  Used by the tests below,
    __graft_entry__.smoke() and bench.py's closure_parity object."""
    dm = _dm(model, cfg, **({"vposer": vposer} if vposer is not None else {}))
    frames = synth_frames(model, cfg, B)
    fb = H.engine_batch_from_frames(dm, cfg, frames, range(B), lbs_mode=mode)
    rng = np.random.RandomState(seed)
    P = H.random_params(rng, B, scale=0.5)
    if vposer is not None:
        P["pose_embedding"] = rng.normal(size=(B, fb.nemb)).astype(np.float32)
    else:
        P["pose_embedding"] = frames["reg_pose"] + 0.1 * rng.normal(size=(B, 63)).astype(np.float32)
    P["global_orient"] = frames["reg_global"] + 0.1 * rng.normal(size=(B, 3)).astype(np.float32)
    P["cam_translation"] = (frames["cam_t"] + 0.3 * rng.normal(size=(B, 3))).astype(np.float32)
    est = (frames["cam_t"][:, 2] + 1.0).astype(np.float32)
    fb.set_frames(frames["keypoints"], _jw(cfg, frames), _cmask(cfg, frames), frames["focal"],
                  np.tile([frames["W"] * 0.5, frames["H"] * 0.5], (B, 1)), 1000.0 / frames["H"], est_tz=est)
    if vposer is not None:
        fb.set_params(**P)
    else:
        fb.set_params(regression_pose=frames["reg_pose"], **P)
    P["est_tz"] = est
    out = {}
    for stage in ([-1] + list(range(fb.n_stages)) if stages is None else stages):
        loss, grad = fb.closure(stage)
        le_max = ge_max = 0.0
        for i in range(B):
            lo, go = _oracle_closure(model, cfg, frames, i, P, stage)
            le, ge = H.check_closure(label, stage, loss[i], lo, grad[i], go) if check else H.closure_errors(loss[i], lo, grad[i], go)
            le_max, ge_max = max(le_max, le), max(ge_max, ge)
            # the dead body_pose parameter receives no gradient (fit_single_frame.py:554-559)
            if stage >= 0 and not cfg["use_vposer"]:
                assert np.all(grad[i][13:13 + 63] == 0)
        out[stage] = (le_max, ge_max)
    fb.close(); dm.close()
    return out


@pytest.mark.parametrize("which,mode", [("body", "rows"), ("body", "dense"), ("full", "rows"), ("full", "dense")])
def test_closure_matches_oracle(gpu, synth_model, cfg_body, cfg_full, which, mode):
    cfg = cfg_body if which == "body" else cfg_full
    closure_probe(synth_model, cfg, mode, "%s-%s" % (which, mode))


def _conf_thr(cfg, K):
    """fit_single_frame.py:276-287: the confidence threshold applies to the body keypoints of the format, 0 to the others."""
    from smplifyx_amd import engine
    nb = min(K, engine.NUM_BODY_JOINTS[cfg.get("format", "coco25")])
    return np.array([cfg.get("confidence_threshold", 0) or 0] * nb + [0] * (K - nb))


def _jw(cfg, frames):
    kp = frames["keypoints"]; B, K = kp.shape[:2]
    thr = _conf_thr(cfg, K)
    jw = np.tile(H.base_joint_weights(cfg, K), (B, 1))
    jw[kp[:, :, 2] < thr[None]] = 0
    return jw


def _cmask(cfg, frames):
    kp = frames["keypoints"]; B, K = kp.shape[:2]
    thr = _conf_thr(cfg, K)
    low = kp[:, :, 2] < thr[None]
    m = np.zeros((B, K), np.float32)
    for b in range(B):
        for j in cfg["init_joints_idxs"]:
            if kp[b, j, 0] != 0 and kp[b, j, 1] != 0 and not low[b, j]:
                m[b, j] = 1
    return m


_FRAME_CACHE = {}


def synth_frames(model, cfg, n):
    from smplifyx_amd import synthetic
    key = (cfg["use_hands"], cfg["use_face"], cfg.get("format", "coco25"), bool(cfg.get("use_face_contour")), n)
    if key not in _FRAME_CACHE:
        K = len(H.joint_map_for(cfg))
        _FRAME_CACHE[key] = synthetic.make_frames(n, H.oracle_joints_fn(model, cfg), K, focal=5000.0)
    return _FRAME_CACHE[key]


def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))


@pytest.mark.parametrize("mode,reuse", [("rows", False), ("rows", True), ("dense", True)])
def test_fit_matches_reference_on_wellposed_frames(gpu, synth_model, cfg_body, mode, reuse):
    """Whole schedule (camera stage + 3 body stages) for 2 synthetic frames against the REAL
    reference's fit_single_frame (tests/golden/e2e_synth.npz, generated by tools/make_goldens.py).
    The optimisation is chaotic beyond the camera stage: the golden file also holds the
    reference's fp64 run, and |ref32 - ref64| is the yardstick for what 'equal' can mean."""
    g = _golden("e2e_synth")
    cfg = dict(cfg_body); cfg["use_camera_prior"] = False
    dm = _dm(synth_model, cfg)
    B = 2
    frames = dict(keypoints=g["keypoints"], reg_pose=g["reg_pose"], reg_global=g["reg_global"], H=600, W=800, focal=5000.0)
    fb = H.engine_batch_from_frames(dm, cfg, frames, range(B), lbs_mode=mode, reuse=reuse)
    fb.guess_init(cfg["body_tri_idxs"])
    fb.fit()
    st = fb.stats()
    got = fb.get_params()
    for i in range(B):
        ref32, ref64 = g["f%d_f32_losses" % i], g["f%d_f64_losses" % i]
        spread = np.abs(ref32 - ref64) / np.abs(ref64)
        rel = np.abs(st["stage_loss"][i] - ref32) / np.abs(ref32)
        assert rel[0] < 1e-4, (i, st["stage_loss"][i], ref32)             # camera stage: well conditioned
        # first body stage: the reference's own fp32-vs-fp64 difference is ONE sample of how far two
        # correctly rounded evaluations of this trajectory drift apart; allow twice that sample
        assert rel[1] < max(2 * spread[1], 3e-3), (i, rel, spread)
        assert np.all(rel[2:] < np.maximum(3 * spread[2:], 5e-2)), (i, rel, spread)   # chaotic tail
        dpose = np.abs(got["pose_embedding"][i] - g["f%d_f32_body_pose" % i][0]).max()
        spose = np.abs(g["f%d_f64_body_pose" % i] - g["f%d_f32_body_pose" % i]).max()
        assert dpose < max(2 * spose, 2e-2), (dpose, spose)
        if not reuse:
            ev = g["f%d_f32_evals" % i]
            assert abs(int(st["stage_evals"][i].sum()) - int(ev.sum())) <= 0.5 * ev.sum()
            assert np.array_equal(st["stage_evals"][i], st["stage_ref_evals"][i])


def test_demo_config1_matches_reference(gpu, synth_model, cfg_body):
    """BASELINE config 1: the two demo/ frames (real blended keypoints + ExPose/PIXIE priors,
    camera prior, body-only, cfg_files/fit_smplx_combined_coco25.yaml) on the synthetic model,
    against the reference's own fit (tests/golden/demo_config1.npz, fp32).  The problem is
    ill-posed (real keypoints vs synthetic geometry): SURVEY.md 0 measured +-5 % on the final
    loss for the reference against itself; the camera stage is well conditioned."""
    g = _golden("demo_config1"); e = _golden("euler")
    from smplifyx_amd import engine
    cfg = dict(cfg_body)
    dm = _dm(synth_model, cfg)
    for name in ("02_cropped", "18_cropped"):
        kp = g[name + "_keypoints"]; Hh, Ww = [int(v) for v in g[name + "_HW"]]
        focal = float(g[name + "_focal"])
        frames = dict(keypoints=kp, reg_pose=e[name + "_combined_pose"][None], reg_global=e[name + "_global"][None],
                      H=Hh, W=Ww, focal=focal)
        fb = H.engine_batch_from_frames(dm, cfg, frames, [0], lbs_mode="dense", reuse=False)
        est = np.array([g[name + "_cam_prior_t"][2]], np.float32)
        fb.set_frames(kp, _jw(cfg, frames), _cmask(cfg, frames), focal, g[name + "_cam_prior_center"][None].astype(np.float32),
                      1000.0 / Hh, est_tz=est)
        fb.set_params(regression_pose=frames["reg_pose"], global_orient=frames["reg_global"],
                      pose_embedding=frames["reg_pose"], cam_translation=g[name + "_cam_prior_t"][None].astype(np.float32))
        fb.fit()
        st = fb.stats()
        ref = g[name + "_losses"]
        rel = np.abs(st["stage_loss"][0] - ref) / np.abs(ref)
        assert rel[0] < 1e-4, (name, st["stage_loss"][0], ref)
        assert np.all(rel[1:] < 5e-2), (name, st["stage_loss"][0], ref)


@pytest.mark.parametrize("name,yaml_", [("coco25", "fit_smplx_combined_coco25.yaml"), ("halpe", "fit_smplx_combined_halpe.yaml")])
@pytest.mark.parametrize("mode", ["rows", "dense"])
def test_full_model_fit_matches_reference(gpu, synth_model, name, yaml_, mode):
    """Hands + face + contour (K = 135 / 136 keypoints, every prior term of SMPLifyLoss active, hand / face
    joint weights of the schedule) with a regression prior, coco25 and halpe formats, against the REAL
    reference's fit_single_frame (tests/golden/e2e_full.npz, fp32 and fp64 runs): camera stage 1e-4, body
    stages within the reference's own fp32 / fp64 spread as everywhere else (LAB_NOTES.md §3)."""
    from smplifyx_amd import driver
    g = _golden("e2e_full")
    cfg = H.load_cfg(yaml_, interpenetration=False)
    cfg["use_camera_prior"] = False
    dm = _dm(synth_model, cfg)
    kp = g[name + "_keypoints"]
    K = kp.shape[1]
    assert K == len(H.joint_map_for(cfg))
    res = driver.fit_frames(dm, cfg, kp, H.base_joint_weights(cfg, K), 600, 800, 5000.0, reg_pose=g[name + "_reg_pose"],
                            reg_global=g[name + "_reg_global"], lbs_mode=mode)
    ref32, ref64 = g[name + "_f32_losses"], g[name + "_f64_losses"]
    spread = np.abs(ref32 - ref64) / np.abs(ref64)
    rel = np.abs(res["stage_loss"][0] - ref32) / np.abs(ref32)
    assert rel[0] < 1e-4, (rel, res["stage_loss"][0], ref32)
    # (one frame: the 16-frame set of the same configuration, test_full_model_set_matches_reference, allows a MEAN |difference|
    #  of 6e-3 after this stage; a single trajectory is held to 1e-2)
    assert rel[1] < max(2 * spread[1], 1e-2), (rel, spread)
    assert np.all(rel[2:] < np.maximum(3 * spread[2:], 5e-2)), (rel, spread)
    assert res["left_hand_pose"].shape == (1, 12) and res["expression"].shape == (1, 10) and np.all(np.isfinite(res["jaw_pose"]))


@pytest.mark.parametrize("mode", ["rows", "dense"])
def test_benchmark_configuration_matches_reference(gpu, synth_model, mode):
    """bench.py's own configuration (fit_smplx_smplifyx.yaml weights, 5 body stages, body-only, regression
    prior) on frames 0-63 of its sequence against the REAL reference (tests/golden/e2e_bench.npz: 64 fits in
    fp32 and 64 in fp64).  The optimisation is chaotic, so single frames say nothing: the DISTRIBUTION of the
    differences to the reference's fp32 fits has to sit inside what the reference's own fp64 fits show over the
    same frames (fp64 ends 1.95 % lower on average, mean |difference| 2.6 %, median 2.2 %).
      camera stage (well conditioned): every frame within 2e-4;
      every body stage: signed mean difference within +-1 x the reference's own mean |fp64 - fp32| of that stage
        (floor 3e-3), mean |difference| within 1.5 x it;
      final loss: signed mean within +-1 x the reference's mean |fp64 - fp32|, median and 90th percentile of
        |difference| within 1.5 x the reference's own;
      work: mean closure evaluations between 0.8 x the reference's fp32 count and 1.1 x its fp64 count."""
    import bench as BB
    from smplifyx_amd import driver
    g = _golden("e2e_bench")
    cfg = BB.build_cfg("body")
    dm = _dm(synth_model, cfg)
    n = g["keypoints"].shape[0]
    assert n >= 32
    res = driver.fit_frames(dm, cfg, g["keypoints"], H.base_joint_weights(cfg, 25), 600, 800, 5000.0, reg_pose=g["reg_pose"],
                            reg_global=g["reg_global"], lbs_mode=mode)
    ours = res["stage_loss"].astype(np.float64)
    r32 = np.stack([g["f%d_f32_losses" % i] for i in range(n)])
    r64 = np.stack([g["f%d_f64_losses" % i] for i in range(n)])
    # frames on which the reference disagrees with itself by more than 25 % after some stage (frame 51: two basins) are not
    # averaged: which basin a run ends in is not a property of the arithmetic under test
    ok = BB.self_consistent_frames(r32, r64)
    assert ok.sum() >= n - 2, np.flatnonzero(~ok)
    assert np.abs((ours[:, 0] - r32[:, 0]) / r32[:, 0]).max() < 2e-4       # (the reference's fp32 vs fp64: 6e-5 at most)
    ours, r32, r64 = ours[ok], r32[ok], r64[ok]
    d = (ours - r32) / np.abs(r32)
    y = (r64 - r32) / np.abs(r32)
    for k in range(1, ours.shape[1]):
        yard = max(np.abs(y[:, k]).mean(), 3e-3)
        assert abs(d[:, k].mean()) <= yard, (k, d[:, k].mean(), yard)
        assert np.abs(d[:, k]).mean() <= 1.5 * yard, (k, np.abs(d[:, k]).mean(), yard)
    st = BB.parity_stats(ours[:, -1], r32[:, -1], r64[:, -1])
    assert abs(st["final_loss_rel_delta_signed_mean"]) <= st["reference_f32_vs_f64_rel_delta_mean"], st
    assert st["final_loss_rel_delta_median"] <= 1.5 * st["reference_f32_vs_f64_rel_delta_median"], st
    assert st["final_loss_rel_delta_p90"] <= 1.5 * st["reference_f32_vs_f64_rel_delta_p90"], st
    ev = res["stage_ref_evals"][ok].sum(1).mean()
    e32 = np.mean([g["f%d_f32_evals" % i].sum() for i in range(n)]); e64 = np.mean([g["f%d_f64_evals" % i].sum() for i in range(n)])
    assert 0.8 * e32 <= ev <= 1.1 * e64, (ev, e32, e64)


@pytest.mark.parametrize("mode", ["rows", "dense"])
def test_full_model_set_matches_reference(gpu, synth_model, mode):
    """cfg_files/fit_smplx_combined_halpe.yaml (hands + face + contour, K = 136, every prior term, joint confidences; the
    interpenetration term off: the reference's package is absent) on 16 frames against the REAL reference's fits
    (tests/golden/e2e_full_set.npz, fp32 and fp64).  The reference against itself: camera stage 1e-6, first two body
    stages 0.2 % / 0.06 % (mean |fp64 - fp32|), last stage (face keypoints at weight 2 on the dynamic contour's lookup
    table: non-smooth) 11 % on average and up to 31 %.  Required: camera stage 2e-4 per frame; body stages 1-2 signed mean
    within +- max(the reference's own mean |difference|, 3e-3) and mean |difference| within twice that; last stage signed
    mean within +- the reference's own mean |difference| and median |difference| within 1.5 x the reference's own."""
    from smplifyx_amd import driver
    g = _golden("e2e_full_set")
    cfg = H.load_cfg("fit_smplx_combined_halpe.yaml", interpenetration=False)
    cfg["use_camera_prior"] = False
    dm = _dm(synth_model, cfg)
    kp = g["keypoints"]
    n, K = kp.shape[:2]
    assert K == len(H.joint_map_for(cfg)) == 136 and n >= 16
    res = driver.fit_frames(dm, cfg, kp, H.base_joint_weights(cfg, K), 600, 800, 5000.0, reg_pose=g["reg_pose"],
                            reg_global=g["reg_global"], lbs_mode=mode)
    ours = res["stage_loss"].astype(np.float64)
    r32 = np.stack([g["f%d_f32_losses" % i] for i in range(n)]); r64 = np.stack([g["f%d_f64_losses" % i] for i in range(n)])
    d = (ours - r32) / np.abs(r32); y = (r64 - r32) / np.abs(r32)
    assert np.abs(d[:, 0]).max() < 2e-4, d[:, 0]
    for k in (1, 2):
        yard = max(np.abs(y[:, k]).mean(), 3e-3)
        assert abs(d[:, k].mean()) <= yard and np.abs(d[:, k]).mean() <= 2 * yard, (k, d[:, k].mean(), np.abs(d[:, k]).mean(), yard)
    assert abs(d[:, 3].mean()) <= np.abs(y[:, 3]).mean(), (d[:, 3].mean(), np.abs(y[:, 3]).mean())
    assert np.median(np.abs(d[:, 3])) <= 1.5 * np.median(np.abs(y[:, 3])), (np.median(np.abs(d[:, 3])), np.median(np.abs(y[:, 3])))
    assert np.all(np.isfinite(res["left_hand_pose"])) and np.all(np.isfinite(res["expression"]))


@pytest.mark.parametrize("mode", ["rows", "dense"])
def test_vposer_set_matches_reference(gpu, synth_model, mode):
    """BASELINE config 3 (full SMPL-X K = 135, VPoser decode in the loop, the 5 stages of fit_smplx_smplifyx.yaml, zero
    latent start) on 16 frames against the REAL reference's fits (tests/golden/e2e_vposer_set.npz, fp32 and fp64).  The
    reference against itself (mean |fp64 - fp32|): 0.8 % / 1.2 % / 1.2 % / 3.9 % after body stages 1-4 and 24 % (up to
    51 %) after the last one (face keypoints on the dynamic contour's lookup table: non-smooth).  Required: camera stage
    2e-4 per frame; stages 1-3 signed mean within +- max(the reference's own mean |difference|, 5e-3) and mean |difference|
    within twice that; stage 4 median within that yard, median |difference| within twice the reference's, mean |difference|
    within three yards, and the bounds of stages 1-3 on the frames left without the two largest differences; last stage signed mean within +- the reference's own mean |difference| and median within 1.5 x."""
    from smplifyx_amd import synthetic
    g = _golden("e2e_vposer_set")
    cfg = H.load_cfg("fit_smplx_smplifyx.yaml")
    dm = _dm(synth_model, cfg, vposer=synthetic.make_synthetic_vposer(0))
    n = g["keypoints"].shape[0]
    frames = dict(keypoints=g["keypoints"], H=600, W=800, focal=5000.0)
    fb = H.engine_batch_from_frames(dm, cfg, frames, range(n), lbs_mode=mode, reuse=True)
    fb.guess_init(cfg["body_tri_idxs"])
    fb.fit()
    ours = fb.stats()["stage_loss"].astype(np.float64)
    r32 = np.stack([g["f%d_f32_losses" % i] for i in range(n)]); r64 = np.stack([g["f%d_f64_losses" % i] for i in range(n)])
    d = (ours - r32) / np.abs(r32); y = (r64 - r32) / np.abs(r32)
    assert np.abs(d[:, 0]).max() < 2e-4, d[:, 0]
    for k in (1, 2, 3):
        yard = max(np.abs(y[:, k]).mean(), 5e-3)
        assert abs(d[:, k].mean()) <= yard and np.abs(d[:, k]).mean() <= 2 * yard, (k, d[:, k].mean(), np.abs(d[:, k]).mean(), yard)
    # body stage 4: single frames land in another basin from one build to the next (profiles/r04_vposer_set_probe.txt: four
    # builds of this library that differ in summation order only -- per-frame differences of -16 % ... +55 %, medians
    # -0.3 % ... +0.07 %; the reference's own fp64 fits differ from its fp32 fits by -12 % ... +14 % per frame).  The mean of 16
    # such draws is decided by its largest one, so the centre is tested on medians and the mean only as a gross bound.
    yard = max(np.abs(y[:, 4]).mean(), 5e-3)
    assert abs(np.median(d[:, 4])) <= yard, (np.median(d[:, 4]), yard)
    assert np.median(np.abs(d[:, 4])) <= 2 * max(np.median(np.abs(y[:, 4])), 5e-3), (np.median(np.abs(d[:, 4])), np.median(np.abs(y[:, 4])))
    assert np.abs(d[:, 4]).mean() <= 3 * yard, (np.abs(d[:, 4]).mean(), yard)
    # ... and so that the medians cannot hide a regression across the set: the bounds stages 1-3 are held to --
    # signed mean within the yard, mean |difference| within twice it -- on the 14 frames left when the two largest draws are set
    # aside (the four builds of profiles/r04_vposer_set_probe.txt: 0.006 / 0.026, -0.004 / 0.017, 0.000 / 0.013, -0.002 / 0.011
    # against 0.039 / 0.079; a shift of every frame by the reference's own fp32 / fp64 spread fails it)
    keep = np.argsort(np.abs(d[:, 4]))[:-2]
    assert abs(d[keep, 4].mean()) <= yard and np.abs(d[keep, 4]).mean() <= 2 * yard, (d[keep, 4].mean(), np.abs(d[keep, 4]).mean(), yard)
    assert abs(d[:, 5].mean()) <= np.abs(y[:, 5]).mean(), (d[:, 5].mean(), np.abs(y[:, 5]).mean())
    assert np.median(np.abs(d[:, 5])) <= 1.5 * np.median(np.abs(y[:, 5])), (np.median(np.abs(d[:, 5])), np.median(np.abs(y[:, 5])))


def test_continuous_batching_matches_resident_batch(gpu, synth_model):
    """Dense mode with a column pool smaller than the job (cfg.slots: frames queue and take over the columns of
    frames that finish) gives every frame the result it has when all frames are resident -- bit for bit: frames are
    independent and a GEMM column's arithmetic does not depend on its index or on the other columns."""
    import bench as BB
    from smplifyx_amd import driver
    g = _golden("e2e_bench")
    cfg = BB.build_cfg("body")
    dm = _dm(synth_model, cfg)
    n = 80
    kp = np.concatenate([g["keypoints"]] * 3)[:n]; rp = np.concatenate([g["reg_pose"]] * 3)[:n]; rg = np.concatenate([g["reg_global"]] * 3)[:n]
    kp = kp.copy(); kp[32:, :, :2] += 0.37          # (not exact repeats)
    a = driver.fit_frames(dm, cfg, kp, H.base_joint_weights(cfg, 25), 600, 800, 5000.0, reg_pose=rp, reg_global=rg, lbs_mode="dense")
    b = driver.fit_frames(dm, cfg, kp, H.base_joint_weights(cfg, 25), 600, 800, 5000.0, reg_pose=rp, reg_global=rg, lbs_mode="dense",
                          slots=32)
    for k in ("stage_loss", "stage_evals", "pose_embedding", "betas", "global_orient", "cam_translation"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k        # (a frame that ends non-finite does so in both)


def test_pool_queue_ordered_by_predicted_cost_returns_the_callers_order(gpu, synth_model):
    """driver.fit_frames through a column pool smaller than the job: the queue is ordered by what is known about a frame
    before the fit (side views are fitted twice, frames with fewer than 3 of the 4 camera keypoints start from an
    under-determined camera: driver.predicted_cost), longest first -- and every result comes back in the caller's frame
    order, bit for bit what the resident run (one column per frame, frame order) produces."""
    import bench as BB
    from smplifyx_amd import driver, synthetic
    import torch
    cfg = BB.build_cfg("body")
    dm = _dm(synth_model, cfg)
    B = 160
    dev = torch.device("cuda")

    def joints_fn(P):
        z = lambda n: torch.zeros([len(P["betas"]), n], device=dev)
        t = lambda a: torch.tensor(a, device=dev)
        _, j, _ = dm.lbs_forward(t(P["global_orient"]), t(P["body_pose"]), t(P["betas"]), z(10), z(3), z(3), z(3), z(12), z(12),
                                 return_verts=False, return_full_pose=False)
        return j.cpu().numpy()
    fr = synthetic.make_frames(B, joints_fn, 25, start=0, focal=5000.0)
    jw = H.base_joint_weights(cfg, 25)
    cost = driver.predicted_cost(cfg, driver.prepare_frames(cfg, fr["keypoints"], jw))
    assert 0 < (cost > 1).sum() < B                      # some frames are predicted long (the generator drops 10 % of the keypoints)
    kw = dict(reg_pose=fr["reg_pose"], reg_global=fr["reg_global"], lbs_mode="dense", reuse_entry_eval=True)
    ra = driver.fit_frames(dm, cfg, fr["keypoints"], jw, fr["H"], fr["W"], fr["focal"], slots=0, **kw)
    rb = driver.fit_frames(dm, cfg, fr["keypoints"], jw, fr["H"], fr["W"], fr["focal"], slots=48, order="auto", **kw)
    rc = driver.fit_frames(dm, cfg, fr["keypoints"], jw, fr["H"], fr["W"], fr["focal"], slots=48, order="given", **kw)
    assert set(ra) == set(rb) == set(rc)
    for k in ra:
        assert np.array_equal(np.asarray(ra[k]), np.asarray(rb[k]), equal_nan=True), k
        assert np.array_equal(np.asarray(ra[k]), np.asarray(rc[k]), equal_nan=True), k
    # the predictor is worth ordering by: the predicted-long frames take more evaluations than the others
    ev = ra["stage_evals"].sum(1)
    assert ev[cost > 1].mean() > 1.2 * ev[cost == 1].mean(), (ev[cost > 1].mean(), ev[cost == 1].mean())
    assert driver.auto_slots(256) == 0 and driver.auto_slots(512) == 0 and driver.auto_slots(1024) == 512


def test_one_gpu_share_of_the_8192_frame_job(gpu, synth_model):
    """BASELINE configs[3] as one rank sees it: 1 024 frames (bench.py's generator) -- resident, and through a pool
    of 256 GEMM columns (`bench.py --frames 1024 --slots 256`).  Camera stage + first body stage: every frame finishes, no
    frame depends on the pool (bitwise), and the shard's records survive pack / unpack with exact frame indices."""
    import bench as BB
    from smplifyx_amd import driver, dist as sd
    import torch
    cfg = BB.build_cfg("body")
    dm = _dm(synth_model, cfg)
    B = 1024
    dev = torch.device("cuda")

    def joints_fn(P):
        z = lambda n: torch.zeros([len(P["betas"]), n], device=dev)
        t = lambda a: torch.tensor(a, device=dev)
        _, j, _ = dm.lbs_forward(t(P["global_orient"]), t(P["body_pose"]), t(P["betas"]), z(10), z(3), z(3), z(3), z(12), z(12),
                                 return_verts=False, return_full_pose=False)
        return j.cpu().numpy()
    from smplifyx_amd import synthetic
    fr = synthetic.make_frames(B, joints_fn, 25, start=3 * B, focal=5000.0)           # rank 3's frames
    jw = H.base_joint_weights(cfg, 25)
    got = []
    for slots in (0, 256):
        fb, prep = driver._make_batch(dm, cfg, fr["keypoints"], jw, fr["H"], fr["W"], fr["focal"], fr["reg_pose"], fr["reg_global"],
                                      None, None, "dense", True, slots=slots)
        fb.fit(first_stage=-1, last_stage=0)
        st, P = fb.stats(), fb.get_params()
        got.append((st, P))
        fb.close()
    (sa, pa), (sb, pb) = got
    assert np.all(sa["stage_evals"][:, :2] > 0) and np.isfinite(sa["stage_loss"][:, :2]).mean() > 0.99
    for k in ("stage_loss", "stage_evals", "stage_ref_evals"):
        assert np.array_equal(sa[k][:, :2], sb[k][:, :2], equal_nan=True), k
    for k in ("pose_embedding", "betas", "global_orient", "cam_translation"):
        assert np.array_equal(pa[k], pb[k], equal_nan=True), k
    res = dict(pa, stage_evals=sa["stage_evals"], final_loss=sa["stage_loss"][:, 1])
    rec = sd.pack_records(res, 3 * B)
    u = sd.unpack_records(rec, sd.record_fields(res))
    assert np.array_equal(u["frame"][:, 0], 3 * B + np.arange(B)) and np.array_equal(u["betas"], pa["betas"])


def test_shard_invariance_bitwise(gpu, synth_model, cfg_body):
    """A frame's result does not depend on which other frames share its batch."""
    cfg = dict(cfg_body); cfg["use_camera_prior"] = False
    dm = _dm(synth_model, cfg)
    frames = synth_frames(synth_model, cfg, 3)
    res = []
    for idx in ([0, 1, 2], [2], [1, 2]):
        fb = H.engine_batch_from_frames(dm, cfg, frames, idx, lbs_mode="rows", reuse=True)
        fb.guess_init(cfg["body_tri_idxs"])
        fb.fit()
        res.append((idx, fb.get_params(), fb.stats()))
    p0 = res[0][1]
    for idx, p, st in res[1:]:
        for k in p:
            assert np.array_equal(p[k][-1], p0[k][2]), k


def test_side_view_second_orientation_on_device(gpu, synth_model, cfg_body):
    """Frames whose 2-D shoulders coincide are fitted twice (flipped orientation) inside the
    batch; the result equals running the two passes by hand (fit_single_frame.py:527-551,662-667)."""
    from smplifyx_amd import driver, engine
    g = _golden("e2e_synth")
    cfg = dict(cfg_body); cfg["use_camera_prior"] = False
    dm = _dm(synth_model, cfg)
    kp = g["keypoints"].copy()
    kp[1, 5, :2] = kp[1, 2, :2] + 3.0            # frame 1: shoulders 3 px apart -> side view
    jw = H.base_joint_weights(cfg, 25)
    res = driver.fit_frames(dm, cfg, kp, jw, 600, 800, 5000.0, reg_pose=g["reg_pose"], reg_global=g["reg_global"],
                            lbs_mode="rows", reuse_entry_eval=True)
    assert list(res["n_orient"]) == [1, 2]
    # by hand: pass 0 on a batch without the side-view program, then pass 1 from the flipped orientation
    frames = dict(keypoints=kp, reg_pose=g["reg_pose"], reg_global=g["reg_global"], H=600, W=800, focal=5000.0)
    fb = H.engine_batch_from_frames(dm, cfg, frames, [1], lbs_mode="rows", reuse=True)
    fb.guess_init(cfg["body_tri_idxs"])
    fb.fit(first_stage=-1, last_stage=-1)
    go_cam = fb.get_params()["global_orient"].copy()
    fb.fit(first_stage=0, last_stage=fb.n_stages - 1)
    p0, s0 = fb.get_params(), fb.stats()
    fb2 = H.engine_batch_from_frames(dm, cfg, frames, [1], lbs_mode="rows", reuse=True)
    flip = driver.flipped_orientation(go_cam[0]).astype(np.float32)[None]
    fb2.set_params(regression_pose=g["reg_pose"][1:2], global_orient=flip, pose_embedding=p0["pose_embedding"],
                   cam_translation=p0["cam_translation"])
    fb2.fit(first_stage=0, last_stage=fb2.n_stages - 1)
    p1, s1 = fb2.get_params(), fb2.stats()
    l0, l1 = s0["stage_loss"][0, -1], s1["stage_loss"][0, -1]
    want = p0 if l0 < l1 else p1
    assert np.allclose(res["final_loss"][1], min(l0, l1), rtol=1e-6)
    for k in ("global_orient", "pose_embedding", "betas", "cam_translation"):
        assert np.allclose(res[k][1], want[k][0], rtol=0, atol=1e-6), k
    assert res["stage_evals"][1, 1:].sum() == s0["stage_evals"][0, 1:].sum() + s1["stage_evals"][0, 1:].sum()


@pytest.mark.parametrize("mode", ["rows", "dense"])
def test_side_view_fit_matches_reference(gpu, synth_model, cfg_body, mode):
    """The same side-view frame through the REAL reference (tests/golden/e2e_side.npz: both orientation
    passes, fp32 and fp64): camera stage 1e-4; the first body stage of the first pass and the kept final
    loss (minimum over the two passes) within the reference's own fp32 / fp64 spread.  Which pass wins is
    itself chaotic here (pass 1 in the reference's fp32 run, pass 2 in its fp64 run: 10040.4 vs 10041.6 /
    10039.8 vs 10023.8), so the kept parameters are not compared."""
    from smplifyx_amd import driver
    g = _golden("e2e_side")
    cfg = dict(cfg_body); cfg["use_camera_prior"] = False
    dm = _dm(synth_model, cfg)
    res = driver.fit_frames(dm, cfg, g["keypoints"], H.base_joint_weights(cfg, 25), 600, 800, 5000.0, reg_pose=g["reg_pose"],
                            reg_global=g["reg_global"], lbs_mode=mode)
    assert list(res["n_orient"]) == [2]
    r32, r64 = g["f32_losses"], g["f64_losses"]
    spread = np.abs(r32 - r64) / np.abs(r64)
    sl = res["stage_loss"][0]
    assert abs(sl[0] - r32[0]) / r32[0] < 1e-4, (sl, r32)
    tol = max(3 * spread[1:].max(), 5e-2)
    kept_ref = min(r32[3], r32[6])
    assert abs(res["final_loss"][0] - kept_ref) / kept_ref < tol, (res["final_loss"], r32, tol)
    assert min(abs(sl[1] - r32[1]) / r32[1], abs(sl[1] - r32[4]) / r32[4]) < max(2 * spread[[1, 4]].max(), 3e-3), (sl, r32)
    ev = res["stage_evals"][0, 1:].sum()
    assert 0.5 * g["f32_evals"][1:].sum() <= ev <= 1.5 * max(g["f32_evals"][1:].sum(), g["f64_evals"][1:].sum()), ev


def test_closure_with_vposer_matches_oracle(gpu, synth_model):
    """BASELINE config 3: full SMPL-X (hands + face + contour, K=135), VPoser decode in the loop
    (cfg_files/fit_smplx_smplifyx.yaml: use_vposer True, 5 stages).  Loss + gradient wrt the
    88-long variable vector (32-D latent last) against oracle autograd (fp64), rows and dense."""
    from smplifyx_amd import synthetic
    cfg = H.load_cfg("fit_smplx_smplifyx.yaml")
    assert cfg["use_vposer"] and cfg["use_hands"] and cfg["use_face"]
    vpw = synthetic.make_synthetic_vposer(0)
    dm = _dm(synth_model, cfg, vposer=vpw)
    B = 2
    frames = synth_frames(synth_model, cfg, 3)
    frames = {k: (v[:B] if isinstance(v, np.ndarray) else v) for k, v in frames.items()}
    rng = np.random.RandomState(5)
    P = H.random_params(rng, B, scale=0.5)
    P["pose_embedding"] = rng.normal(size=(B, 32)).astype(np.float32)
    P["global_orient"] = frames["reg_global"] + 0.1 * rng.normal(size=(B, 3)).astype(np.float32)
    P["cam_translation"] = (frames["cam_t"] + 0.3 * rng.normal(size=(B, 3))).astype(np.float32)
    est = (frames["cam_t"][:, 2] + 1.0).astype(np.float32)
    for mode in ("rows", "dense"):
        fb = H.engine_batch_from_frames(dm, cfg, frames, range(B), lbs_mode=mode)
        assert fb.num_vars(0) == 88 and fb.nemb == 32
        fb.set_frames(frames["keypoints"], _jw(cfg, frames), _cmask(cfg, frames), frames["focal"],
                      np.tile([frames["W"] * 0.5, frames["H"] * 0.5], (B, 1)), 1000.0 / frames["H"], est_tz=est)
        fb.set_params(**P)
        Q = dict(P); Q["est_tz"] = est
        for stage in (-1, 0, 3, 4):
            loss, grad = fb.closure(stage)
            for i in range(B):
                lo, go = _oracle_closure(synth_model, cfg, frames, i, Q, stage)
                H.check_closure("vposer-full-%s" % mode, stage, loss[i], lo, grad[i], go)
        # decoded body pose of the accepted latent
        bp = fb.get_params()["body_pose"]
        from oracle.vposer import VPoserRef
        want = VPoserRef(vpw, torch.float64).decode(torch.tensor(P["pose_embedding"], dtype=torch.float64)).view(B, -1).numpy()
        assert np.abs(bp - want).max() < 2e-5


def test_fit_with_vposer_matches_reference(gpu, synth_model):
    """BASELINE config 3 end to end: the reference's fit_single_frame (with oracle SMPLXRef +
    VPoserRef plugged in) vs the engine, full K=135, VPoser latent, 5 stages
    (tests/golden/e2e_vposer.npz holds the reference's fp32 and fp64 runs)."""
    from smplifyx_amd import synthetic
    g = _golden("e2e_vposer")
    cfg = H.load_cfg("fit_smplx_smplifyx.yaml")
    dm = _dm(synth_model, cfg, vposer=synthetic.make_synthetic_vposer(0))
    frames = dict(keypoints=g["keypoints"], H=600, W=800, focal=5000.0)
    for mode in ("rows", "dense"):
        fb = H.engine_batch_from_frames(dm, cfg, frames, [0], lbs_mode=mode, reuse=True)
        fb.guess_init(cfg["body_tri_idxs"])
        fb.fit()
        st = fb.stats()
        ref32, ref64 = g["f0_f32_losses"], g["f0_f64_losses"]
        spread = np.abs(ref32 - ref64) / np.abs(ref64)
        rel = np.abs(st["stage_loss"][0] - ref32) / np.abs(ref32)
        assert rel[0] < 1e-4, (mode, st["stage_loss"][0], ref32)
        # non-smooth objective (4-branch quaternion in the decoder): later stages only land in the
        # same regime; two roundings of the HIP optimiser itself (sequential vs blocked two-loop
        # recursion) differ by up to 17 % here
        assert np.all(rel[1:5] < np.maximum(5 * spread[1:5], 2.5e-1)), (mode, rel, spread)
        # last stage (face landmarks weight 2): the dynamic-contour LUT makes the objective
        # non-smooth on the synthetic head (random landmark triangles); every implementation --
        # the reference in fp32 (133 evaluations) and fp64 (396), the engine -- stops after a
        # failed line search at a different point: 1.9e5 / 2.3e5 / 2.9e5..6.7e5.  Same regime only.
        assert np.isfinite(st["stage_loss"][0, 5]) and 0.2 < st["stage_loss"][0, 5] / ref32[5] < 5.0
