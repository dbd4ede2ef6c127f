"""Gaussian-mixture body pose prior (smplifyx/prior.py:100-231, body_prior_type 'gmm').
tests/golden/gmm.npz holds the REAL reference's MaxMixturePrior on a synthetic 8-component 63-D
mixture (tools/make_goldens.py gmm): buffers, values, autograd gradients, and one whole
fit_single_frame run that starts from the mixture's mean."""
import os
import pickle

import numpy as np
import pytest
import torch

import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gmm.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def _mixture(g):
    return dict(means=g["means"], covars=g["covars"], weights=g["weights"])


@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-10)])
def test_oracle_mixture_prior_matches_reference(g, tag, dtype, tol):
    from oracle.prior_gmm import MaxMixtureRef
    pr = MaxMixtureRef(g["means"], g["covars"], g["weights"], dtype)
    x = torch.tensor(g["poses"], dtype=dtype, requires_grad=True)
    val = pr(x)
    val.sum().backward()
    assert np.allclose(pr.get_mean().numpy(), g["mean_" + tag], rtol=tol, atol=tol)
    assert np.allclose(pr.nll_weights.numpy(), g["nll_weights_" + tag], rtol=10 * tol, atol=0)
    assert np.allclose(val.detach().numpy(), g["val_" + tag], rtol=tol, atol=0)
    assert np.linalg.norm(x.grad.numpy() - g["grad_" + tag]) <= tol * np.linalg.norm(g["grad_" + tag])


@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-10)])
def test_product_mixture_prior_module_matches_reference(g, tmp_path, tag, dtype, tol):
    """smplifyx_amd.prior.create_prior('gmm', prior_folder, num_gaussians) reading a gmm_08.pkl, the way
    smplifyx/main.py:176-181 builds it."""
    from smplifyx_amd import prior
    with open(tmp_path / "gmm_08.pkl", "wb") as fh:
        pickle.dump(_mixture(g), fh)
    pr = prior.create_prior("gmm", prior_folder=str(tmp_path), num_gaussians=8, dtype=dtype)
    x = torch.tensor(g["poses"], dtype=dtype, requires_grad=True)
    val = pr(x, None)
    val.sum().backward()
    assert np.allclose(pr.get_mean().numpy(), g["mean_" + tag], rtol=tol, atol=tol)
    assert np.allclose(pr.precisions.numpy(), g["precisions_" + tag], rtol=tol, atol=tol * np.abs(g["precisions_" + tag]).max())
    assert np.allclose(val.detach().numpy(), g["val_" + tag], rtol=tol, atol=0)
    assert np.linalg.norm(x.grad.numpy() - g["grad_" + tag]) <= tol * np.linalg.norm(g["grad_" + tag])
    unmerged = prior.MaxMixturePrior(gmm=_mixture(g), num_gaussians=8, dtype=dtype, use_merged=False)
    assert torch.isfinite(unmerged(x.detach()[:1])).all()         # (the per-component form is written for batch 1, as in the reference)
    with pytest.raises(FileNotFoundError):
        prior.create_prior("gmm", prior_folder=str(tmp_path / "nope"), num_gaussians=8)


@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-10)])
def test_per_component_form_matches_reference(g, tag, dtype, tol):
    """MaxMixturePrior(use_merged=False) (prior.py:203-231) against tests/golden/gmm_unmerged.npz: the REAL reference's
    module on the same mixture and poses (tools/make_goldens.py gmm_unmerged), one pose at a time like the reference runs."""
    from smplifyx_amd import prior
    u = np.load(os.path.join(os.path.dirname(GOLD), "gmm_unmerged.npz"))
    pr = prior.MaxMixturePrior(gmm=_mixture(g), num_gaussians=8, dtype=dtype, use_merged=False)
    for i in range(u["poses"].shape[0]):
        x = torch.tensor(u["poses"][i:i + 1], dtype=dtype, requires_grad=True)
        v = pr(x, None)
        v.sum().backward()
        assert abs(float(v.reshape(-1)[0]) - u["val_" + tag][i]) <= tol * abs(u["val_" + tag][i])
        assert np.linalg.norm(x.grad.numpy()[0] - u["grad_" + tag][i]) <= tol * np.linalg.norm(u["grad_" + tag][i])


def _cfg():
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False, use_vposer=False,
                     body_prior_type="gmm")
    cfg["use_camera_prior"] = False
    return cfg


def test_oracle_frame_driver_with_mixture_prior_matches_reference_fit(g, synth_model):
    """Whole schedule from the mixture's mean (fit_single_frame.py:250-252) in fp64 against the reference's fp64 run."""
    from oracle.fit_frame import FrameFit
    from oracle.prior_gmm import MaxMixtureRef
    cfg = _cfg()
    dtype = torch.float64
    K = g["keypoints"].shape[1]
    torch.set_num_threads(4)
    ff = FrameFit(H.oracle_model(synth_model, cfg, dtype), g["keypoints"][:1], 600, 800, 5000.0, cfg,
                  H.base_joint_weights(cfg, K), dtype=dtype,
                  body_pose_prior=MaxMixtureRef(g["means"], g["covars"], g["weights"], dtype))
    ref = ff.run()
    got = np.array([ref["cam_loss"]] + list(ref["stage_losses"]))
    want64, want32 = g["e2e_f64_losses"], g["e2e_f32_losses"]
    spread = np.abs(want32 - want64) / np.abs(want64)
    rel = np.abs(got - want64) / np.abs(want64)
    assert rel[0] < 1e-8
    assert np.all(rel < np.maximum(3 * spread, 1e-6)), (rel, spread)


@pytest.mark.gpu
def test_closure_with_mixture_prior_matches_oracle(g, synth_model):
    """HIP closure (loss + gradient with respect to the 182 variables) with the mixture prior against
    oracle autograd in fp64: loss 1e-5, gradient 1e-4 relative (helpers.check_closure), rows and dense paths."""
    import test_gpu_parity as T
    from oracle.fit_frame import FrameFit
    from oracle.prior_gmm import MaxMixtureRef
    from smplifyx_amd import engine, prior, synthetic
    cfg = _cfg()
    dm = T._dm(synth_model, cfg)
    B = 3
    K = len(H.joint_map_for(cfg))
    frames = synthetic.make_frames(B, H.oracle_joints_fn(synth_model, cfg), K, focal=5000.0)
    pr = prior.MaxMixturePrior(gmm=_mixture(g), num_gaussians=8)
    rng = np.random.RandomState(4)
    P = H.random_params(rng, B, scale=0.4)
    P["pose_embedding"] = (g["means"][[1, 4, 6]] + 0.2 * rng.normal(size=(B, 63))).astype(np.float32)
    P["cam_translation"] = (frames["cam_t"] + 0.3 * rng.normal(size=(B, 3))).astype(np.float32)
    est = (frames["cam_t"][:, 2] + 1.0).astype(np.float32)
    jw = np.tile(H.base_joint_weights(cfg, K), (B, 1))
    for mode in ("rows", "dense"):
        fb = engine.FrameBatch(dm, B, cfg, lbs_mode=mode, reuse_entry_eval=False, has_regression_pose=False)
        fb.set_gmm(pr)
        fb.set_frames(frames["keypoints"], jw, np.zeros((B, K), np.float32), frames["focal"],
                      np.tile([frames["W"] * 0.5, frames["H"] * 0.5], (B, 1)), 1000.0 / frames["H"], est_tz=est)
        fb.set_params(**P)
        for stage in (0, 2):
            loss, grad = fb.closure(stage)
            for i in range(B):
                ff = FrameFit(H.oracle_model(synth_model, cfg, torch.float64), frames["keypoints"][i:i + 1], frames["H"],
                              frames["W"], frames["focal"], cfg, H.base_joint_weights(cfg, K), dtype=torch.float64,
                              body_pose_prior=MaxMixtureRef(g["means"], g["covars"], g["weights"], torch.float64))
                bm = ff.bm
                with torch.no_grad():
                    for k, v in P.items():
                        if k == "pose_embedding": ff.pose_embedding.copy_(torch.tensor(v[i:i + 1], dtype=torch.float64))
                        elif k == "cam_translation": ff.cam_t.copy_(torch.tensor(v[i:i + 1], dtype=torch.float64))
                        else: getattr(bm, k).copy_(torch.tensor(v[i:i + 1], dtype=torch.float64))
                ps = [p for p in bm.parameters() if p.requires_grad] + [ff.pose_embedding]
                w = dict(ff.stages[stage]); w["data_weight"] = ff.data_weight
                w["bending_prior_weight"] = 3.17 * w["body_pose_weight"]
                for p in ps: p.grad = None
                lo = ff.body_terms(stage, w, ff.jw.clone())["total"]
                lo.backward()
                go = np.concatenate([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).numpy() for p in ps])
                lov = float(lo.detach())
                H.check_closure("gmm-body-%s" % mode, stage, loss[i], lov, grad[i], go)
        fb.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rows", "dense"])
def test_fit_with_mixture_prior_matches_reference_fit(g, synth_model, mode):
    """driver.fit_frames with body_prior_type 'gmm' (no regression prior: the body pose starts from the
    mixture's mean) against the reference's fit_single_frame, tolerances from the reference's own
    fp32 / fp64 difference as everywhere else (LAB_NOTES.md §3)."""
    import test_gpu_parity as T
    from smplifyx_amd import driver, prior
    cfg = _cfg()
    dm = T._dm(synth_model, cfg)
    K = g["keypoints"].shape[1]
    res = driver.fit_frames(dm, cfg, g["keypoints"][:1], H.base_joint_weights(cfg, K), 600, 800, 5000.0, lbs_mode=mode,
                            body_pose_prior=prior.MaxMixturePrior(gmm=_mixture(g), num_gaussians=8))
    want32, want64 = g["e2e_f32_losses"], g["e2e_f64_losses"]
    spread = np.abs(want32 - want64) / np.abs(want64)
    rel = np.abs(res["stage_loss"][0] - want32) / np.abs(want32)
    assert rel[0] < 1e-4, (rel, res["stage_loss"][0], want32)
    assert rel[1] < max(2 * spread[1], 3e-3), (rel, spread)
    assert np.all(rel[2:] < np.maximum(3 * spread[2:], 5e-2)), (rel, spread)
    # use_vposer False without a regression prior and without the mixture: the reference crashes, the driver says why
    cfg2 = dict(cfg); cfg2["body_prior_type"] = "l2"
    with pytest.raises(ValueError):
        driver.fit_frames(dm, cfg2, g["keypoints"][:1], H.base_joint_weights(cfg, K), 600, 800, 5000.0, lbs_mode=mode)


@pytest.mark.gpu
def test_drop_in_modules_with_mixture_prior(g, synth_model, tmp_path):
    """The reference's own entry points with body_prior_type 'gmm': fit_single_frame(body_pose_prior=
    create_prior('gmm', prior_folder, num_gaussians)) as smplifyx/main.py builds it, and one stage of the
    create_loss -> create_fitting_closure -> run_fitting sequence started from get_mean()
    (fit_single_frame.py:250-252)."""
    import test_gpu_dropin as DI
    from smplifyx_amd import fitting, prior
    from smplifyx_amd.fit_single_frame import fit_single_frame
    from smplifyx_amd.optimizers import optim_factory
    with open(tmp_path / "gmm_08.pkl", "wb") as fh:
        pickle.dump(_mixture(g), fh)
    cfg = _cfg()
    cfg["regression_prior"] = None
    cfg["prior_folder"], cfg["num_gaussians"] = str(tmp_path), 8
    dtype = torch.float32
    mk = lambda t: prior.create_prior(prior_type=t, dtype=dtype, prior_folder=str(tmp_path), num_gaussians=8)
    want32, want64 = g["e2e_f32_losses"], g["e2e_f64_losses"]
    spread = np.abs(want32 - want64) / np.abs(want64)
    # (1) fit_single_frame
    bm, camera = DI._setup(synth_model, cfg)
    a = dict(cfg); a["focal_length"] = 5000.0
    for k in ("result_folder", "output_folder", "mesh_folder"):
        a.pop(k, None)
    fn = str(tmp_path / "000.pkl")
    jw = torch.tensor(H.base_joint_weights(cfg, 25)).unsqueeze(0)
    result, final = fit_single_frame(np.zeros((600, 800, 3), np.float32), g["keypoints"][:1], body_model=bm, camera=camera,
                                     joint_weights=jw, dtype=dtype, shape_prior=mk("l2"), expr_prior=None,
                                     body_pose_prior=mk("gmm"), left_hand_prior=None, right_hand_prior=None, jaw_prior=None,
                                     angle_prior=mk("angle"), result_fn=fn, result_folder=str(tmp_path), **a)
    assert abs(final - want32[-1]) / want32[-1] < max(3 * spread[-1], 5e-2), (final, want32)
    assert pickle.load(open(fn, "rb"))["body_pose"].shape == (1, 63)
    # (2) fine-grained sequence, first body stage from the mixture's mean with the reference's camera result
    bm, camera = DI._setup(synth_model, cfg)
    dev = torch.device("cuda")
    gmm = mk("gmm").to(dev)
    pose_embedding = gmm.get_mean().clone().detach().requires_grad_(True)
    bm.reset_params(body_pose=pose_embedding)
    kd = torch.tensor(g["keypoints"][:1], dtype=dtype, device=dev)
    gt_joints, joints_conf = kd[:, :, :2], kd[:, :, 2].reshape(1, -1)
    joint_weights = torch.tensor(H.base_joint_weights(cfg, 25), device=dev).unsqueeze(0)
    low = [k for k in range(25) if float(joints_conf[0, k]) < cfg["confidence_threshold"]]
    joint_weights[:, low] = 0
    loss = fitting.create_loss(loss_type="smplify", joint_weights=joint_weights, rho=cfg["rho"], use_joints_conf=True,
                               use_face=False, use_hands=False, body_pose_prior=gmm, shape_prior=mk("l2"),
                               angle_prior=mk("angle"), interpenetration=False, dtype=dtype, regression_pose=None,
                               num_stages=3).to(dev)
    with torch.no_grad():
        camera.translation[:] = torch.tensor(g["e2e_f32_camera_translation"], dtype=dtype)
        camera.center[:] = torch.tensor([800, 600], dtype=dtype) * 0.5
    with fitting.FittingMonitor(**cfg) as monitor:
        final_params = [p for p in bm.parameters() if p.requires_grad] + [pose_embedding]
        body_opt, cg = optim_factory.create_optimizer(final_params, **cfg)
        w = {"data_weight": 1000 / 600, "body_pose_weight": torch.tensor(cfg["body_pose_prior_weights"][0], device=dev),
             "shape_weight": torch.tensor(cfg["shape_weights"][0], device=dev)}
        w["bending_prior_weight"] = 3.17 * w["body_pose_weight"]
        loss.reset_loss_weights(w)
        closure = monitor.create_fitting_closure(body_opt, bm, camera=camera, gt_joints=gt_joints, joints_conf=joints_conf,
                                                 joint_weights=joint_weights, loss=loss, create_graph=cg, use_vposer=False,
                                                 pose_embedding=pose_embedding, return_verts=True, return_full_pose=True)
        l0 = float(closure(stage=0))
        assert pose_embedding.grad is not None and np.isfinite(l0)
        # at the mean, before any step, the prior term alone is min_m(...) * w^2: the loss must exceed it
        pr = float(gmm(pose_embedding.detach(), None)) * cfg["body_pose_prior_weights"][0] ** 2
        assert l0 > pr > 0
        lf = monitor.run_fitting(body_opt, closure, final_params, bm, 0, pose_embedding=pose_embedding, use_vposer=False)
        assert lf < l0


@pytest.mark.gpu
def test_closure_with_per_component_mixture_prior_matches_reference_values(g, synth_model):
    """MaxMixturePrior(use_merged=False) on the device (sfx_batch_set_gmm_form): the closure's loss and embedding gradient
    change, relative to the merged form, by exactly what the reference's two forms differ by at the same pose
    (tests/golden/gmm.npz / gmm_unmerged.npz: values and gradients of the REAL reference), scaled by body_pose_weight^2."""
    import test_gpu_parity as T
    from smplifyx_amd import engine, prior, synthetic
    cfg = _cfg()
    dm = T._dm(synth_model, cfg)
    u = np.load(os.path.join(os.path.dirname(GOLD), "gmm_unmerged.npz"))
    idx = [2, 9, 17, 21]
    B = len(idx)
    K = len(H.joint_map_for(cfg))
    frames = synthetic.make_frames(B, H.oracle_joints_fn(synth_model, cfg), K, focal=5000.0)
    P = H.random_params(np.random.RandomState(4), B, scale=0.4)
    P["pose_embedding"] = u["poses"][idx].astype(np.float32)
    P["cam_translation"] = frames["cam_t"].astype(np.float32)
    est = (frames["cam_t"][:, 2] + 1.0).astype(np.float32)
    jw = np.tile(H.base_joint_weights(cfg, K), (B, 1))
    res = {}
    for merged in (True, False):
        pr = prior.MaxMixturePrior(gmm=_mixture(g), num_gaussians=8, use_merged=merged)
        fb = engine.FrameBatch(dm, B, cfg, lbs_mode="rows", reuse_entry_eval=False, has_regression_pose=False)
        fb.set_gmm(pr)
        fb.set_frames(frames["keypoints"], jw, np.zeros((B, K), np.float32), frames["focal"],
                      np.tile([frames["W"] * 0.5, frames["H"] * 0.5], (B, 1)), 1000.0 / frames["H"], est_tz=est)
        fb.set_params(**P)
        res[merged] = fb.closure(1)
        fb.close()
    bpw2 = float(cfg["body_pose_prior_weights"][1]) ** 2
    n_emb0 = 10 + 3 + 63 + 12 + 12 + 9 + 10             # embedding = last 63 entries of the 182 variables
    for q, i in enumerate(idx):
        # float32 poses: compare with the reference's fp64 values at the fp64 poses to 1e-4 (the poses were rounded to fp32)
        dv = (u["val_f64"][i] - g["val_f64"][i]) * bpw2
        dg = (u["grad_f64"][i] - g["grad_f64"][i]) * bpw2
        got_v = float(res[False][0][q]) - float(res[True][0][q])
        got_g = res[False][1][q][n_emb0:] - res[True][1][q][n_emb0:]
        scale = abs(u["val_f64"][i] * bpw2)
        assert abs(got_v - dv) <= 2e-4 * scale, (i, got_v, dv, scale)
        assert np.linalg.norm(got_g - dg) <= 2e-4 * np.linalg.norm(u["grad_f64"][i] * bpw2), (i,)
