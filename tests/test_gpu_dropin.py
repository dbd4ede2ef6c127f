"""GPU tests of the drop-in call surface: the reference's own call sequence
-- smplx.create / create_camera / create_prior / create_loss / FittingMonitor /
create_optimizer / create_fitting_closure / run_fitting / fit_single_frame -- executed against
this package's modules, checked against goldens of the real reference."""
import os
import pickle

import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup(synth_model, cfg, vposer=None):
    from smplifyx_amd import smplx, utils
    from smplifyx_amd.camera import create_camera
    jm = utils.JointMapper(H.joint_map_for(cfg))
    model_params = dict(model_path=synth_model, joint_mapper=jm, create_global_orient=True,
                        create_body_pose=not cfg.get("use_vposer"), create_betas=True, create_left_hand_pose=True,
                        create_right_hand_pose=True, create_expression=True, create_jaw_pose=True,
                        create_leye_pose=True, create_reye_pose=True, create_transl=False, dtype=torch.float32)
    args = {k: v for k, v in cfg.items() if k not in model_params and k != "gender"}
    bm = smplx.create(gender="neutral", vposer=vposer, **model_params, **args).to("cuda")
    cam = create_camera(focal_length_x=5000.0, focal_length_y=5000.0, dtype=torch.float32, **cfg).to("cuda")
    cam.rotation.requires_grad = False
    return bm, cam


def test_reference_call_sequence_per_stage(synth_model):
    """The body of fit_single_frame.py:413-612, written with this package's modules exactly
    as the reference writes it, must reproduce the reference's per-stage losses."""
    from smplifyx_amd import fitting, prior
    from smplifyx_amd.optimizers import optim_factory
    g = np.load(os.path.join(GOLD, "e2e_synth.npz"))
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)
    cfg["use_camera_prior"] = False
    bm, camera = _setup(synth_model, cfg)
    dev = torch.device("cuda")
    i = 0
    dtype = torch.float32
    keypoints = g["keypoints"][i:i + 1]
    pose_embedding = torch.tensor(g["reg_pose"][i:i + 1], device=dev, requires_grad=True)
    global_pose = torch.tensor(g["reg_global"][i:i + 1], device=dev)
    bm.reset_params(global_orient=global_pose, body_pose=pose_embedding)
    kd = torch.tensor(keypoints, dtype=dtype, device=dev)
    gt_joints, joints_conf = kd[:, :, :2], kd[:, :, 2].reshape(1, -1)
    joint_weights = torch.tensor(H.base_joint_weights(cfg, 25), device=dev).unsqueeze(0)
    low = [k for k in range(25) if float(joints_conf[0, k]) < cfg["confidence_threshold"]]
    joint_weights[:, low] = 0
    init_idxs = [k for k in cfg["init_joints_idxs"]
                 if float(gt_joints[0, k, 0]) != 0 and float(gt_joints[0, k, 1]) != 0 and k not in low]
    init_t = fitting.guess_init(bm, gt_joints, cfg["body_tri_idxs"], use_vposer=False, pose_embedding=pose_embedding,
                                model_type="smplx", focal_length=5000.0, dtype=dtype).reshape(1, -1)
    with torch.no_grad():
        camera.translation[:] = init_t
        camera.center[:] = torch.tensor([800, 600], dtype=dtype) * 0.5
    mk = lambda t: prior.create_prior(prior_type=t, dtype=dtype)
    camera_loss = fitting.create_loss("camera_init", joints_conf=joints_conf, use_conf=cfg["use_conf_for_camera_init"],
                                      trans_estimation=init_t, init_joints_idxs=torch.tensor(init_idxs, device=dev),
                                      depth_loss_weight=1e2, dtype=dtype).to(dev)
    loss = fitting.create_loss(loss_type="smplify", joint_weights=joint_weights, rho=cfg["rho"], use_joints_conf=True,
                               use_face=False, use_hands=False, body_pose_prior=mk("l2"), shape_prior=mk("l2"),
                               angle_prior=mk("angle"), interpenetration=False, dtype=dtype,
                               regression_pose=pose_embedding.clone().detach(), num_stages=3).to(dev)
    losses = []
    with fitting.FittingMonitor(**cfg) as monitor:
        data_weight = 1000 / 600
        camera_loss.reset_loss_weights({"data_weight": data_weight})
        camera.translation.requires_grad = True
        bm.global_orient.requires_grad = True
        cam_params = [camera.translation, bm.global_orient]
        opt, cg = optim_factory.create_optimizer(cam_params, **cfg)
        fit_camera = monitor.create_fitting_closure(opt, bm, camera, gt_joints, camera_loss, create_graph=cg,
                                                    use_vposer=False, pose_embedding=pose_embedding,
                                                    return_full_pose=False, return_verts=False)
        # the closure alone: loss value + .grad on the optimised tensors
        l0 = fit_camera(stage=0)
        assert camera.translation.grad is not None and bm.global_orient.grad is not None
        assert torch.isfinite(l0)
        losses.append(monitor.run_fitting(opt, fit_camera, cam_params, bm, stage=0, use_vposer=False,
                                          pose_embedding=pose_embedding))
        orient = bm.global_orient.detach().cpu().numpy()
        bm.reset_params(global_orient=orient, body_pose=pose_embedding)
        bpw, sw = cfg["body_pose_prior_weights"], cfg["shape_weights"]
        for opt_idx in range(3):
            final_params = [p for p in bm.parameters() if p.requires_grad] + [pose_embedding]
            body_opt, cg = optim_factory.create_optimizer(final_params, **cfg)
            body_opt.zero_grad()
            w = {"data_weight": data_weight, "body_pose_weight": torch.tensor(bpw[opt_idx], device=dev),
                 "shape_weight": torch.tensor(sw[opt_idx], device=dev)}
            w["bending_prior_weight"] = 3.17 * w["body_pose_weight"]
            loss.reset_loss_weights(w)
            closure = monitor.create_fitting_closure(body_opt, bm, camera=camera, gt_joints=gt_joints,
                                                     joints_conf=joints_conf, joint_weights=joint_weights, loss=loss,
                                                     create_graph=cg, use_vposer=False, pose_embedding=pose_embedding,
                                                     return_verts=True, return_full_pose=True)
            losses.append(monitor.run_fitting(body_opt, closure, final_params, bm, opt_idx,
                                              pose_embedding=pose_embedding, use_vposer=False))
        out = bm(return_verts=True, body_pose=pose_embedding)
        assert out.vertices.shape == (1, 10475, 3) and out.joints.shape == (1, 25, 3)
    ref32, ref64 = g["f0_f32_losses"], g["f0_f64_losses"]
    spread = np.abs(ref32 - ref64) / np.abs(ref64)
    rel = np.abs(np.array(losses) - ref32) / np.abs(ref32)
    assert rel[0] < 1e-4 and rel[1] < max(spread[1], 2e-3), (losses, ref32)
    assert np.all(rel[2:] < np.maximum(3 * spread[2:], 5e-2)), (losses, ref32)


@pytest.mark.parametrize("maxiters,lbfgs_max_iter", [(3, None), (30, None), (30, 7)])
def test_optimizer_step_matches_run_fitting(synth_model, maxiters, lbfgs_max_iter):
    """Driving the outer loop by hand with optimizer.step(closure) (reference fitting.py:174-195:
    ftol on step-entry losses, gtol on var.grad) walks the same trajectory as run_fitting on
    device, bit for bit -- the optimiser state (history, H_diag, t) survives between step() calls.
    (30, 7): a caller's own LBFGS(max_iter=7) under FittingMonitor(maxiters=30) -- two numbers where optim_factory.py:15
    passes one; refused until round 4 -- agrees between the two drivers as well (the L-BFGS state survives the step boundary, so
    the iterates are those of max_iter = 30 cut into steps of 7: what differs is where run_fitting's tests look --
    tests/test_gpu_optimizer_steps.py compares that trace with the specification machine)."""
    from smplifyx_amd import fitting
    from smplifyx_amd.optimizers import optim_factory
    g = np.load(os.path.join(GOLD, "e2e_synth.npz"))
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)
    cfg["maxiters"] = maxiters
    dev = torch.device("cuda")
    res = []
    for manual in (False, True):
        bm, camera = _setup(synth_model, cfg)
        pose_embedding = torch.tensor(g["reg_pose"][:1], device=dev, requires_grad=True)
        bm.reset_params(global_orient=torch.tensor(g["reg_global"][:1]), body_pose=pose_embedding)
        kd = torch.tensor(g["keypoints"][:1], device=dev)
        gt, conf = kd[:, :, :2], kd[:, :, 2].reshape(1, -1)
        with torch.no_grad():
            camera.translation[:] = torch.tensor([[0.0, 0.0, 18.0]]); camera.center[:] = torch.tensor([400.0, 300.0])
        cl = fitting.create_loss("camera_init", joints_conf=conf, use_conf=True,
                                 trans_estimation=torch.tensor([[0.0, 0.0, 18.0]]),
                                 init_joints_idxs=torch.tensor([2, 5, 9, 12]), depth_loss_weight=1e2).to(dev)
        cl.reset_loss_weights({"data_weight": 1000 / 600})
        params = [camera.translation, bm.global_orient]
        opt, _ = optim_factory.create_optimizer(params, **cfg)
        if lbfgs_max_iter:
            from smplifyx_amd.optimizers.lbfgs_ls import LBFGS
            opt = LBFGS(params, lr=cfg.get("lr", 1.0), max_iter=lbfgs_max_iter, line_search_fn="strong_Wolfe")
        with fitting.FittingMonitor(**cfg) as mon:
            c = mon.create_fitting_closure(opt, bm, camera, gt, cl, use_vposer=False, pose_embedding=pose_embedding,
                                           return_verts=False)
            if manual:
                prev = None
                for n in range(cfg["maxiters"]):
                    l = opt.step(lambda: c(stage=0))
                    if n > 0 and prev is not None:
                        from smplifyx_amd.utils import rel_change
                        if rel_change(prev, l.item()) <= cfg["ftol"]:
                            break
                    if all(abs(float(v.grad.view(-1).max())) < cfg["gtol"] for v in params if v.grad is not None):
                        break                                   # fitting.py:191-193
                    prev = l.item()
                res.append((prev, camera.translation.detach().cpu().numpy().copy()))
            else:
                v = mon.run_fitting(opt, c, params, bm, 0, use_vposer=False, pose_embedding=pose_embedding)
                res.append((v, camera.translation.detach().cpu().numpy().copy()))
    assert res[0][0] == res[1][0], res
    assert np.array_equal(res[0][1], res[1][1]), res


def test_fit_single_frame_writes_reference_pickle(synth_model, tmp_path):
    from smplifyx_amd import prior
    from smplifyx_amd.fit_single_frame import fit_single_frame
    from scipy.spatial.transform import Rotation as Rot
    g = np.load(os.path.join(GOLD, "e2e_synth.npz"))
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)
    cfg["use_camera_prior"] = False
    cfg["regression_prior"] = "ExPose"
    bm, camera = _setup(synth_model, cfg)
    i = 1
    expose = {"body_pose": Rot.from_euler("XYZ", g["reg_pose"][i].reshape(21, 3).astype(np.float64)).as_matrix().astype(np.float32),
              "global_orient": Rot.from_euler("XYZ", g["reg_global"][i].astype(np.float64)[None]).as_matrix().astype(np.float32)}
    a = dict(cfg); a["focal_length"] = 5000.0
    for k in ("result_folder", "output_folder", "mesh_folder"):      # main.py pops these before the call
        a.pop(k, None)
    mk = lambda t: prior.create_prior(prior_type=t, dtype=torch.float32)
    fn = str(tmp_path / "000.pkl")
    jw = torch.tensor(H.base_joint_weights(cfg, 25)).unsqueeze(0)
    result, final = fit_single_frame(np.zeros((600, 800, 3), np.float32), g["keypoints"][i:i + 1], body_model=bm,
                                     camera=camera, joint_weights=jw, dtype=torch.float32, shape_prior=mk("l2"),
                                     expr_prior=None, body_pose_prior=mk("l2"), left_hand_prior=None, right_hand_prior=None,
                                     jaw_prior=None, angle_prior=mk("angle"), result_fn=fn, expose_results=expose,
                                     result_folder=str(tmp_path), **a)
    res = pickle.load(open(fn, "rb"))
    assert list(res.keys()) == ["camera_rotation", "camera_translation", "camera_center", "H", "W", "focal_length",
                                "betas", "global_orient", "body_pose", "left_hand_pose", "right_hand_pose",
                                "jaw_pose", "leye_pose", "reye_pose", "expression"]
    ref32, ref64 = g["f%d_f32_losses" % i], g["f%d_f64_losses" % i]
    spread = abs(ref32[-1] - ref64[-1]) / ref64[-1]
    assert abs(final - ref32[-1]) / ref32[-1] < max(3 * spread, 5e-2)
    assert res["body_pose"].shape == (1, 63) and res["camera_translation"].shape == (1, 3)
    assert np.abs(res["camera_translation"] - g["f%d_f32_camera_translation" % i]).max() < 5e-2


def test_vposer_latent_regression_prior(synth_model, tmp_path):
    """cfg_files/fit_smplx_combined_vposer_coco25.yaml: VPoser latent started from
    vposer.encode(regression prior) (fit_single_frame.py:245) and pulled towards it in the LAST
    stage only (fitting.py:391-395).  (i) the prior term is exactly body_pose_weight^2 |z - r|^2
    at the last stage and |z|^2 before; (ii) the drop-in fit_single_frame runs that cfg."""
    from smplifyx_amd import engine, prior, synthetic, vposer
    from smplifyx_amd.fit_single_frame import fit_single_frame
    from scipy.spatial.transform import Rotation as Rot
    g = np.load(os.path.join(GOLD, "e2e_synth.npz"))
    cfg = H.load_cfg("fit_smplx_combined_vposer_coco25.yaml", use_hands=False, use_face=False)
    assert cfg["use_vposer"] and cfg["regression_prior"] == "combined"
    cfg["use_camera_prior"] = False
    cfg["regression_prior"] = "ExPose"
    vpw = synthetic.make_synthetic_vposer(0, encoder_inputs=63)
    bm, camera = _setup(synth_model, cfg, vposer=vpw)
    dm = bm.device_model
    z0 = vposer.encode(vpw, g["reg_pose"][:1])
    assert z0.shape == (1, 32)
    # (i) analytic check of the latent prior term
    n_st = len(cfg["body_pose_prior_weights"])
    losses = {}
    for tag, reg in (("a", z0), ("b", z0 + 0.25)):
        fb = engine.FrameBatch(dm, 1, cfg, lbs_mode="rows", has_regression_pose=True)
        kp = g["keypoints"][:1]
        fb.set_frames(kp, np.tile(H.base_joint_weights(cfg, 25), (1, 1)), np.zeros((1, 25), np.float32), 5000.0,
                      np.array([[400.0, 300.0]], np.float32), 1000.0 / 600)
        z = z0 + 0.1
        fb.set_params(regression_pose=reg, pose_embedding=z, global_orient=g["reg_global"][:1],
                      cam_translation=np.array([[0.0, 0.0, 20.0]], np.float32))
        losses[tag] = [fb.closure(s)[0][0] for s in range(n_st)]
        fb.close()
    z = z0 + 0.1
    bpw = cfg["body_pose_prior_weights"]
    for s in range(n_st - 1):
        assert losses["a"][s] == losses["b"][s], s                  # target unused before the last stage
    want = bpw[-1] ** 2 * (np.sum((z - z0) ** 2) - np.sum((z - z0 - 0.25) ** 2))
    got = float(losses["a"][-1]) - float(losses["b"][-1])
    assert abs(got - want) <= 1e-3 * abs(want) + 1e-4 * abs(float(losses["a"][-1])), (got, want)
    # (ii) the whole drop-in call
    expose = {"body_pose": Rot.from_euler("XYZ", g["reg_pose"][0].reshape(21, 3).astype(np.float64)).as_matrix().astype(np.float32),
              "global_orient": Rot.from_euler("XYZ", g["reg_global"][0].astype(np.float64)[None]).as_matrix().astype(np.float32)}
    a = dict(cfg); a["focal_length"] = 5000.0
    for k in ("result_folder", "output_folder", "mesh_folder"):
        a.pop(k, None)
    mk = lambda t: prior.create_prior(prior_type=t, dtype=torch.float32)
    fn = str(tmp_path / "000.pkl")
    jw = torch.tensor(H.base_joint_weights(cfg, 25)).unsqueeze(0)
    result, final = fit_single_frame(np.zeros((600, 800, 3), np.float32), g["keypoints"][:1], body_model=bm,
                                     camera=camera, joint_weights=jw, dtype=torch.float32, shape_prior=mk("l2"),
                                     expr_prior=None, body_pose_prior=mk("l2"), left_hand_prior=None, right_hand_prior=None,
                                     jaw_prior=None, angle_prior=mk("angle"), result_fn=fn, expose_results=expose,
                                     result_folder=str(tmp_path), **a)
    assert np.isfinite(final) and result["body_pose"].shape == (1, 63)
    assert np.all(np.isfinite(result["body_pose"])) and np.all(np.isfinite(result["camera_translation"]))


def test_batched_main_writes_reference_layout(synth_model, tmp_path):
    """smplifyx_amd.main.main(**cfg): the reference's main.py interface (dataset folder, model
    folder, ExPose result files, output layout) with all frames fitted as one batch."""
    import json
    from PIL import Image
    from scipy.spatial.transform import Rotation as Rot
    from smplifyx_amd import main as amd_main
    g = np.load(os.path.join(GOLD, "e2e_synth.npz"))
    data, models, expose_dir, out = tmp_path / "data", tmp_path / "models" / "smplx", tmp_path / "expose", tmp_path / "out"
    for d in (data / "images", data / "keypoints", models, expose_dir):
        os.makedirs(d)
    np.savez(models / "SMPLX_NEUTRAL.npz", **synth_model)
    names = ["frame_a", "frame_b"]
    for i, n in enumerate(names):
        Image.fromarray(np.zeros((600, 800, 3), np.uint8)).save(data / "images" / (n + ".png"))
        json.dump({"people": [{"pose_keypoints_2d": [float(v) for v in g["keypoints"][i].reshape(-1)],
                               "hand_left_keypoints_2d": [], "hand_right_keypoints_2d": [], "face_keypoints_2d": []}]},
                  open(data / "keypoints" / (n + "_keypoints.json"), "w"))
        os.makedirs(expose_dir / (n + ".jpg"))
        np.savez(expose_dir / (n + ".jpg") / (n + ".jpg_params.npz"),
                 body_pose=Rot.from_euler("XYZ", g["reg_pose"][i].reshape(21, 3).astype(np.float64)).as_matrix().astype(np.float32),
                 global_orient=Rot.from_euler("XYZ", g["reg_global"][i].astype(np.float64)[None]).as_matrix().astype(np.float32))
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)
    cfg.update(use_camera_prior=False, use_face_contour=False, regression_prior="ExPose", data_folder=str(data), model_folder=str(tmp_path / "models"),
               output_folder=str(out), expose_results_directory=str(expose_dir), pixie_results_directory=None,
               focal_length=5000.0, save_vertices=True)
    n = amd_main.main(**cfg)
    assert n == 2 and os.path.isfile(out / "conf.yaml")
    for i, nme in enumerate(names):
        res = pickle.load(open(out / "results" / nme / "000.pkl", "rb"))
        assert list(res.keys()) == ["camera_rotation", "camera_translation", "camera_center", "H", "W", "focal_length",
                                    "betas", "global_orient", "body_pose", "left_hand_pose", "right_hand_pose",
                                    "jaw_pose", "leye_pose", "reye_pose", "expression"]
        assert res["H"] == 600 and res["W"] == 800 and res["camera_center"].tolist() == [[400.0, 300.0]]
        assert np.abs(res["camera_translation"] - g["f%d_f32_camera_translation" % i]).max() < 5e-2
        assert np.abs(res["betas"] - g["f%d_f32_betas" % i]).max() < 0.25
        assert os.path.isdir(out / "meshes" / nme) and os.path.isdir(out / "images" / nme / "000")
        assert os.path.getsize(out / "results" / nme / "vertices.ply") > 10475 * 12


@pytest.mark.parametrize("optim_type,lr,iters", [("adam", 0.01, 25), ("lbfgs", 1.0, 5), ("sgd", None, 10), ("rmsprop", 1e-3, 10)])
def test_other_optimizers_are_driven_from_the_host(synth_model, optim_type, lr, iters):
    """optim_factory's adam / lbfgs / sgd / rmsprop (smplifyx/optimizers/optim_factory.py:45-63): torch.optim
    objects stepping the caller's tensors, every closure evaluation (loss + .grad) on the device.  The
    camera stage (6 variables, well conditioned) must make progress with each of them; the batched
    driver refuses anything but 'lbfgsls' loudly."""
    from smplifyx_amd import driver, fitting
    from smplifyx_amd.optimizers import optim_factory
    g = np.load(os.path.join(GOLD, "e2e_synth.npz"))
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)
    cfg.update(use_camera_prior=False, optim_type=optim_type, lr=lr if lr is not None else 1.0, maxiters=iters)
    bm, camera = _setup(synth_model, cfg)
    dev = torch.device("cuda")
    dtype = torch.float32
    pose_embedding = torch.tensor(g["reg_pose"][:1], device=dev, requires_grad=True)
    bm.reset_params(global_orient=torch.tensor(g["reg_global"][:1], device=dev), body_pose=pose_embedding)
    kd = torch.tensor(g["keypoints"][:1], dtype=dtype, device=dev)
    gt_joints, joints_conf = kd[:, :, :2], kd[:, :, 2].reshape(1, -1)
    init_idxs = [k for k in cfg["init_joints_idxs"] if float(gt_joints[0, k, 0]) != 0]
    init_t = fitting.guess_init(bm, gt_joints, cfg["body_tri_idxs"], use_vposer=False, pose_embedding=pose_embedding,
                                model_type="smplx", focal_length=5000.0, dtype=dtype).reshape(1, -1)
    with torch.no_grad():
        camera.translation[:] = init_t
        camera.center[:] = torch.tensor([800, 600], dtype=dtype) * 0.5
    camera_loss = fitting.create_loss("camera_init", joints_conf=joints_conf, use_conf=cfg["use_conf_for_camera_init"],
                                      trans_estimation=init_t, init_joints_idxs=torch.tensor(init_idxs, device=dev),
                                      depth_loss_weight=1e2, dtype=dtype).to(dev)
    with fitting.FittingMonitor(**cfg) as monitor:
        camera_loss.reset_loss_weights({"data_weight": 1000 / 600})
        camera.translation.requires_grad = True
        bm.global_orient.requires_grad = True
        cam_params = [camera.translation, bm.global_orient]
        opt, cg = optim_factory.create_optimizer(cam_params, **cfg)
        assert type(opt).__module__.startswith("torch.optim")
        fit_camera = monitor.create_fitting_closure(opt, bm, camera, gt_joints, camera_loss, create_graph=cg, use_vposer=False,
                                                    pose_embedding=pose_embedding, return_full_pose=False, return_verts=False)
        l0 = float(fit_camera(stage=0))
        if lr is None:      # plain gradient descent needs a step matched to the gradient's scale: move 1 cm per step at most
            gmax = max(float(p.grad.abs().max()) for p in cam_params)
            for grp in opt.param_groups:
                grp["lr"] = 0.01 / gmax
                grp["momentum"] = 0.0
                grp["nesterov"] = False
        t0 = camera.translation.detach().clone()
        lf = monitor.run_fitting(opt, fit_camera, cam_params, bm, stage=0, use_vposer=False, pose_embedding=pose_embedding)
        l1 = float(fit_camera(stage=0))
        assert lf is not None and np.isfinite(lf) and np.isfinite(l1) and l1 < l0, (optim_type, l0, lf, l1)
        assert not torch.equal(camera.translation.detach(), t0)
    with pytest.raises(NotImplementedError):
        driver.fit_frames(bm.device_model, cfg, g["keypoints"][:1], H.base_joint_weights(cfg, 25), 600, 800, 5000.0,
                          reg_pose=g["reg_pose"][:1], reg_global=g["reg_global"][:1])


@pytest.mark.parametrize("yaml_", ["fit_smplx_combined_coco25.yaml", "fit_smplx_combined_halpe.yaml",
                                   "fit_smplx_combined_vposer_coco25.yaml", "fit_smplx_smplifyx.yaml"])
def test_main_runs_every_shipped_cfg_unmodified(tmp_path, yaml_):
    """`main(**parse_config(['-c', cfg]))` with each cfg_files/*.yaml AS SHIPPED -- hands + face + contour keypoints,
    interpenetration, regression / camera priors, VPoser, and the flags the reference's cfgs carry for things outside the
    fitting path (visualize, interactive, use_gender_classifier: warnings, not errors).  Only locations are overridden
    (data / model / output folders, regression result directories, part segmentation and VPoser files)."""
    import json
    import warnings
    import joblib
    from PIL import Image
    from scipy.spatial.transform import Rotation as Rot
    from smplifyx_amd import cmd_parser, main as amd_main, synthetic
    model = synthetic.make_synthetic_model(0, surface=True)
    parts = synthetic.make_synthetic_parts(model)
    raw = cmd_parser.load_config(os.path.join(H.CFG_DIR, yaml_), {})
    assert raw["visualize"] and raw["use_gender_classifier"] and raw["interpenetration"] and raw["use_hands"] and raw["use_face"]
    K = len(H.joint_map_for(raw))
    nb = {"coco25": 25, "halpe": 26}[raw["format"]]
    assert K == nb + 42 + 51 + 17
    frames = synthetic.make_frames(2, H.oracle_joints_fn(model, raw), K, focal=5000.0)
    data, models, out = tmp_path / "data", tmp_path / "models" / "smplx", tmp_path / "out"
    expose_dir, pixie_dir = tmp_path / "expose", tmp_path / "pixie"
    for d in (data / "images", data / "keypoints", models, expose_dir, pixie_dir):
        os.makedirs(d)
    np.savez(models / "SMPLX_NEUTRAL.npz", **model)
    with open(tmp_path / "parts.pkl", "wb") as fh:
        pickle.dump({"segm": parts["segm"], "parents": parts["parents"]}, fh, protocol=2)
    np.savez(tmp_path / "vposer.npz", **synthetic.make_synthetic_vposer(0, encoder_inputs=63))
    names = ["frame_a", "frame_b"]
    rot = lambda a: Rot.from_euler("XYZ", np.asarray(a, np.float64).reshape(-1, 3)).as_matrix().astype(np.float32)
    for i, n in enumerate(names):
        Image.fromarray(np.zeros((600, 800, 3), np.uint8)).save(data / "images" / (n + ".png"))
        kp = frames["keypoints"][i]
        face = np.concatenate([kp[nb + 42 + 51:], kp[nb + 42:nb + 42 + 51]])         # json order: 17 contour, then 51 inner
        flat = lambda a: [float(v) for v in np.asarray(a).reshape(-1)]
        json.dump({"people": [{"pose_keypoints_2d": flat(kp[:nb]), "hand_left_keypoints_2d": flat(kp[nb:nb + 21]),
                               "hand_right_keypoints_2d": flat(kp[nb + 21:nb + 42]), "face_keypoints_2d": flat(face)}]},
                  open(data / "keypoints" / (n + "_keypoints.json"), "w"))
        os.makedirs(expose_dir / (n + ".jpg")); os.makedirs(pixie_dir / n)
        np.savez(expose_dir / (n + ".jpg") / (n + ".jpg_params.npz"), body_pose=rot(frames["reg_pose"][i]),
                 global_orient=rot(frames["reg_global"][i]), transl=frames["cam_t"][i].astype(np.float64),
                 center=np.array([400.0, 300.0], np.float32))
        joblib.dump({"body_pose": rot(frames["reg_pose"][i]), "global_pose": rot(frames["reg_global"][i])},
                    pixie_dir / n / (n + "_param.pkl"))
    cfg = cmd_parser.parse_config(["-c", os.path.join(H.CFG_DIR, yaml_), "--data_folder", str(data), "--model_folder",
                                   str(tmp_path / "models"), "--output_folder", str(out), "--part_segm_fn", str(tmp_path / "parts.pkl"),
                                   "--vposer_ckpt", str(tmp_path / "vposer.npz"), "--expose_results_directory", str(expose_dir),
                                   "--pixie_results_directory", str(pixie_dir), "--focal_length", "5000"])
    for k in ("visualize", "interactive", "use_gender_classifier", "interpenetration", "use_hands", "use_face", "maxiters",
              "df_cone_height", "max_collisions", "use_vposer", "regression_prior", "use_camera_prior"):
        assert cfg[k] == raw[k], k
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        n = amd_main.main(**cfg)
    assert n == 2
    msgs = " ".join(str(w.message) for w in wlist)
    assert "use_gender_classifier" in msgs and "visualize" in msgs
    for nme in names:
        res = pickle.load(open(out / "results" / nme / "000.pkl", "rb"))
        assert res["left_hand_pose"].shape == (1, 12) and res["expression"].shape == (1, 10) and res["body_pose"].shape == (1, 63)
        assert all(np.all(np.isfinite(np.asarray(v, np.float64))) for v in res.values())
        assert os.path.getsize(out / "results" / nme / "vertices.ply") > 10475 * 12


def test_create_loss_with_interpenetration_objects(synth_model):
    """fit_single_frame.py:300-328 + fitting.py:437-455: create_loss(search_tree=BVH(...), pen_distance=
    DistanceFieldPenetrationLoss(...), tri_filtering_module=FilterFaces(...), interpenetration=True) through the fitting
    closure: the value and gradients equal those of the engine batch built from the cfg keys (the route fit_single_frame /
    main take), and the term is really in (coll_loss_weight 0 gives a smaller loss)."""
    from smplifyx_amd import fitting, prior, synthetic, engine
    from smplifyx_amd.mesh_intersection.bvh_search_tree import BVH
    import smplifyx_amd.mesh_intersection.loss as collisions_loss
    from smplifyx_amd.mesh_intersection.filter_faces import FilterFaces
    import test_gpu_parity as T
    cfg = H.load_cfg("fit_smplx_combined_halpe.yaml", use_hands=False, use_face=False, interpenetration=True)
    cfg["use_camera_prior"] = False
    cfg["df_cone_height"] = 1e-2
    model = synthetic.make_synthetic_model(0, surface=True)
    parts = synthetic.make_synthetic_parts(model)
    bm, camera = _setup(model, cfg)
    dev = torch.device("cuda")
    K = len(H.joint_map_for(cfg))
    frames = synthetic.make_frames(1, H.oracle_joints_fn(model, cfg), K, focal=5000.0)
    kd = torch.tensor(frames["keypoints"][:1], device=dev)
    gt_joints, joints_conf = kd[:, :, :2], kd[:, :, 2].reshape(1, -1)
    joint_weights = torch.tensor(H.base_joint_weights(cfg, K), device=dev).unsqueeze(0)
    rng = np.random.RandomState(3)
    pose = (frames["reg_pose"][:1] + 0.3 * rng.normal(size=(1, 63))).astype(np.float32)      # bent enough to collide
    pose_embedding = torch.tensor(pose, device=dev, requires_grad=True)
    bm.reset_params(global_orient=frames["reg_global"][:1], body_pose=pose_embedding.detach())
    with torch.no_grad():
        camera.translation[:] = torch.tensor(frames["cam_t"][:1], device=dev)
        camera.center[:] = torch.tensor([frames["W"] * 0.5, frames["H"] * 0.5], device=dev)
    search_tree = BVH(max_collisions=cfg["max_collisions"])
    pen_distance = collisions_loss.DistanceFieldPenetrationLoss(sigma=cfg["df_cone_height"], point2plane=False, vectorized=True,
                                                                penalize_outside=cfg["penalize_outside"])
    filter_faces = FilterFaces(faces_segm=parts["segm"], faces_parents=parts["parents"], ign_part_pairs=cfg["ign_part_pairs"]).to(dev)
    pen_p2p = collisions_loss.DistanceFieldPenetrationLoss(sigma=cfg["df_cone_height"], point2plane=True, vectorized=True,
                                                           penalize_outside=cfg["penalize_outside"])
    assert pen_p2p.point2plane and not pen_distance.point2plane
    mk = lambda t: prior.create_prior(prior_type=t, dtype=torch.float32)
    vals = {}
    for cw in (1.0, 0.0):
        loss = fitting.create_loss(loss_type="smplify", joint_weights=joint_weights, rho=cfg["rho"], use_joints_conf=True,
                                   use_face=False, use_hands=False, body_pose_prior=mk("l2"), shape_prior=mk("l2"),
                                   angle_prior=mk("angle"), interpenetration=True, search_tree=search_tree,
                                   pen_distance=pen_distance, tri_filtering_module=filter_faces, dtype=torch.float32,
                                   regression_pose=torch.tensor(frames["reg_pose"][:1], device=dev), num_stages=3).to(dev)
        w = {"data_weight": 1000.0 / frames["H"], "body_pose_weight": torch.tensor(cfg["body_pose_prior_weights"][2], device=dev),
             "shape_weight": torch.tensor(cfg["shape_weights"][2], device=dev), "coll_loss_weight": torch.tensor(cw, device=dev)}
        w["bending_prior_weight"] = 3.17 * w["body_pose_weight"]
        loss.reset_loss_weights(w)
        with fitting.FittingMonitor(**cfg) as monitor:
            closure = monitor.create_fitting_closure(None, bm, camera=camera, gt_joints=gt_joints, joints_conf=joints_conf,
                                                     joint_weights=joint_weights, loss=loss, use_vposer=False,
                                                     pose_embedding=pose_embedding, return_verts=True, return_full_pose=True)
            vals[cw] = (float(closure(stage=2)), pose_embedding.grad.detach().cpu().numpy().copy())
            if closure._fb is not None:
                closure._fb.close()
    assert vals[1.0][0] > vals[0.0][0] * (1 + 1e-6), vals         # the term is in
    # the same through the engine batch of the cfg route
    dm = bm.device_model
    fb = H.engine_batch_from_frames(dm, cfg, frames, [0], lbs_mode="dense")
    fb.set_frames(frames["keypoints"][:1], joint_weights.cpu().numpy(), np.zeros((1, K), np.float32), frames["focal"],
                  np.array([[frames["W"] * 0.5, frames["H"] * 0.5]]), 1000.0 / frames["H"])
    fb.set_params(regression_pose=frames["reg_pose"][:1], global_orient=frames["reg_global"][:1], pose_embedding=pose,
                  cam_translation=frames["cam_t"][:1].astype(np.float32))
    l2, g2 = fb.closure(2)
    assert float(l2[0]) == vals[1.0][0]
    assert np.array_equal(g2[0][-63:], vals[1.0][1].reshape(-1))
    fb.close()
    # the part filter belongs to the loss: one built WITHOUT a FilterFaces module after the filtered ones above filters nothing
    # (it used to inherit the previous loss's labels through the shared device model) -- more pairs, a larger term
    def closure_value(tf, pd=pen_distance):
        loss = fitting.create_loss(loss_type="smplify", joint_weights=joint_weights, rho=cfg["rho"], use_joints_conf=True,
                                   use_face=False, use_hands=False, body_pose_prior=mk("l2"), shape_prior=mk("l2"),
                                   angle_prior=mk("angle"), interpenetration=True, search_tree=search_tree,
                                   pen_distance=pd, tri_filtering_module=tf, dtype=torch.float32,
                                   regression_pose=torch.tensor(frames["reg_pose"][:1], device=dev), num_stages=3).to(dev)
        w = {"data_weight": 1000.0 / frames["H"], "body_pose_weight": torch.tensor(cfg["body_pose_prior_weights"][2], device=dev),
             "shape_weight": torch.tensor(cfg["shape_weights"][2], device=dev), "coll_loss_weight": torch.tensor(1.0, device=dev)}
        w["bending_prior_weight"] = 3.17 * w["body_pose_weight"]
        loss.reset_loss_weights(w)
        with fitting.FittingMonitor(**cfg) as monitor:
            closure = monitor.create_fitting_closure(None, bm, camera=camera, gt_joints=gt_joints, joints_conf=joints_conf,
                                                     joint_weights=joint_weights, loss=loss, use_vposer=False,
                                                     pose_embedding=pose_embedding, return_verts=True, return_full_pose=True)
            v = float(closure(stage=2))
            if closure._fb is not None:
                closure._fb.close()
        return v
    unfiltered = closure_value(None)
    assert unfiltered > vals[1.0][0] * (1 + 1e-6), (unfiltered, vals[1.0][0])
    assert closure_value(filter_faces) == vals[1.0][0]              # and back again
    assert closure_value(None) == unfiltered
    # DistanceFieldPenetrationLoss(point2plane=True) (fit_single_frame.py:93,314) reaches the device: every Psi^2 weighted by
    # (n_f . n_g)^2 <= 1 -- a smaller term on the same pairs --, and the default form is back afterwards
    p2p = closure_value(filter_faces, pen_p2p)
    assert vals[0.0][0] < p2p < vals[1.0][0], (vals[0.0][0], p2p, vals[1.0][0])
    assert closure_value(filter_faces) == vals[1.0][0]
