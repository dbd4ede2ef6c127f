#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (imported from
/root/reference, this container only) on seeded inputs.  The reference Python never
ships; only these numeric arrays do.  Re-run: `python tools/make_goldens.py [what ...]`
with what in {objective, lbfgs, euler, tables, e2e, demo, e2e_vposer, parser, eval, gmm, e2e_full, e2e_full_set, e2e_vposer_set, e2e_side, e2e_bench, gmm_unmerged, e2e_bench_raw_delta, smplx_topology, e2e_pen_set, objective_pen, cubic_overflow}.

The LBS itself has no reference implementation here (external `smplx`, absent): wherever a
body model is needed the reference drives oracle.body_model.SMPLXRef built from
smplifyx_amd.synthetic.make_synthetic_model(0) -- so these goldens pin everything that IS
in the reference tree (loss, camera, priors, L-BFGS, schedule) and, for e2e, the
composition of all of it.
"""
import contextlib
import io
import json
import os
import pickle
import sys
import tempfile
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import ref_import  # noqa: E402

PEN_DEFAULT_FRAMES = "0,1,2,3,4,5,6,7"
GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
ref = ref_import.import_reference()
torch.set_num_threads(1)


def _save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: np.asarray(v).shape for k, v in arrays.items()})


# ------------------------------------------------------------------------------------------
def gen_objective():
    """GMoF, camera, priors, SMPLifyLoss (value, every gradient), camera-init loss."""
    from collections import namedtuple
    rng = np.random.RandomState(0)
    out = {}
    x = rng.normal(size=(4, 7, 2)) * 80
    out["gmof_in"] = x
    out["gmof_out"] = ref.utils.GMoF(rho=100)(torch.tensor(x)).numpy()
    # camera
    B, K = 1, 9
    pts = torch.tensor(rng.normal(size=(B, K, 3)) + [0, 0, 8.0], requires_grad=True)
    cam = ref.camera.create_camera(focal_length_x=1234.5, focal_length_y=1234.5, dtype=torch.float64,
                                   center=torch.tensor([[400.0, 300.0]], dtype=torch.float64))
    with torch.no_grad():
        cam.translation[:] = torch.tensor([[0.1, -0.2, 1.5]], dtype=torch.float64)
    uv = cam(pts)
    (uv * torch.tensor(rng.normal(size=(B, K, 2)))).sum().backward()
    out.update(cam_pts=pts.detach().numpy(), cam_uv=uv.detach().numpy(), cam_dpts=pts.grad.numpy(),
               cam_dt=cam.translation.grad.numpy(), cam_t=cam.translation.detach().numpy())
    # priors
    pose = rng.normal(size=(1, 63)) * 0.5
    out["angle_in"] = pose
    out["angle_out"] = ref.prior.SMPLifyAnglePrior(dtype=torch.float64)(torch.tensor(pose)).numpy()
    out["relchange"] = np.array([ref.utils.rel_change(a, b) for a, b in [(10.0, 9.0), (0.5, 0.6), (-3.0, 2.0)]])
    # SMPLifyLoss at every stage of the combined schedule, body-only and full
    MO = namedtuple("MO", ["joints", "full_pose", "betas", "body_pose", "left_hand_pose", "right_hand_pose",
                           "expression", "jaw_pose", "vertices"])
    for tag, K, uh, uf in (("body", 25, False, False), ("full", 135, True, True)):
        mk = lambda *s: torch.tensor(rng.normal(size=s), dtype=torch.float64, requires_grad=True)
        joints = torch.tensor(rng.normal(size=(1, K, 3)) * 0.4 + [0, 0, 9.0], dtype=torch.float64, requires_grad=True)
        full_pose, betas = mk(1, 165), mk(1, 10)
        lh, rh, expr, jaw = mk(1, 45), mk(1, 45), mk(1, 10), mk(1, 3)
        emb = mk(1, 63)
        reg = torch.tensor(rng.normal(size=(1, 63)), dtype=torch.float64)
        gt = torch.tensor(rng.normal(size=(1, K, 2)) * 60 + [400, 300], dtype=torch.float64)
        conf = torch.tensor(rng.uniform(size=(1, K)), dtype=torch.float64)
        jw = torch.tensor((rng.uniform(size=(1, K)) > 0.2).astype(np.float64))
        cam = ref.camera.create_camera(focal_length_x=5000.0, focal_length_y=5000.0, dtype=torch.float64,
                                       center=torch.tensor([[400.0, 300.0]], dtype=torch.float64))
        with torch.no_grad():
            cam.translation[:] = torch.tensor([[0.05, 0.1, 20.0]], dtype=torch.float64)
        pri = lambda t: ref.prior.create_prior(prior_type=t, dtype=torch.float64)
        loss = ref.fitting.create_loss("smplify", rho=100, use_joints_conf=True, use_face=uf, use_hands=uh,
                                       body_pose_prior=pri("l2"), shape_prior=pri("l2"), angle_prior=pri("angle"),
                                       expr_prior=pri("l2"), left_hand_prior=pri("l2"), right_hand_prior=pri("l2"),
                                       jaw_prior=pri("l2"), interpenetration=False, dtype=torch.float64,
                                       regression_pose=reg, num_stages=3)
        W = dict(data_weight=1000 / 600, body_pose_weight=300.0, shape_weight=50.0, bending_prior_weight=3.17 * 300.0,
                 hand_prior_weight=4.78, expr_prior_weight=5.0,
                 jaw_prior_weight=torch.tensor([100.0, 1000.0, 1000.0], dtype=torch.float64))
        loss.reset_loss_weights(W)
        mo = MO(joints, full_pose, betas, emb, lh, rh, expr, jaw, None)
        total = loss(mo, camera=cam, gt_joints=gt, joints_conf=conf, body_model_faces=None, joint_weights=jw,
                     stage=1, use_vposer=False, pose_embedding=emb)
        total.backward()
        g = lambda t: (t.grad.numpy() if t.grad is not None else np.zeros(tuple(t.shape)))
        out.update({tag + "_" + k: v for k, v in dict(
            joints=joints.detach().numpy(), full_pose=full_pose.detach().numpy(), betas=betas.detach().numpy(),
            lh=lh.detach().numpy(), rh=rh.detach().numpy(), expr=expr.detach().numpy(), jaw=jaw.detach().numpy(),
            emb=emb.detach().numpy(), reg=reg.numpy(), gt=gt.numpy(), conf=conf.numpy(), jw=jw.numpy(),
            total=np.array(total.item()), d_joints=g(joints), d_full_pose=g(full_pose), d_betas=g(betas),
            d_lh=g(lh), d_rh=g(rh), d_expr=g(expr), d_jaw=g(jaw), d_emb=g(emb), d_cam_t=cam.translation.grad.numpy(),
        ).items()})
        # camera-init loss with and without the use_conf quirk
        for uc in (False, True):
            cl = ref.fitting.create_loss("camera_init", joints_conf=conf, use_conf=uc,
                                         trans_estimation=torch.tensor([[0.0, 0.0, 18.0]], dtype=torch.float64),
                                         init_joints_idxs=torch.tensor([9, 12, 2, 5]), depth_loss_weight=1e2,
                                         dtype=torch.float64)
            cl.reset_loss_weights({"data_weight": 1000 / 600})
            j2 = joints.detach().clone().requires_grad_(True)
            cam.translation.grad = None
            v = cl(MO(j2, None, None, None, None, None, None, None, None), camera=cam, gt_joints=gt)
            v.backward()
            out["%s_caminit%d" % (tag, uc)] = np.array(v.item())
            out["%s_caminit%d_dj" % (tag, uc)] = j2.grad.numpy()
            out["%s_caminit%d_dt" % (tag, uc)] = cam.translation.grad.numpy().copy()
    _save("objective", **out)


def gen_lbfgs():
    """Reference run_fitting + LBFGS('lbfgsls') trajectories on analytic objectives."""
    def rosen(x):
        return (100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2).sum()

    def quad(x):
        n = x.numel()
        a = torch.arange(1, n + 1, dtype=x.dtype) ** 2
        return 0.5 * (a * (x - 0.3) ** 2).sum() + 0.1 * torch.sin(3 * x).sum()
    out = {}
    for fname, fn in (("rosen", rosen), ("quad", quad)):
        for N in (2, 6, 10):
            for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
                rng = np.random.RandomState(N)
                x0 = rng.uniform(-1, 1, size=N)
                x = torch.tensor(x0, dtype=dt, requires_grad=True)
                opt, _ = ref.optim_factory.create_optimizer([x], optim_type="lbfgsls", lr=1.0, maxiters=30)
                trace = []

                def closure(stage=0):
                    opt.zero_grad()
                    l = fn(x)
                    l.backward()
                    trace.append(np.concatenate([[l.item()], x.detach().numpy().astype(np.float64)]))
                    return l
                mon = ref.fitting.FittingMonitor(maxiters=30, ftol=1e-9, gtol=1e-9)
                with mon:
                    res = mon.run_fitting(opt, closure, [x], None, 0, use_vposer=False)
                key = "%s_%d_%s" % (fname, N, tag)
                out[key + "_x0"] = x0
                out[key + "_trace"] = np.stack(trace)
                out[key + "_res"] = np.array(res)
                out[key + "_xf"] = x.detach().numpy().astype(np.float64)
    _save("lbfgs", **out)


def gen_euler():
    import joblib
    out = {}
    rng = np.random.RandomState(0)
    from scipy.spatial.transform import Rotation as Rot
    Rs = Rot.from_rotvec(rng.normal(size=(64, 3)) * 1.2).as_matrix().astype(np.float32)
    out["rand_R"] = Rs
    out["rand_euler"] = np.stack([ref.utils._compute_euler_from_matrix(torch.tensor(R)).numpy()[0] for R in Rs])
    for name in ("02_cropped", "18_cropped"):
        ex = np.load(os.path.join(ref_import.REF_ROOT, "demo/ExPose_results/%s.jpg/%s.jpg_params.npz" % (name, name)),
                     allow_pickle=True)
        px = joblib.load(os.path.join(ref_import.REF_ROOT, "demo/PIXIE_results/%s/%s_param.pkl" % (name, name)))
        ep = [ref.utils._compute_euler_from_matrix(torch.tensor(r)) for r in ex["body_pose"]]
        pp = [ref.utils._compute_euler_from_matrix(torch.tensor(r)) for r in px["body_pose"]]
        g = ref.utils._compute_euler_from_matrix(torch.tensor(ex["global_orient"]))
        out[name + "_expose_R"] = ex["body_pose"]; out[name + "_pixie_R"] = np.asarray(px["body_pose"])
        out[name + "_expose_gR"] = ex["global_orient"]
        out[name + "_combined_pose"] = torch.cat(ep[:19] + pp[19:]).reshape(-1).numpy()
        out[name + "_global"] = g.numpy().reshape(-1)
    _save("euler", **out)


def gen_tables():
    out = {}
    for fmt in ("coco25", "halpe", "coco_wholebody", "coco19"):
        for h in (0, 1):
            for f in (0, 1):
                for c in (0, 1):
                    out["%s_%d%d%d" % (fmt, h, f, c)] = ref.utils.smpl_to_annotation(
                        "smplx", use_hands=bool(h), use_face=bool(f), use_face_contour=bool(c), format=fmt)
    _save("tables", **out)


# ------------------------------------------------------------------------------------------
def _run_reference_fit(bm, cfg, keypoints, H_, W_, focal, jw, dtype, pixie=None, expose=None, body_pose_prior=None):
    img = np.zeros((H_, W_, 3), np.float32)
    cam = ref.camera.create_camera(focal_length_x=float(focal), focal_length_y=float(focal), dtype=dtype, **cfg)
    cam.rotation.requires_grad = False
    a = dict(cfg); a["focal_length"] = float(focal)
    mk = lambda t: ref.prior.create_prior(prior_type=t, dtype=dtype)
    fn = tempfile.mktemp(suffix=".pkl")
    stage = []
    orig = ref.fitting.FittingMonitor.run_fitting

    def rf(self, *aa, **kk):
        r = orig(self, *aa, **kk)
        stage.append((r, self.steps))
        return r
    ref.fitting.FittingMonitor.run_fitting = rf
    uf, uh = cfg["use_face"], cfg["use_hands"]
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            ref.fit_single_frame.fit_single_frame(
                img, keypoints, body_model=bm, camera=cam,
                joint_weights=torch.tensor(jw, dtype=dtype).unsqueeze(0), dtype=dtype,
                shape_prior=mk("l2"), expr_prior=mk("l2") if uf else None,
                body_pose_prior=body_pose_prior if body_pose_prior is not None else mk(cfg["body_prior_type"]),
                left_hand_prior=mk("l2") if uh else None, right_hand_prior=mk("l2") if uh else None,
                jaw_prior=mk("l2") if uf else None, angle_prior=mk("angle"),
                result_fn=fn, pixie_results=pixie, expose_results=expose, pare_results=None, **a)
    finally:
        ref.fitting.FittingMonitor.run_fitting = orig
    res = pickle.load(open(fn, "rb"))
    os.remove(fn)
    losses = np.array([s[0] for s in stage], np.float64)
    steps = np.array([s[1] for s in stage])
    evals = np.diff(np.concatenate([[0], steps]))
    return res, losses, evals


class _FakeRegression(dict):
    """Stands in for the ExPose .npz: rotation matrices whose xyz-euler triples are the wanted prior."""


def gen_e2e():
    """Reference fit_single_frame on 2 well-posed synthetic frames (body-only, combined-cfg
    3-stage schedule, guess_init camera), fp32 and fp64."""
    import helpers as H
    from smplifyx_amd import synthetic
    model = synthetic.make_synthetic_model(0)
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False, use_cuda=False)
    cfg["use_camera_prior"] = False
    K = len(H.joint_map_for(cfg))
    frames = synthetic.make_frames(2, H.oracle_joints_fn(model, cfg), K, focal=5000.0)
    out = dict(keypoints=frames["keypoints"], reg_pose=frames["reg_pose"], reg_global=frames["reg_global"])
    from scipy.spatial.transform import Rotation as Rot
    for i in range(2):
        # regressor outputs whose euler conversion gives exactly the synthetic prior
        bp = Rot.from_euler("XYZ", frames["reg_pose"][i].reshape(21, 3).astype(np.float64)).as_matrix().astype(np.float32)
        go = Rot.from_euler("XYZ", frames["reg_global"][i].astype(np.float64)[None]).as_matrix().astype(np.float32)
        expose = {"body_pose": bp, "global_orient": go}
        c = dict(cfg); c["regression_prior"] = "ExPose"
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            bm = H.oracle_model(model, cfg, dtype)
            res, losses, evals = _run_reference_fit(bm, c, frames["keypoints"][i:i + 1], frames["H"], frames["W"],
                                                    frames["focal"], H.base_joint_weights(cfg, K), dtype, expose=expose)
            out["f%d_%s_losses" % (i, tag)] = losses
            out["f%d_%s_evals" % (i, tag)] = evals
            for k in ("camera_translation", "global_orient", "betas", "body_pose"):
                out["f%d_%s_%s" % (i, tag, k)] = np.asarray(res[k], np.float64)
            print("e2e frame", i, tag, losses, evals)
    _save("e2e_synth", **out)


def gen_e2e_vposer():
    """BASELINE config 3: full SMPL-X (hands + face + contour, K=135), VPoser decode in the loop,
    5-stage schedule of cfg_files/fit_smplx_smplifyx.yaml, zero-latent init, guess_init camera:
    the reference's fit_single_frame driving oracle SMPLXRef + VPoserRef (synthetic weights)."""
    import helpers as H
    from smplifyx_amd import synthetic
    from oracle.vposer import VPoserRef
    model = synthetic.make_synthetic_model(0)
    cfg = H.load_cfg("fit_smplx_smplifyx.yaml", use_cuda=False)
    K = len(H.joint_map_for(cfg))
    frames = synthetic.make_frames(1, H.oracle_joints_fn(model, cfg), K, focal=5000.0)
    out = dict(keypoints=frames["keypoints"])
    vpw = synthetic.make_synthetic_vposer(0)
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        vp = VPoserRef(vpw, dtype)
        ref.fit_single_frame.load_vposer = lambda ckpt, vp_model="snapshot": (vp, None)
        bm = H.oracle_model(model, cfg, dtype)
        res, losses, evals = _run_reference_fit(bm, cfg, frames["keypoints"][:1], frames["H"], frames["W"],
                                                frames["focal"], H.base_joint_weights(cfg, K), dtype)
        out["f0_%s_losses" % tag] = losses
        out["f0_%s_evals" % tag] = evals
        for k in ("camera_translation", "global_orient", "betas", "body_pose", "expression", "jaw_pose"):
            out["f0_%s_%s" % (tag, k)] = np.asarray(res[k], np.float64)
        print("e2e vposer", tag, losses, evals)
    _save("e2e_vposer", **out)


def _vposer_task(task):
    """One reference fit of frame i under BASELINE config 3 (worker of gen_e2e_vposer_set)."""
    i, tag = task
    import helpers as H
    from smplifyx_amd import synthetic
    from oracle.vposer import VPoserRef
    dtype = torch.float32 if tag == "f32" else torch.float64
    model = synthetic.make_synthetic_model(0)
    cfg = H.load_cfg("fit_smplx_smplifyx.yaml", use_cuda=False)
    K = len(H.joint_map_for(cfg))
    frames = synthetic.make_frames(1, H.oracle_joints_fn(model, cfg), K, start=i, focal=5000.0)
    vp = VPoserRef(synthetic.make_synthetic_vposer(0), dtype)
    ref.fit_single_frame.load_vposer = lambda ckpt, vp_model="snapshot": (vp, None)
    bm = H.oracle_model(model, cfg, dtype)
    res, losses, evals = _run_reference_fit(bm, cfg, frames["keypoints"], frames["H"], frames["W"], frames["focal"],
                                            H.base_joint_weights(cfg, K), dtype)
    print("e2e vposer set frame", i, tag, losses, evals, flush=True)
    return i, tag, frames["keypoints"][0], losses, evals


def gen_e2e_vposer_set():
    """A SET of reference fits under BASELINE config 3 (full SMPL-X K = 135, VPoser decode in the loop, 5 stages of
    cfg_files/fit_smplx_smplifyx.yaml, zero-latent start) for a distributional comparison: frames 0..N-1
    (SFX_GOLDEN_VPOSER_FRAMES, default 16), fp32 and fp64."""
    import multiprocessing as mp
    n = int(os.environ.get("SFX_GOLDEN_VPOSER_FRAMES", "16"))
    tasks = [(i, tag) for i in range(n) for tag in ("f32", "f64")]
    with mp.get_context("fork").Pool(int(os.environ.get("SFX_GOLDEN_WORKERS", "6"))) as pool:
        results = pool.map(_vposer_task, tasks, chunksize=1)
    kp = [None] * n
    out = {}
    for i, tag, k_, losses, evals in results:
        kp[i] = k_
        out["f%d_%s_losses" % (i, tag)] = losses
        out["f%d_%s_evals" % (i, tag)] = evals
    out.update(keypoints=np.stack(kp))
    _save("e2e_vposer_set", **out)


def gen_e2e_full():
    """Reference fit_single_frame with hands + face + face contour (K = 135 keypoints, all prior terms of
    SMPLifyLoss active) and a regression prior, use_vposer False: cfg_files/fit_smplx_combined_coco25.yaml
    (3 stages) and cfg_files/fit_smplx_combined_halpe.yaml (halpe keypoints K = 136, joint confidences
    used) without the interpenetration term, one synthetic frame each, fp32 and fp64."""
    import helpers as H
    from smplifyx_amd import synthetic
    from scipy.spatial.transform import Rotation as Rot
    model = synthetic.make_synthetic_model(0)
    out = {}
    for name, yaml_ in (("coco25", "fit_smplx_combined_coco25.yaml"), ("halpe", "fit_smplx_combined_halpe.yaml")):
        cfg = H.load_cfg(yaml_, use_cuda=False, interpenetration=False)
        cfg["use_camera_prior"] = False
        K = len(H.joint_map_for(cfg))
        frames = synthetic.make_frames(1, H.oracle_joints_fn(model, cfg), K, focal=5000.0)
        out[name + "_keypoints"] = frames["keypoints"]
        out[name + "_reg_pose"], out[name + "_reg_global"] = frames["reg_pose"], frames["reg_global"]
        bp = Rot.from_euler("XYZ", frames["reg_pose"][0].reshape(21, 3).astype(np.float64)).as_matrix().astype(np.float32)
        go = Rot.from_euler("XYZ", frames["reg_global"][0].astype(np.float64)[None]).as_matrix().astype(np.float32)
        c = dict(cfg); c["regression_prior"] = "ExPose"
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            bm = H.oracle_model(model, cfg, dtype)
            res, losses, evals = _run_reference_fit(bm, c, frames["keypoints"][:1], frames["H"], frames["W"], frames["focal"],
                                                    H.base_joint_weights(cfg, K), dtype,
                                                    expose={"body_pose": bp, "global_orient": go})
            out["%s_%s_losses" % (name, tag)] = losses
            out["%s_%s_evals" % (name, tag)] = evals
            for k in ("camera_translation", "global_orient", "betas", "body_pose", "left_hand_pose", "right_hand_pose",
                      "expression", "jaw_pose"):
                out["%s_%s_%s" % (name, tag, k)] = np.asarray(res[k], np.float64)
            print("e2e full", name, tag, losses, evals)
    _save("e2e_full", **out)


def _full_task(task):
    """One reference fit of frame i with hands + face + contour (cfg_files/fit_smplx_combined_halpe.yaml, K = 136,
    interpenetration off: the package is absent) in one precision (worker of gen_e2e_full_set)."""
    i, tag = task
    import helpers as H
    from smplifyx_amd import synthetic
    from scipy.spatial.transform import Rotation as Rot
    dtype = torch.float32 if tag == "f32" else torch.float64
    model = synthetic.make_synthetic_model(0)
    cfg = H.load_cfg("fit_smplx_combined_halpe.yaml", use_cuda=False, interpenetration=False)
    cfg["use_camera_prior"] = False
    K = len(H.joint_map_for(cfg))
    frames = synthetic.make_frames(1, H.oracle_joints_fn(model, cfg), K, start=i, focal=5000.0)
    bp = Rot.from_euler("XYZ", frames["reg_pose"][0].reshape(21, 3).astype(np.float64)).as_matrix().astype(np.float32)
    go = Rot.from_euler("XYZ", frames["reg_global"][0].astype(np.float64)[None]).as_matrix().astype(np.float32)
    c = dict(cfg); c["regression_prior"] = "ExPose"
    bm = H.oracle_model(model, cfg, dtype)
    res, losses, evals = _run_reference_fit(bm, c, frames["keypoints"], frames["H"], frames["W"], frames["focal"],
                                            H.base_joint_weights(cfg, K), dtype, expose={"body_pose": bp, "global_orient": go})
    print("e2e full set frame", i, tag, losses, evals, flush=True)
    return i, tag, frames["keypoints"][0], frames["reg_pose"][0], frames["reg_global"][0], losses, evals


def gen_e2e_full_set():
    """A SET of reference fits of the full model (halpe cfg, hands + face + contour, K = 136, every prior term, joint
    confidences; 3 stages) for a distributional comparison like e2e_bench: frames 0..N-1 (SFX_GOLDEN_FULL_FRAMES, default
    16), fp32 and fp64."""
    import multiprocessing as mp
    n = int(os.environ.get("SFX_GOLDEN_FULL_FRAMES", "16"))
    tasks = [(i, tag) for i in range(n) for tag in ("f32", "f64")]
    with mp.get_context("fork").Pool(int(os.environ.get("SFX_GOLDEN_WORKERS", "6"))) as pool:
        results = pool.map(_full_task, tasks, chunksize=1)
    kp = [None] * n; rp = [None] * n; rg = [None] * n
    out = {}
    for i, tag, k_, p_, g_, losses, evals in results:
        kp[i], rp[i], rg[i] = k_, p_, g_
        out["f%d_%s_losses" % (i, tag)] = losses
        out["f%d_%s_evals" % (i, tag)] = evals
    out.update(keypoints=np.stack(kp), reg_pose=np.stack(rp), reg_global=np.stack(rg))
    _save("e2e_full_set", **out)


def _cv2_rodrigues(x):
    """Functional stand-in for the one cv2 call on the fitting path (fit_single_frame.py:529-531; cv2 is
    not installed and is otherwise a MagicMock): rotation vector <-> matrix via scipy."""
    from scipy.spatial.transform import Rotation as Rot
    x = np.asarray(x, np.float64)
    if x.size == 3:
        return Rot.from_rotvec(x.reshape(3)).as_matrix(), None
    return Rot.from_matrix(x.reshape(3, 3)).as_rotvec().reshape(3, 1), None


def gen_e2e_side():
    """A side view (2-D shoulders closer than side_view_thsh): the reference fits the frame twice, from
    the camera-stage orientation and from that orientation turned by pi about y, and keeps the fit with
    the lower final loss (fit_single_frame.py:461-463,527-551,662-667).  Frame 1 of e2e_synth with the
    left shoulder moved next to the right one; per-stage losses of BOTH passes and the kept result."""
    import types
    import helpers as H
    from smplifyx_amd import synthetic
    from scipy.spatial.transform import Rotation as Rot
    g = np.load(os.path.join(GOLD, "e2e_synth.npz"))
    model = synthetic.make_synthetic_model(0)
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False, use_cuda=False)
    cfg["use_camera_prior"] = False
    kp = g["keypoints"][1:2].copy()
    kp[0, 5, :2] = kp[0, 2, :2] + 3.0
    out = dict(keypoints=kp, reg_pose=g["reg_pose"][1:2], reg_global=g["reg_global"][1:2])
    bp = Rot.from_euler("XYZ", g["reg_pose"][1].reshape(21, 3).astype(np.float64)).as_matrix().astype(np.float32)
    go = Rot.from_euler("XYZ", g["reg_global"][1].astype(np.float64)[None]).as_matrix().astype(np.float32)
    c = dict(cfg); c["regression_prior"] = "ExPose"
    old_cv2 = ref.fit_single_frame.cv2
    ref.fit_single_frame.cv2 = types.SimpleNamespace(Rodrigues=_cv2_rodrigues)
    try:
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            bm = H.oracle_model(model, cfg, dtype)
            res, losses, evals = _run_reference_fit(bm, c, kp, 600, 800, 5000.0, H.base_joint_weights(cfg, 25), dtype,
                                                    expose={"body_pose": bp, "global_orient": go})
            assert len(losses) == 7, losses                 # camera + 3 stages x 2 orientations
            out[tag + "_losses"], out[tag + "_evals"] = losses, evals
            for k in ("camera_translation", "global_orient", "betas", "body_pose"):
                out["%s_%s" % (tag, k)] = np.asarray(res[k], np.float64)
            print("e2e side", tag, losses, evals)
    finally:
        ref.fit_single_frame.cv2 = old_cv2
    _save("e2e_side", **out)


def _bench_task(task):
    """One reference fit of benchmark frame i in one precision (worker of gen_e2e_bench): per-stage results plus
    the per-LBFGS.step trace (step-entry loss, cumulative closure evaluations, cumulative inner iterations)."""
    i, tag = task
    import helpers as H
    from smplifyx_amd import synthetic
    from scipy.spatial.transform import Rotation as Rot
    dtype = torch.float32 if tag == "f32" else torch.float64
    model = synthetic.make_synthetic_model(0)
    cfg = H.load_cfg("fit_smplx_smplifyx.yaml", use_hands=False, use_face=False, use_vposer=False, use_cuda=False)
    cfg["use_camera_prior"] = False
    K = len(H.joint_map_for(cfg))
    frames = synthetic.make_frames(1, H.oracle_joints_fn(model, cfg), K, start=i, focal=5000.0,
                                   min_camera_keypoints=None if _BENCH_RAW else 3,      # (3 = bench.MIN_CAMERA_KEYPOINTS)
                                   camera_keypoints=cfg.get("init_joints_idxs", (9, 12, 2, 5)))
    c = dict(cfg); c["regression_prior"] = "ExPose"
    bp = Rot.from_euler("XYZ", frames["reg_pose"][0].reshape(21, 3).astype(np.float64)).as_matrix().astype(np.float32)
    go = Rot.from_euler("XYZ", frames["reg_global"][0].astype(np.float64)[None]).as_matrix().astype(np.float32)
    trace = []
    orig_step = ref.lbfgs_ls.LBFGS.step

    def step(self, closure):
        r = orig_step(self, closure)
        st = self.state[self._params[0]]
        trace.append((float(r), st["func_evals"], st["n_iter"], id(self)))
        return r
    ref.lbfgs_ls.LBFGS.step = step
    try:
        bm = H.oracle_model(model, cfg, dtype)
        res, losses, evals = _run_reference_fit(bm, c, frames["keypoints"], frames["H"], frames["W"], frames["focal"],
                                                H.base_joint_weights(cfg, K), dtype, expose={"body_pose": bp, "global_orient": go})
    finally:
        ref.lbfgs_ls.LBFGS.step = orig_step
    ids = []
    for t in trace:
        if not ids or ids[-1] != t[3]:
            ids.append(t[3])
    stage_of = np.array([ids.index(t[3]) for t in trace], np.int32)      # one optimiser object per stage
    out = {"losses": losses, "evals": evals,
           "step_loss": np.array([t[0] for t in trace], np.float64),
           "step_evals": np.array([t[1] for t in trace], np.int32),
           "step_iters": np.array([t[2] for t in trace], np.int32),
           "step_stage": stage_of}
    for k in ("camera_translation", "global_orient", "betas", "body_pose", "left_hand_pose", "right_hand_pose",
              "jaw_pose", "leye_pose", "reye_pose", "expression"):
        out[k] = np.asarray(res[k], np.float64)
    print("e2e bench frame", i, tag, losses, evals, flush=True)
    return i, tag, frames["keypoints"][0], frames["reg_pose"][0], frames["reg_global"][0], out


def gen_e2e_bench():
    """The benchmark's own configuration (bench.py build_cfg('body'): cfg_files/fit_smplx_smplifyx.yaml
    weights, 5 body stages, body-only keypoints, use_vposer False + regression prior) through the
    reference: frames 0..N-1 of the benchmark's synthetic sequence (N = SFX_GOLDEN_BENCH_FRAMES, default 32),
    fp32 and fp64, one process per (frame, precision).  bench.py's reference_parity leg and
    test_benchmark_configuration_matches_reference compare the DISTRIBUTION of final losses against these."""
    import multiprocessing as mp
    n = int(os.environ.get("SFX_GOLDEN_BENCH_FRAMES", "32"))
    out = {}
    kp = [None] * n; rp = [None] * n; rg = [None] * n
    have = 0
    old_path = os.path.join(GOLD, "e2e_bench.npz")
    if os.path.exists(old_path) and os.environ.get("SFX_GOLDEN_BENCH_EXTEND") == "1":      # keep the fits already made
        g = np.load(old_path)
        have = min(n, g["keypoints"].shape[0])
        for k in g.files:
            if k.startswith("f") and int(k[1:].split("_")[0]) < have:
                out[k] = g[k]
        for i in range(have):
            kp[i], rp[i], rg[i] = g["keypoints"][i], g["reg_pose"][i], g["reg_global"][i]
    redo = [int(x) for x in os.environ.get("SFX_GOLDEN_BENCH_REDO", "").split(",") if x.strip()]     # frames to fit again
    for i in redo:
        for k in [k for k in out if k.startswith("f%d_" % i)]:
            del out[k]
    tasks = [(i, tag) for i in list(range(have, n)) + [r for r in redo if r < have] for tag in ("f32", "f64")]
    with mp.get_context("fork").Pool(int(os.environ.get("SFX_GOLDEN_WORKERS", "6"))) as pool:
        results = pool.map(_bench_task, tasks, chunksize=1)
    for i, tag, k_, p_, g_, o in results:
        kp[i], rp[i], rg[i] = k_, p_, g_
        for key, v in o.items():
            out["f%d_%s_%s" % (i, tag, key)] = v
    out.update(keypoints=np.stack(kp), reg_pose=np.stack(rp), reg_global=np.stack(rg))
    _save("e2e_bench", **out)


def gen_demo():
    """BASELINE config 1: the two demo/ frames, body-only, combined regression prior +
    camera prior (cfg_files/fit_smplx_combined_coco25.yaml), reference fit in fp32."""
    import joblib
    from PIL import Image
    import helpers as H
    from smplifyx_amd import synthetic
    model = synthetic.make_synthetic_model(0)
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False, use_cuda=False)
    K = len(H.joint_map_for(cfg))
    out = {}
    for name in ("02_cropped", "18_cropped"):
        R = ref_import.REF_ROOT
        kp = json.load(open(os.path.join(R, "demo/keypoints/%s_blended.json" % name)))["people"][0]
        keypoints = np.array(kp["pose_keypoints_2d"], np.float32).reshape(1, -1, 3)
        expose = np.load(os.path.join(R, "demo/ExPose_results/%s.jpg/%s.jpg_params.npz" % (name, name)), allow_pickle=True)
        pixie = joblib.load(os.path.join(R, "demo/PIXIE_results/%s/%s_param.pkl" % (name, name)))
        W_, H_ = Image.open(os.path.join(R, "demo/images/%s.jpg" % name)).size
        focal = (W_ ** 2 + H_ ** 2) ** 0.5
        bm = H.oracle_model(model, cfg, torch.float32)
        res, losses, evals = _run_reference_fit(bm, cfg, keypoints, H_, W_, focal, H.base_joint_weights(cfg, K),
                                                torch.float32, pixie=pixie, expose=expose)
        transl = np.array(expose["transl"], np.float64); transl[-1] /= (5000 / focal)
        out.update({name + "_" + k: v for k, v in dict(
            keypoints=keypoints, HW=np.array([H_, W_]), focal=np.array(focal), losses=losses, evals=evals,
            cam_prior_t=transl, cam_prior_center=np.asarray(expose["center"], np.float64),
            camera_translation=res["camera_translation"], global_orient=res["global_orient"], betas=res["betas"],
            body_pose=res["body_pose"]).items()})
        print("demo", name, losses, evals)
    _save("demo_config1", **out)


def _reference_gmm(gmm, dtype):
    """The reference's MaxMixturePrior on a synthetic mixture written as gmm_08.pkl into a temp folder."""
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "gmm_08.pkl"), "wb") as fh:
        pickle.dump(gmm, fh)
    with contextlib.redirect_stdout(io.StringIO()):
        return ref.prior.MaxMixturePrior(prior_folder=d, num_gaussians=8, dtype=dtype)


def gen_gmm():
    """prior.py:100-231 on smplifyx_amd.synthetic.make_synthetic_gmm(0) (8 components, 63-D):
    buffers, get_mean, values and autograd gradients at random poses (fp32 / fp64), and one whole
    fit_single_frame run with body_prior_type 'gmm' (use_vposer False, no regression prior: the body
    pose starts from the mixture's mean, fit_single_frame.py:250-252)."""
    import helpers as H
    from smplifyx_amd import synthetic
    gmm = synthetic.make_synthetic_gmm(0)
    out = dict(means=gmm["means"], covars=gmm["covars"], weights=gmm["weights"])
    rng = np.random.RandomState(77)
    base = gmm["means"][rng.randint(0, 8, size=24)]
    poses = base + rng.normal(size=base.shape) * np.repeat([0.02, 0.1, 0.3], 8)[:, None]
    out["poses"] = poses
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        pr = _reference_gmm(gmm, dtype)
        x = torch.tensor(poses, dtype=dtype, requires_grad=True)
        val = pr(x, None)
        val.sum().backward()
        out["val_" + tag] = val.detach().numpy().astype(np.float64)
        out["grad_" + tag] = x.grad.numpy().astype(np.float64)
        out["mean_" + tag] = pr.get_mean().numpy().astype(np.float64)
        out["nll_weights_" + tag] = pr.nll_weights.numpy().astype(np.float64)
        out["precisions_" + tag] = pr.precisions.numpy().astype(np.float64)
    # e2e: body-only 3-stage schedule of the combined cfg, no regression prior, use_vposer False
    model = synthetic.make_synthetic_model(0)
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False, use_cuda=False,
                     use_vposer=False, body_prior_type="gmm")
    cfg["use_camera_prior"] = False
    cfg["regression_prior"] = None
    K = len(H.joint_map_for(cfg))
    frames = synthetic.make_frames(1, H.oracle_joints_fn(model, cfg), K, focal=5000.0)
    out["keypoints"] = frames["keypoints"]
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        bm = H.oracle_model(model, cfg, dtype)
        res, losses, evals = _run_reference_fit(bm, cfg, frames["keypoints"][:1], frames["H"], frames["W"], frames["focal"],
                                                H.base_joint_weights(cfg, K), dtype, body_pose_prior=_reference_gmm(gmm, dtype))
        out["e2e_%s_losses" % tag] = losses
        out["e2e_%s_evals" % tag] = evals
        for k in ("camera_translation", "global_orient", "betas", "body_pose"):
            out["e2e_%s_%s" % (tag, k)] = np.asarray(res[k], np.float64)
        print("gmm e2e", tag, losses, evals)
    _save("gmm", **out)


def gen_gmm_unmerged():
    """MaxMixturePrior(use_merged=False) (prior.py:203-231: the per-component form) of the reference on the same
    synthetic mixture and poses as gen_gmm: values and autograd gradients, one pose at a time (the reference's indexing
    `nll_weights[:, min_idx]` is only meaningful for batch size 1, the size the reference runs at), fp32 / fp64."""
    from smplifyx_amd import synthetic
    gmm = synthetic.make_synthetic_gmm(0)
    g = np.load(os.path.join(GOLD, "gmm.npz"))
    poses = g["poses"]
    out = dict(poses=poses)
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        d = tempfile.mkdtemp()
        with open(os.path.join(d, "gmm_08.pkl"), "wb") as fh:
            pickle.dump(gmm, fh)
        with contextlib.redirect_stdout(io.StringIO()):
            pr = ref.prior.MaxMixturePrior(prior_folder=d, num_gaussians=8, dtype=dtype, use_merged=False)
        vals, grads = [], []
        for i in range(poses.shape[0]):
            x = torch.tensor(poses[i:i + 1], dtype=dtype, requires_grad=True)
            v = pr(x, None)
            v.sum().backward()
            vals.append(float(v.reshape(-1)[0])); grads.append(x.grad.numpy()[0].astype(np.float64))
        out["val_" + tag] = np.array(vals, np.float64)
        out["grad_" + tag] = np.stack(grads)
    _save("gmm_unmerged", **out)


def gen_e2e_bench_raw_delta():
    """The benchmark's RAW SURVEY 8(d) sequence (no minimum of camera-initialisation keypoints) differs from the sequence
    e2e_bench.npz holds in the frames that lose two of those keypoints (23 and 51 of the first 64): their raw keypoints
    and reference fits (fp32 / fp64), so that bench.py can score its raw headline against the reference as well."""
    import helpers as H
    from smplifyx_amd import synthetic
    g = np.load(os.path.join(GOLD, "e2e_bench.npz"))
    n = g["keypoints"].shape[0]
    model = synthetic.make_synthetic_model(0)
    cfg = H.load_cfg("fit_smplx_smplifyx.yaml", use_hands=False, use_face=False, use_vposer=False, use_cuda=False)
    K = len(H.joint_map_for(cfg))
    raw = synthetic.make_frames(n, H.oracle_joints_fn(model, cfg), K, focal=5000.0)
    differ = [i for i in range(n) if np.abs(raw["keypoints"][i] - g["keypoints"][i]).max() > 0]
    out = {"frames": np.array(differ)}
    import multiprocessing as mp
    global _BENCH_RAW
    _BENCH_RAW = True
    with mp.get_context("fork").Pool(4) as pool:
        results = pool.map(_bench_task, [(i, tag) for i in differ for tag in ("f32", "f64")], chunksize=1)
    kp = {}; rp = {}; rg = {}
    for i, tag, k_, p_, g_, o in results:
        kp[i], rp[i], rg[i] = k_, p_, g_
        for key, v in o.items():
            out["f%d_%s_%s" % (i, tag, key)] = v
    out.update(keypoints=np.stack([kp[i] for i in differ]), reg_pose=np.stack([rp[i] for i in differ]),
               reg_global=np.stack([rg[i] for i in differ]))
    _save("e2e_bench_raw_delta", **out)


_BENCH_RAW = False


def gen_eval():
    """utils.ProcrustesAlignment / ScaleAlignment / PelvisAlignment(+MPJPE) / mpjpe / v2v of the
    reference on seeded point sets (fscore thresholds None: open3d is absent)."""
    rng = np.random.RandomState(11)
    out = {}
    U = ref.utils
    for tag, n in (("j", 14), ("v", 400)):
        gt = rng.normal(size=(n, 3))
        A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        est = 1.3 * gt @ A.T + rng.normal(size=(1, 3)) + 0.05 * rng.normal(size=(n, 3))
        out[tag + "_gt"], out[tag + "_est"] = gt, est
        out[tag + "_procrustes"] = U.ProcrustesAlignment()(est, gt)
        out[tag + "_procrustes_cols"] = U.ProcrustesAlignment()(est.T.copy(), gt.T.copy())
        out[tag + "_scale"] = U.ScaleAlignment()(est, gt)
        pa = U.PelvisAlignment()(gt, est)
        out[tag + "_pelvis_gt"], out[tag + "_pelvis_est"] = pa
        out[tag + "_mpjpe"] = U.mpjpe(est, gt)
        out[tag + "_v2v"] = U.vertex_to_vertex_error(est, gt)
        out[tag + "_pelvis_mpjpe"] = U.PelvisAlignmentMPJPE()(est, gt)["point"]
        out[tag + "_procrustes_mpjpe"] = U.ProcrustesAlignmentMPJPE()(est, gt)["point"]
    _save("evaluation", **out)


def gen_parser():
    """data_parser.read_keypoints / dataset weights / shoulders of the reference on the demo
    keypoint files.  The fixture keeps numbers only: the json fields as arrays (tests rebuild a
    json from them) and what the reference returns for every flag combination."""
    import json
    out = {}
    demo = os.path.join(ref_import.REF_ROOT, "demo")
    for name in ("02_cropped", "18_cropped"):
        fn = os.path.join(demo, "keypoints", name + "_blended.json")
        data = json.load(open(fn))
        out[name + "_n_people"] = np.array(len(data["people"]))
        for p, person in enumerate(data["people"]):
            for key in ("pose_keypoints_2d", "hand_left_keypoints_2d", "hand_right_keypoints_2d", "face_keypoints_2d"):
                out["%s_p%d_%s" % (name, p, key)] = np.asarray(person[key], np.float64)
        for hands in (0, 1):
            for face in (0, 1):
                for contour in (0, 1):
                    kt = ref.data_parser.read_keypoints(fn, use_hands=bool(hands), use_face=bool(face),
                                                         use_face_contour=bool(contour))
                    out["%s_kp_h%d_f%d_c%d" % (name, hands, face, contour)] = np.stack(kt.keypoints)
    for fmt, cls in (("coco25", ref.data_parser.COCO25), ("halpe", ref.data_parser.Halpe),
                     ("coco_wholebody", ref.data_parser.COCO_Wholebody)):
        for hands in (0, 1):
            for face in (0, 1):
                for contour in (0, 1):
                    ds = cls(demo, use_hands=bool(hands), use_face=bool(face), use_face_contour=bool(contour),
                             joints_to_ign=[1, 9, 12])
                    out["%s_jw_h%d_f%d_c%d" % (fmt, hands, face, contour)] = ds.get_joint_weights().numpy()
        ds = cls(demo)
        out[fmt + "_shoulders"] = np.array([ds.get_left_shoulder(), ds.get_right_shoulder()])
        out[fmt + "_n_items"] = np.array(len(ds))
    _save("parser", **out)


PEN_CACHE = os.path.join(ROOT, "gpurun_out", "pen_set_cache")       # (scratch: not committed, not shipped)
PEN_PRIOR_SEED = 1000            # bench.py --workload pen / tools/pen_collapse_probe.py: RandomState(1000 + rank) camera-prior noise


def _pen_frame_inputs(i, model, cfg):
    """Frame i of the `--workload pen` job: keypoints (oracle forward), regression prior, the synthetic "ExPose" camera prior
    (true translation + 5 cm noise, row i of RandomState(1000).normal(size=(n, 3)) -- the same row for every n > i)."""
    import helpers as H
    from smplifyx_amd import synthetic
    K = len(H.joint_map_for(cfg))
    frames = synthetic.make_frames(1, H.oracle_joints_fn(model, cfg), K, start=i, focal=5000.0)
    noise = np.random.RandomState(PEN_PRIOR_SEED).normal(size=(i + 1, 3))[i]
    cam_prior_t = (frames["cam_t"][0] + 0.05 * noise).astype(np.float32)
    return frames, cam_prior_t


def _pen_task(task):
    """One REAL-reference fit of frame i of the configs[4] job -- cfg_files/fit_smplx_combined_halpe.yaml verbatim (hands + face,
    K = 136, regression prior 'combined', camera prior, interpenetration: max_collisions 128, df_cone_height 1e-4,
    coll_loss_weights [0, 0.1, 1], the cfg's ign_part_pairs) on synthetic.make_topology_model(0) (the real SMPL-X faces and part
    table) -- in one precision, `term` on or off.  The reference's fit_single_frame runs as it stands (python -O: the two CUDA
    asserts at fit_single_frame.py:305-308 are stripped): it imports BVH / DistanceFieldPenetrationLoss / FilterFaces
    (:301-303 -> oracle/mesh_intersection_cpu.py), reads the part table from `part_segm_fn` (:316-324), and SMPLifyLoss.forward
    evaluates fitting.py:437-455."""
    i, tag, term = task
    import types
    import helpers as H
    from smplifyx_amd import synthetic
    from scipy.spatial.transform import Rotation as Rot
    cache = os.path.join(PEN_CACHE, "f%d_%s_%d.npz" % (i, tag, term))
    if os.path.exists(cache):            # (a fit takes minutes: every finished task is kept on disk until the set is assembled)
        z = np.load(cache)
        return i, tag, term, z["in_keypoints"], z["in_reg_pose"], z["in_reg_global"], z["in_cam_prior_t"], \
            {k: z[k] for k in z.files if not k.startswith("in_")}
    dtype = torch.float32 if tag == "f32" else torch.float64
    model = synthetic.make_topology_model(0)
    parts = synthetic.topology_parts()
    cfg = H.load_cfg("fit_smplx_combined_halpe.yaml", use_cuda=False, interpenetration=bool(term))
    assert cfg["regression_prior"] == "combined" and cfg["use_camera_prior"] and cfg["max_collisions"] == 128
    frames, cam_prior_t = _pen_frame_inputs(i, model, cfg)
    K = frames["keypoints"].shape[1]
    M = ref_import.install_mesh_intersection(np.asarray(model["f"]), (parts["segm"], parts["parents"], cfg["ign_part_pairs"]))
    made = {}
    orig_bvh_init = M.BVH.__init__

    def bvh_init(self, *a, **k):
        orig_bvh_init(self, *a, **k)
        made["bvh"] = self
    M.BVH.__init__ = bvh_init
    segm_fn = tempfile.mktemp(suffix="_parts.pkl")
    with open(segm_fn, "wb") as fh:
        pickle.dump({"segm": np.asarray(parts["segm"]), "parents": np.asarray(parts["parents"])}, fh)
    bp = Rot.from_euler("XYZ", frames["reg_pose"][0].reshape(21, 3).astype(np.float64)).as_matrix().astype(np.float32)
    go = Rot.from_euler("XYZ", frames["reg_global"][0].astype(np.float64)[None]).as_matrix().astype(np.float32)
    # regression_prior 'combined' (fit_single_frame.py:209-236): ExPose's joints [:19] + PIXIE's [19:], ExPose's global orientation
    # and camera (center, transl: :389-396; focal 5000 leaves transl as it is)
    expose = {"body_pose": bp, "global_orient": go, "center": np.array([frames["W"] * 0.5, frames["H"] * 0.5]),
              "transl": cam_prior_t.astype(np.float64).copy()}
    pixie = {"body_pose": bp, "global_pose": go}
    c = dict(cfg); c["part_segm_fn"] = segm_fn
    t0 = __import__("time").time()
    old_cv2 = ref.fit_single_frame.cv2
    ref.fit_single_frame.cv2 = types.SimpleNamespace(Rodrigues=_cv2_rodrigues)        # (side views: fit_single_frame.py:529-531)
    try:
        bm = H.oracle_model(model, cfg, dtype)
        res, losses, evals = _run_reference_fit(bm, c, frames["keypoints"], frames["H"], frames["W"], frames["focal"],
                                                H.base_joint_weights(cfg, K), dtype, pixie=pixie, expose=expose)
    finally:
        ref.fit_single_frame.cv2 = old_cv2
        M.BVH.__init__ = orig_bvh_init
        os.remove(segm_fn)
    # a side view is fitted from two orientations (fit_single_frame.py:527-551): camera + 3 stages x 2; the fit with the lower
    # final loss is the result (:662-667) -- `losses` / `evals` of THAT pass, the raw sequences beside them
    out = {"losses_all": losses, "evals_all": evals}
    n_st = 3
    if len(losses) == 1 + 2 * n_st:
        second = not (losses[n_st] < losses[2 * n_st])          # (:663: min over the two final losses, the first on a tie / NaN)
        sel = [0] + list(range(1 + n_st, 1 + 2 * n_st) if second else range(1, 1 + n_st))
        losses, evals = losses[sel], evals[sel]
        out["kept_orientation"] = np.array(int(second))
    else:
        out["kept_orientation"] = np.array(0)
    out.update({"losses": losses, "evals": evals})
    for k in ("camera_translation", "global_orient", "betas", "body_pose", "left_hand_pose", "right_hand_pose",
              "jaw_pose", "leye_pose", "reye_pose", "expression"):
        out[k] = np.asarray(res[k], np.float64)
    out["finite"] = np.array(all(np.isfinite(v).all() for v in out.values()))
    bvh = made.get("bvh")
    out["bvh_calls"] = np.array(bvh.calls if bvh is not None else 0)
    out["bvh_pairs_cut"] = np.array(bvh.pairs_cut if bvh is not None else 0)
    out["bvh_max_pairs"] = np.array(bvh.max_pairs if bvh is not None else 0)
    os.makedirs(PEN_CACHE, exist_ok=True)
    np.savez_compressed(cache, in_keypoints=frames["keypoints"][0], in_reg_pose=frames["reg_pose"][0], in_reg_global=frames["reg_global"][0],
                        in_cam_prior_t=cam_prior_t, **out)
    print("e2e pen frame", i, tag, "term" if term else "no term", out["losses_all"], out["evals_all"], "bvh calls %s cut %s max unordered pairs %s" %
          (out["bvh_calls"], out["bvh_pairs_cut"], out["bvh_max_pairs"]), "%.0f s" % (__import__("time").time() - t0), flush=True)
    return i, tag, term, frames["keypoints"][0], frames["reg_pose"][0], frames["reg_global"][0], cam_prior_t, out


def gen_e2e_pen_set():
    """BASELINE configs[4] through the REAL reference with the interpenetration term ON (see _pen_task): frames
    SFX_GOLDEN_PEN_FRAMES (comma list of indices of the `--workload pen` sequence; the default holds the frames the device path
    collapses on -- tools/pen_collapse_probe.py -- next to ordinary ones), fp32 and fp64 with the term, fp32 without it.
    Run as  python -O tools/make_goldens.py e2e_pen_set  (-O strips the reference's two CUDA asserts).  Minutes per fit."""
    import multiprocessing as mp
    if __debug__:
        raise SystemExit("e2e_pen_set: run under python -O (fit_single_frame.py:305-308 asserts use_cuda and a CUDA device)")
    idx = [int(x) for x in os.environ.get("SFX_GOLDEN_PEN_FRAMES", PEN_DEFAULT_FRAMES).split(",") if x.strip()]
    path = os.path.join(GOLD, "e2e_pen_set.npz")
    out = {}
    if os.path.exists(path) and os.environ.get("SFX_GOLDEN_PEN_EXTEND") == "1":           # keep the fits already made
        g = np.load(path)
        out = {k: g[k] for k in g.files}
        have = set(int(x) for x in out.get("frames", []))
        idx = sorted(have | set(idx))
    else:
        have = set()
    tasks = [(i, tag, term) for tag, term in (("f64", 1), ("f32", 1), ("f32", 0)) for i in idx if i not in have]       # (longest first)
    with mp.get_context("fork").Pool(int(os.environ.get("SFX_GOLDEN_WORKERS", "6"))) as pool:
        results = list(pool.imap_unordered(_pen_task, tasks, chunksize=1))
    for i, tag, term, k_, p_, g_, c_, o in results:
        out["f%d_keypoints" % i], out["f%d_reg_pose" % i], out["f%d_reg_global" % i], out["f%d_cam_prior_t" % i] = k_, p_, g_, c_
        for key, v in o.items():
            out["f%d_%s%s_%s" % (i, tag, "" if term else "_noterm", key)] = v
    out["frames"] = np.array(idx, np.int64)
    _save("e2e_pen_set", **out)


def gen_objective_pen():
    """fitting.py:437-455 at closure level: the REAL SMPLifyLoss.forward with interpenetration=True over the CPU stand-ins for the
    three mesh_intersection objects (oracle/mesh_intersection_cpu.py), fp64, on a small triangle soup (its own data: no licensed
    array) -- total and every gradient incl. d total / d vertices -- for the cases the lines distinguish: collision weight 0 (the
    `coll_loss_weight.item() > 0` gate: the term is not evaluated), 0.1 with colliding pairs, and 0.1 on a mesh whose candidate
    pairs the part filter removes entirely (the `collision_idxs.ge(0).sum() > 0` branch).  tests/test_oracle_standins.py holds the
    oracle's objective + penetration term to these numbers."""
    from collections import namedtuple
    rng = np.random.RandomState(7)
    F, K = 300, 25
    c = 0.6 * rng.rand(F, 3)
    verts0 = (c[:, None, :] + 0.08 * rng.randn(F, 3, 3)).reshape(-1, 3) + [0, 0, 9.0]
    faces = np.arange(F * 3).reshape(F, 3)
    segm = rng.randint(0, 3, F)
    parents = np.where(segm == 2, 1, -1)
    out = dict(faces=faces, segm=segm, parents=parents, verts=verts0)
    MO = namedtuple("MO", ["joints", "full_pose", "betas", "body_pose", "left_hand_pose", "right_hand_pose",
                           "expression", "jaw_pose", "vertices"])
    M = ref_import.install_mesh_intersection(faces, None)          # (the package's order: BVH caps, FilterFaces filters; cap 128 never binds here)
    mk = lambda *s_: torch.tensor(rng.normal(size=s_), dtype=torch.float64, requires_grad=True)
    joints = torch.tensor(rng.normal(size=(1, K, 3)) * 0.4 + [0, 0, 9.0], dtype=torch.float64, requires_grad=True)
    full_pose, betas, emb = mk(1, 165), mk(1, 10), mk(1, 63)
    reg = torch.tensor(rng.normal(size=(1, 63)), dtype=torch.float64)
    gt = torch.tensor(rng.normal(size=(1, K, 2)) * 60 + [400, 300], dtype=torch.float64)
    conf = torch.tensor(rng.uniform(size=(1, K)), dtype=torch.float64)
    jw = torch.tensor((rng.uniform(size=(1, K)) > 0.2).astype(np.float64))
    out.update(joints=joints.detach().numpy(), full_pose=full_pose.detach().numpy(), betas=betas.detach().numpy(),
               emb=emb.detach().numpy(), reg=reg.numpy(), gt=gt.numpy(), conf=conf.numpy(), jw=jw.numpy())
    pri = lambda t: ref.prior.create_prior(prior_type=t, dtype=torch.float64)
    for tag, cw, all_one_part in (("w0", 0.0, False), ("w01", 0.1, False), ("nopairs", 0.1, True)):
        sg = np.zeros(F, np.int64) if all_one_part else segm
        cam = ref.camera.create_camera(focal_length_x=5000.0, focal_length_y=5000.0, dtype=torch.float64,
                                       center=torch.tensor([[400.0, 300.0]], dtype=torch.float64))
        with torch.no_grad():
            cam.translation[:] = torch.tensor([[0.05, 0.1, 20.0]], dtype=torch.float64)
        search_tree = M.BVH(max_collisions=128)
        pen_distance = M.DistanceFieldPenetrationLoss(sigma=0.01, point2plane=False, vectorized=True, penalize_outside=True)
        filt = M.FilterFaces(faces_segm=sg, faces_parents=parents, ign_part_pairs=["0,1"] if not all_one_part else None)
        loss = ref.fitting.create_loss("smplify", rho=100, use_joints_conf=True, use_face=False, use_hands=False,
                                       body_pose_prior=pri("l2"), shape_prior=pri("l2"), angle_prior=pri("angle"),
                                       interpenetration=True, search_tree=search_tree, pen_distance=pen_distance,
                                       tri_filtering_module=filt, dtype=torch.float64, regression_pose=reg, num_stages=3)
        W = dict(data_weight=1000 / 600, body_pose_weight=300.0, shape_weight=50.0, bending_prior_weight=3.17 * 300.0,
                 coll_loss_weight=cw)
        loss.reset_loss_weights(W)
        for t in (joints, full_pose, betas, emb):
            t.grad = None
        vertices = torch.tensor(verts0[None], dtype=torch.float64, requires_grad=True)
        mo = MO(joints, full_pose, betas, emb, None, None, None, None, vertices)
        total = loss(mo, camera=cam, gt_joints=gt, joints_conf=conf, body_model_faces=torch.tensor(faces.reshape(-1)),
                     joint_weights=jw, stage=1, use_vposer=False, pose_embedding=emb)
        total.backward()
        g = lambda t: (t.grad.numpy().copy() if t.grad is not None else np.zeros(tuple(t.shape)))
        out.update({tag + "_" + k: v for k, v in dict(
            total=np.array(total.item()), coll_loss_weight=np.array(cw), segm=sg, bvh_calls=np.array(search_tree.calls),
            pen_calls=np.array(pen_distance.calls), d_joints=g(joints), d_full_pose=g(full_pose), d_betas=g(betas), d_emb=g(emb),
            d_vertices=g(vertices), d_cam_t=cam.translation.grad.numpy().copy()).items()})
        print("objective_pen", tag, total.item(), "bvh calls", search_tree.calls, "pen calls", pen_distance.calls,
              "|d verts|", np.abs(g(vertices)).sum())
    _save("objective_pen", **out)


def gen_cubic_overflow():
    """_cubic_interpolate (lbfgs_ls.py:11-36) where a trial point of the line search evaluates to ~1e29 (a unit step along an L-BFGS
    direction blown up by the interpenetration term: tools/pen_nan_probe.py, frame 92 of the configs[4] job): function values are
    Python floats, directional derivatives 0-d tensors of the model's dtype, so `d1 ** 2` overflows in fp32 and the result is NaN,
    which min(max(...)) passes on; in fp64 it is finite.  Rows: x1, f1, g1, x2, f2, g2 -> t in fp32 and fp64."""
    rows = np.array([[0.0, 9.093459e4, -7.7e4, 1.0, 2.782107e29, 5.0e29],
                     [0.0, 9.093459e4, -7.7e4, 1.0, 2.782107e29, 5.564215e29],
                     [0.0, 1.0e5, -1.0e3, 1.0, 1.0e19, 1.0e19],          # just below the overflow of d1 ** 2 in fp32
                     [0.0, 1.0e5, -1.0e3, 1.0, 1.0e20, 3.0e20],          # just above
                     [0.0, 1.0e5, -1.0e3, 2.0, 3.0e5, 4.0e5],            # an ordinary bracket
                     [1.0, 5.0, -2.0, 0.25, 4.0, 1.0]])
    out = {"rows": rows}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        r = []
        for x1, f1, g1, x2, f2, g2 in rows:
            v = ref.lbfgs_ls._cubic_interpolate(float(x1), float(f1), torch.tensor(g1, dtype=dt), float(x2), float(f2), torch.tensor(g2, dtype=dt))
            r.append(float(v))
        out[tag] = np.array(r, np.float64)
        print("cubic", tag, out[tag])
    _save("cubic_overflow", **out)


def gen_smplx_topology():
    """tests/golden/smplx_topology.npz: a LOCAL, uncommitted build product (SMPL-X licence) -- tools/make_topology.py."""
    import make_topology
    print("wrote", make_topology.build(force=True))


if __name__ == "__main__":
    todo = sys.argv[1:] or ["tables", "euler", "objective", "lbfgs", "e2e", "demo", "e2e_vposer"]
    for w in todo:
        {"objective": gen_objective, "lbfgs": gen_lbfgs, "euler": gen_euler, "tables": gen_tables,
         "e2e": gen_e2e, "demo": gen_demo, "e2e_vposer": gen_e2e_vposer, "parser": gen_parser, "eval": gen_eval,
         "gmm": gen_gmm, "e2e_full": gen_e2e_full, "e2e_full_set": gen_e2e_full_set, "e2e_vposer_set": gen_e2e_vposer_set, "e2e_side": gen_e2e_side, "e2e_bench": gen_e2e_bench,
         "gmm_unmerged": gen_gmm_unmerged, "e2e_pen_set": gen_e2e_pen_set, "objective_pen": gen_objective_pen, "cubic_overflow": gen_cubic_overflow, "e2e_bench_raw_delta": gen_e2e_bench_raw_delta, "smplx_topology": gen_smplx_topology}[w]()
