"""Batched frame driver: everything fit_single_frame() does around the hot loop
(smplifyx/fit_single_frame.py:209-294 initialisation, :358-411 camera prior / guess_init,
:447-612 camera stage + orientation x stage loop, :644-660 result dict) for MANY frames at
once.  All floating-point work of the loop itself happens in libsfx.so; this file only
prepares small per-frame arrays and collects results.
"""
import numpy as np

from . import engine

NUM_BODY_JOINTS = engine.NUM_BODY_JOINTS


def _rotvec_to_mat(r):
    r = np.asarray(r, np.float64).reshape(3)
    a = np.linalg.norm(r)
    if a < 1e-12:
        return np.eye(3)
    k = r / a
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def _mat_to_rotvec(R):
    """cv2.Rodrigues(3x3) semantics: axis * angle with angle in [0, pi]."""
    R = np.asarray(R, np.float64)
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    a = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(v) / 2
    if s < 1e-10:
        if c > 0:
            return np.zeros(3)
        d = np.sqrt(np.maximum((np.diag(R) + 1) / 2, 0))
        if R[0, 1] < 0: d[1] = -d[1]
        if R[0, 2] < 0: d[2] = -d[2]
        return d / np.linalg.norm(d) * a
    return v / (2 * s) * a


def flipped_orientation(body_orient):
    """The 180-degree-about-y candidate of fit_single_frame.py:528-535."""
    return _mat_to_rotvec(_rotvec_to_mat(body_orient) @ _rotvec_to_mat([0.0, np.pi, 0.0]))


def prepare_frames(cfg, keypoints, joint_weights, reg_pose=None, reg_global=None):
    """Per-frame arrays of fit_single_frame.py:276-294: low-confidence zeroing of the joint
    weights, trimmed camera-init joints, shoulder distance test.  keypoints [B,K,3]."""
    kp = np.asarray(keypoints, np.float32)
    B, K = kp.shape[:2]
    nb = NUM_BODY_JOINTS[cfg.get("format", "coco25")]
    thr = np.array([cfg.get("confidence_threshold", 0) or 0] * nb + [0] * 110, np.float64)[:K]
    low = kp[:, :, 2] < thr[None]
    jw = np.broadcast_to(np.asarray(joint_weights, np.float32).reshape(-1, K), (B, K)).copy()
    jw[low] = 0
    cmask = np.zeros((B, K), np.float32)
    for j in cfg.get("init_joints_idxs", (9, 12, 2, 5)):
        ok = (kp[:, j, 0] != 0) & (kp[:, j, 1] != 0) & ~low[:, j]
        cmask[ok, j] = 1
    ls, rs = cfg.get("left_shoulder_idx", 2), cfg.get("right_shoulder_idx", 5)
    sd = np.linalg.norm(kp[:, ls, :2] - kp[:, rs, :2], axis=1)
    both = sd < cfg.get("side_view_thsh", 25.)
    return dict(keypoints=kp, jw=jw, cmask=cmask, try_both=both)


def _make_batch(dm, cfg, keypoints, joint_weights, H, W, focal, reg_pose, reg_global, cam_prior_t,
                cam_prior_center, lbs_mode, reuse_entry_eval, body_pose_prior=None, slots=0):
    """FrameBatch with frames, parameters and the initial camera set the way
    fit_single_frame.py:209-294,358-411 prepares one frame."""
    if cfg.get("optim_type", "lbfgsls") != "lbfgsls":
        raise NotImplementedError("the batched device path runs optim_type 'lbfgsls' (every shipped cfg); %r is available "
                                  "through optimizers.create_optimizer + FittingMonitor.run_fitting (host-driven steps, "
                                  "HIP closure)" % cfg.get("optim_type"))
    prep = prepare_frames(cfg, keypoints, joint_weights)
    kp = prep["keypoints"]
    B, K = kp.shape[:2]
    use_vposer = bool(cfg.get("use_vposer", True))
    has_reg = reg_pose is not None
    gmm = None
    if not has_reg and not use_vposer:
        # fit_single_frame.py:250-252: the body pose starts from body_pose_prior.get_mean(), which only the
        # mixture prior has (body_prior_type 'gmm'); with 'l2' the reference crashes here
        gmm = body_pose_prior
        if gmm is None and cfg.get("body_prior_type") == "gmm":
            from . import prior as _prior
            gmm = _prior.create_prior("gmm", prior_folder=cfg.get("prior_folder", "prior"),
                                      num_gaussians=cfg.get("num_gaussians", 8))
        if gmm is None or not hasattr(gmm, "get_mean"):
            raise ValueError("use_vposer=False without a regression prior needs body_prior_type 'gmm' (the reference "
                             "crashes here: fit_single_frame.py:252 with body_prior_type 'l2')")
    Hh = np.broadcast_to(np.asarray(H, np.float32), (B,))
    Ww = np.broadcast_to(np.asarray(W, np.float32), (B,))
    f = np.broadcast_to(np.asarray(focal, np.float32), (B,))
    fb = engine.FrameBatch(dm, B, cfg, lbs_mode=lbs_mode, reuse_entry_eval=reuse_entry_eval,
                           has_regression_pose=has_reg, side_view=True, slots=slots)
    nemb = fb.nemb
    if gmm is not None:
        fb.set_gmm(gmm)
        emb0 = np.tile(gmm.get_mean().detach().cpu().numpy().astype(np.float32).reshape(1, nemb), (B, 1))
    else:
        emb0 = np.asarray(reg_pose, np.float32).reshape(B, nemb) if has_reg else np.zeros((B, nemb), np.float32)
    go0 = np.asarray(reg_global, np.float32).reshape(B, 3) if reg_global is not None else np.zeros((B, 3), np.float32)
    use_cam_prior = bool(cfg.get("use_camera_prior")) and has_reg and cam_prior_t is not None
    center = (np.asarray(cam_prior_center, np.float32).reshape(B, 2) if use_cam_prior
              else np.stack([Ww * 0.5, Hh * 0.5], 1))
    est = np.asarray(cam_prior_t, np.float32).reshape(B, 3)[:, 2] if use_cam_prior else None
    fb.set_frames(kp, prep["jw"], prep["cmask"], f, center, 1000.0 / Hh, est_tz=est)
    fb.set_params(regression_pose=emb0 if has_reg else None, global_orient=go0, pose_embedding=emb0,
                  cam_translation=(np.asarray(cam_prior_t, np.float32).reshape(B, 3) if use_cam_prior
                                   else np.zeros((B, 3), np.float32)))
    if not use_cam_prior:
        fb.guess_init(cfg.get("body_tri_idxs", [(5, 12), (2, 9)]))
    return fb, prep


def _collect(fb, prep, want_vertices):
    st = fb.stats()
    out = dict(fb.get_params())
    out.update(stage_loss=st["stage_loss"].copy(), stage_evals=st["stage_evals"].copy(),
               stage_ref_evals=st["stage_ref_evals"].copy(),
               n_orient=np.where(prep["try_both"] & (fb.n_stages > 0), 2, 1).astype(np.int32))
    out["final_loss"] = out["stage_loss"][:, -1].copy()
    if fb.cfg.get("interpenetration", False):
        out["pen_order_dependent"] = fb.penetration_flags()      # frames whose collision partners depended on arrival order somewhere
        global last_pen_launches
        last_pen_launches = fb.penetration_launches()            # kernel launches of one step of the term (counted on the captured graph)
    if want_vertices:
        v, j = fb.forward()
        out["vertices"], out["joints"] = v.cpu().numpy(), j.cpu().numpy()
    fb.close()
    return out


last_pen_launches = 0        # of the most recent fit with the interpenetration term (bench.py roofline_pen)
POOL_COLUMNS = 512          # GEMM columns of the pool a job larger than this is run through by default (slots=-1)


def auto_slots(n_frames):
    """Default size of the column pool: jobs of up to POOL_COLUMNS frames are resident (one column per frame), larger ones queue
    through POOL_COLUMNS columns -- the columns a straggler would leave idle at the end of a resident run are refilled
    (1 024 frames of the benchmark's generator: 710 frames/s resident, 830-840 through 512 columns, LAB_NOTES §4.4)."""
    return 0 if n_frames <= POOL_COLUMNS else POOL_COLUMNS


def predicted_cost(cfg, prep):
    """What is known about a frame's fitting time BEFORE the fit (the queue of a column pool is ordered by it, longest first):
    a side view is fitted from two orientations (fit_single_frame.py:527-551: twice the body stages), and a frame that has lost
    two or more of the four camera-initialisation keypoints (cfg init_joints_idxs) starts from an under-determined camera and
    takes 2-3 x the evaluations of a well-posed one (bench.py MIN_CAMERA_KEYPOINTS; the reference's own fp32 / fp64 runs end
    such frames in different basins)."""
    n_cam = prep["cmask"].sum(1)
    n_init = len(cfg.get("init_joints_idxs", (9, 12, 2, 5)))
    cost = np.where(n_cam < min(3, n_init), 2.5, 1.0) * np.where(prep["try_both"], 2.0, 1.0)
    return cost


def fit_frames(dm, cfg, keypoints, joint_weights, H, W, focal, reg_pose=None, reg_global=None,
               cam_prior_t=None, cam_prior_center=None, lbs_mode="dense", reuse_entry_eval=True,
               want_vertices=False, body_pose_prior=None, slots=0, order="auto"):
    """Fit B frames.  Arrays are [B, ...]; H, W, focal scalars or [B].  Returns a dict of
    [B, ...] arrays: the reference's result-pkl fields + per-stage losses / evaluation counts.

    The whole schedule -- camera stage, body stages, and for side views (2-D shoulder distance
    < side_view_thsh, fit_single_frame.py:461-463) the second fit from the orientation flipped
    by pi about y with the lower final loss kept (:527-551,662-667) -- runs on device; frames
    change stage independently.

    slots (dense mode): size of the GEMM column pool when there are more frames than that --
    the reference's loop over frames (main.py:207) as continuous batching: frames queue and take over the
    columns of frames that finish.  Results are unchanged.  slots = -1: auto_slots(B).
    order (pooled runs only): "auto" = the queue is ordered by predicted_cost, longest first (a straggler admitted last holds
    its column -- and the whole job -- long after the others are done); "given" = frame order.  Frames are independent and
    every per-frame sum has a fixed order, so the results do not depend on it (returned in the caller's order, bitwise equal)."""
    B_all = np.asarray(keypoints).shape[0]
    if slots is not None and slots < 0:
        slots = auto_slots(B_all) if lbs_mode == "dense" else 0
    perm = None
    if lbs_mode == "dense" and 0 < slots < B_all and order == "auto":
        cost = predicted_cost(cfg, prepare_frames(cfg, keypoints, joint_weights))
        perm = np.argsort(-cost, kind="stable")
        if np.array_equal(perm, np.arange(B_all)):
            perm = None
    if perm is not None:
        def pm(a):
            if a is None:
                return None
            a = np.asarray(a)
            return a[perm] if a.ndim > 0 and a.shape[0] == B_all and B_all > 1 else a
        jw_a = np.asarray(joint_weights)
        keypoints = np.asarray(keypoints)[perm]
        if jw_a.ndim == 2 and jw_a.shape[0] == B_all and B_all > 1:
            joint_weights = jw_a[perm]
        H, W, focal, reg_pose, reg_global = pm(H), pm(W), pm(focal), pm(reg_pose), pm(reg_global)
        cam_prior_t, cam_prior_center = pm(cam_prior_t), pm(cam_prior_center)
    fb, prep = _make_batch(dm, cfg, np.asarray(keypoints), np.asarray(joint_weights), H, W, focal, reg_pose, reg_global,
                           cam_prior_t, cam_prior_center, lbs_mode, reuse_entry_eval, body_pose_prior, slots)
    pen_on = bool(cfg.get("interpenetration", False))
    work0 = engine.pen_work_get() if pen_on else None
    fb.fit(first_stage=-1, last_stage=fb.n_stages - 1)
    res = _collect(fb, prep, want_vertices)
    if pen_on:      # diagnostics of the interpenetration term over this fit (device counters, engine.pen_work_get; per-frame flags)
        w1 = engine.pen_work_get()
        cut, over = w1["walks_cut"] - work0["walks_cut"], w1["lists_overflowed"] - work0["lists_overflowed"]
        bad = np.flatnonzero(res["pen_order_dependent"])
        # (triangles that met more than 2 x max_collisions partners -- `over` of them -- are not a reason to warn any more: their
        #  kept partners are derived from the grid again, the lowest ids like everywhere else; csrc/collide.hip pen_rewalk)
        if cut or len(bad):
            import warnings
            warnings.warn("interpenetration term: %d bucket walks were cut short during this fit (a mesh folded into a few grid "
                          "cells by a trial step; %d triangles met more than 2 x max_collisions partners): pairs beyond the cut "
                          "are missing and which ones depends on arrival order -- frames %s are not reproducible run to run "
                          "(result key 'pen_order_dependent')" % (cut, over, (bad if perm is None else np.sort(perm[bad])).tolist()[:32]), RuntimeWarning)
    if perm is not None:        # back to the caller's frame order
        inv = np.empty_like(perm)
        inv[perm] = np.arange(B_all)
        res = {k: (np.asarray(v)[inv] if np.ndim(v) > 0 and np.asarray(v).shape[0] == B_all else v) for k, v in res.items()}
    return res
