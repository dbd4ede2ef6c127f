"""Batched fitting engine: Python face of libsfx.so (the MI355X-native entry point).

`DeviceModel` uploads an SMPL-X model once per GPU; `FrameBatch` holds B independent
frames (the reference handles exactly one, fit_single_frame.py:119) and runs the
reference's per-frame schedule for all of them on device.  The drop-in modules
(`smplx`, `fitting`, `fit_single_frame`, ...) are thin adapters over these two classes.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from .synthetic import SMPLX_EXTRA_VERTEX_IDS

NUM_BODY_JOINTS = {"coco25": 25, "halpe": 26, "coco_wholebody": 23}
PARAM_NAMES = ("cam_translation", "global_orient", "betas", "left_hand_pose", "right_hand_pose",
               "expression", "jaw_pose", "leye_pose", "reye_pose", "pose_embedding")


def _jaw_weights(cfg, shape_weights):
    jw = cfg.get("jaw_pose_prior_weights")
    if jw is None:
        return [[x] * 3 for x in shape_weights]
    return [[float(v) for v in e.split(",")] if isinstance(e, str) else [float(v) for v in e] for e in jw]


def stage_weights_from_cfg(cfg):
    """Per-stage weights exactly as fit_single_frame.py:133-207,330-353 assembles them
    (defaults, zip-truncation to the shortest list)."""
    use_hands, use_face = cfg.get("use_hands", True), cfg.get("use_face", True)
    bpw = cfg.get("body_pose_prior_weights") or [4.04 * 1e2, 4.04 * 1e2, 57.4, 4.78]
    n = len(bpw)
    lists = {"data": cfg.get("data_weights") or [1] * n, "bpw": bpw,
             "shape": cfg.get("shape_weights") or [1e2, 5 * 1e1, 1e1, .5 * 1e1]}
    if use_face:
        lists["face"] = cfg.get("face_joints_weights") or [0.0, 0.0, 0.0, 1.0]
        lists["expr"] = cfg.get("expr_weights") or [1e2, 5 * 1e1, 1e1, .5 * 1e1]
        lists["jaw"] = _jaw_weights(cfg, lists["shape"])
    if use_hands:
        lists["hand"] = cfg.get("hand_joints_weights") or [0.0, 0.0, 0.0, 1.0]
        lists["hprior"] = cfg.get("hand_pose_prior_weights") or [1e2, 5 * 1e1, 1e1, .5 * 1e1]
    if cfg.get("interpenetration", False):
        lists["coll"] = cfg.get("coll_loss_weights") or [0.0] * n
    lists["go"] = cfg.get("global_orient_weights") or [20, 10, 7.5, 5, 5]
    for k in ("shape", "face", "expr", "jaw", "hand", "hprior", "coll"):
        if k in lists and len(lists[k]) != n:
            raise AssertionError("Number of Body pose prior weights does not match the number of %s weights" % k)
    ns = min(len(v) for v in lists.values())
    out = []
    for i in range(ns):
        w = capi.StageWeights()
        w.body_pose_weight = lists["bpw"][i]
        w.shape_weight = lists["shape"][i]
        w.hand_prior_weight = lists["hprior"][i] if use_hands else 0.0
        w.expr_prior_weight = lists["expr"][i] if use_face else 0.0
        jaw = lists["jaw"][i] if use_face else [0.0, 0.0, 0.0]
        for q in range(3):
            w.jaw_prior_weight[q] = jaw[q]
        w.hand_joint_weight = lists["hand"][i] if use_hands else 0.0
        w.face_joint_weight = lists["face"][i] if use_face else 0.0
        w.coll_loss_weight = lists["coll"][i] if "coll" in lists else 0.0
        w.bending_prior_weight = -1.0
        out.append(w)
    return out, n


class DeviceModel(object):
    """SMPL-X constants resident in HBM (sfx_model).  `model` is a dict with the .npz key
    set of SURVEY.md appendix A.1 (a real SMPLX_*.npz loaded with numpy, or
    synthetic.make_synthetic_model())."""

    def __init__(self, model, joint_map=None, num_betas=10, num_expression_coeffs=10,
                 num_pca_comps=12, flat_hand_mean=False, use_face_contour=True,
                 extra_vertex_ids=None, vposer=None, use_pca=True):
        """use_pca=False (cmd_parser.py:127, smplx.SMPLX): the hand pose parameters are the 45 axis-angle values
        themselves -- the same kernels with identity 'components'."""
        lib = capi.load()
        self.use_pca = bool(use_pca)
        if not self.use_pca:
            num_pca_comps = 45
        sd = np.asarray(model["shapedirs"])
        es = 300 if sd.shape[-1] >= 400 else 10
        sdirs = np.concatenate([sd[:, :, :num_betas], sd[:, :, es:es + num_expression_coeffs]], -1)
        V = int(np.asarray(model["v_template"]).shape[0])
        faces = capi.i32(np.asarray(model["f"]).astype(np.int64))
        parents = np.asarray(model["kintree_table"])[0].astype(np.int64).copy()
        parents[0] = -1
        lm = np.zeros(45) if flat_hand_mean else np.asarray(model["hands_meanl"])
        rm = np.zeros(45) if flat_hand_mean else np.asarray(model["hands_meanr"])
        pose_mean = np.concatenate([np.zeros(75), lm, rm])
        if extra_vertex_ids is None:
            extra_vertex_ids = model.get("extra_vertex_ids", SMPLX_EXTRA_VERTEX_IDS)
        n_lmk = int(np.asarray(model["lmk_faces_idx"]).shape[0])
        dyn_f = np.asarray(model["dynamic_lmk_faces_idx"])
        n_dyn = int(dyn_f.shape[1]) if use_face_contour else 0
        n_all = 55 + len(extra_vertex_ids) + n_lmk + n_dyn
        if joint_map is None:
            joint_map = np.arange(n_all)
        self.joint_map = np.asarray(joint_map).astype(np.int64)
        self.V, self.F, self.K = V, int(faces.shape[0]), int(len(self.joint_map))
        self.num_betas, self.num_expr, self.num_pca = num_betas, num_expression_coeffs, num_pca_comps
        self.faces = np.asarray(model["f"]).astype(np.int64)
        keep = dict(
            v_template=capi.f32(model["v_template"]), shapedirs=capi.f32(sdirs),
            posedirs=capi.f32(model["posedirs"]), J_regressor=capi.f32(model["J_regressor"]),
            lbs_weights=capi.f32(model["weights"]), parents=capi.i32(parents),
            hands_comp_l=capi.f32(np.asarray(model["hands_componentsl"])[:num_pca_comps] if self.use_pca else np.eye(45)),
            hands_comp_r=capi.f32(np.asarray(model["hands_componentsr"])[:num_pca_comps] if self.use_pca else np.eye(45)),
            pose_mean=capi.f32(pose_mean), faces=faces,
            extra=capi.i32(extra_vertex_ids), lmk_f=capi.i32(model["lmk_faces_idx"]),
            lmk_b=capi.f32(model["lmk_bary_coords"]), dyn_f=capi.i32(dyn_f),
            dyn_b=capi.f32(model["dynamic_lmk_bary_coords"]), jm=capi.i32(self.joint_map))
        d = capi.ModelDesc()
        d.V, d.F, d.J = V, self.F, 55
        d.num_betas, d.num_expr, d.num_pca = num_betas, num_expression_coeffs, num_pca_comps
        d.v_template = capi.fptr(keep["v_template"]); d.shapedirs = capi.fptr(keep["shapedirs"])
        d.posedirs = capi.fptr(keep["posedirs"]); d.J_regressor = capi.fptr(keep["J_regressor"])
        d.lbs_weights = capi.fptr(keep["lbs_weights"]); d.parents = capi.iptr(keep["parents"])
        d.hands_comp_l = capi.fptr(keep["hands_comp_l"]); d.hands_comp_r = capi.fptr(keep["hands_comp_r"])
        d.pose_mean = capi.fptr(keep["pose_mean"]); d.faces = capi.iptr(keep["faces"])
        d.n_extra = len(extra_vertex_ids); d.extra_vertex_ids = capi.iptr(keep["extra"])
        d.n_lmk = n_lmk; d.lmk_faces_idx = capi.iptr(keep["lmk_f"]); d.lmk_bary = capi.fptr(keep["lmk_b"])
        d.n_dyn_rows = int(dyn_f.shape[0]) if n_dyn else 0; d.n_dyn = n_dyn
        d.dyn_lmk_faces_idx = capi.iptr(keep["dyn_f"]); d.dyn_lmk_bary = capi.fptr(keep["dyn_b"])
        d.K = self.K; d.joint_map = capi.iptr(keep["jm"])
        h = C.c_void_p()
        capi.check(lib.sfx_model_create(C.byref(d), C.byref(h)))
        self._h = h
        self._lib = lib
        self.vposer_latent = 0
        self.vposer_weights = None
        if vposer is not None:
            self.set_vposer(vposer)

    def set_parts(self, segm, parents, ign_part_pairs=None):
        """Per-face part labels (smplx_parts_segm.pkl: 'segm', 'parents') and the cfg's
        ign_part_pairs (["9,16", ...]) for batches created with interpenetration=True
        (fit_single_frame.py:316-328)."""
        sg, pr = capi.i32(np.asarray(segm).astype(np.int64)), capi.i32(np.asarray(parents).astype(np.int64))
        assert sg.shape == (self.F,) and pr.shape == (self.F,), "one label per face"
        pairs = []
        for p in ign_part_pairs or []:
            a, b = (int(x) for x in str(p).split(",")) if isinstance(p, str) else p
            pairs.append((a, b))
        ign = capi.i32(np.asarray(pairs, np.int64).reshape(-1, 2))
        capi.check(self._lib.sfx_model_set_parts(self._h, capi.iptr(sg), capi.iptr(pr),
                                                 capi.iptr(ign) if pairs else None, len(pairs)))
        self.has_parts = True

    def clear_parts(self):
        """No part filter (a loss created without a FilterFaces module: the reference filters nothing then)."""
        capi.check(self._lib.sfx_model_set_parts(self._h, None, None, None, 0))
        self.has_parts = False

    def set_vposer(self, w):
        a = {k: capi.f32(v) for k, v in w.items()}
        latent, hidden = a["fc1_w"].shape[1], a["fc1_w"].shape[0]
        capi.check(self._lib.sfx_model_set_vposer(self._h, latent, hidden, capi.fptr(a["fc1_w"]), capi.fptr(a["fc1_b"]),
                                                  capi.fptr(a["fc2_w"]), capi.fptr(a["fc2_b"]),
                                                  capi.fptr(a["out_w"]), capi.fptr(a["out_b"])))
        self.vposer_latent = latent
        self.vposer_weights = a          # the encoder (if present) runs on the host: vposer.encode

    def lbs_forward(self, global_orient, body_pose, betas, expression, jaw_pose, leye_pose, reye_pose,
                    left_hand_pose, right_hand_pose, return_verts=True, return_full_pose=True, stream=None):
        """torch CUDA tensors in/out (containers only); one dense LBS launch sequence."""
        import torch
        B = global_orient.shape[0]
        dev = global_orient.device
        ptr = lambda t: C.c_void_p(t.contiguous().data_ptr()) if t is not None else None
        ins = [t.contiguous().float() if t is not None else None for t in
               (global_orient, body_pose, betas, expression, jaw_pose, leye_pose, reye_pose,
                left_hand_pose, right_hand_pose)]
        verts = torch.empty([B, self.V, 3], dtype=torch.float32, device=dev) if return_verts else None
        joints = torch.empty([B, self.K, 3], dtype=torch.float32, device=dev)
        fp = torch.empty([B, 165], dtype=torch.float32, device=dev) if return_full_pose else None
        s = C.c_void_p(stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        capi.check(self._lib.sfx_lbs_forward(self._h, B, *[ptr(t) for t in ins], ptr(verts), ptr(joints), ptr(fp), s))
        return verts, joints, fp

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sfx_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrameBatch(object):
    """B frames under one configuration (sfx_batch)."""

    def __init__(self, model, B, cfg, lbs_mode="dense", reuse_entry_eval=True, has_regression_pose=True,
                 stages=None, num_body_joints=None, side_view=False, slots=0):
        """`cfg` uses the reference's key names (cmd_parser).  `stages` (list of
        capi.StageWeights) overrides the schedule derived from cfg; `num_body_joints` overrides
        where the per-stage hand/face joint weights start (K = never: weights passed verbatim).
        `slots` (dense mode): GEMM columns when B is larger -- the other frames queue and are admitted as
        columns free up (continuous batching); 0 = one column per frame."""
        self.model, self.B, self.cfg = model, int(B), dict(cfg)
        self._lib = model._lib
        if stages is None:
            stages, _ = stage_weights_from_cfg(cfg)
        self.n_stages = len(stages)
        c = capi.BatchCfg()
        c.B = self.B
        c.n_stages = self.n_stages
        c.use_vposer = int(bool(cfg.get("use_vposer", True)))
        c.use_hands = int(bool(cfg.get("use_hands", True)))
        c.use_face = int(bool(cfg.get("use_face", True)))
        c.use_joints_conf = int(bool(cfg.get("use_joints_conf", False)))
        c.has_regression_pose = int(bool(has_regression_pose))
        c.use_conf_cam_init = int(bool(cfg.get("use_conf_for_camera_init", False)))
        c.num_body_joints = (NUM_BODY_JOINTS[cfg.get("format", "coco25")] if num_body_joints is None
                             else int(num_body_joints))
        c.maxiters = int(cfg.get("maxiters", 30))
        c.ftol = float(cfg.get("ftol", 1e-9)); c.gtol = float(cfg.get("gtol", 1e-9))
        c.lr = float(cfg.get("lr", 1.0)); c.rho = float(cfg.get("rho", 100))
        c.depth_loss_weight = float(cfg.get("depth_loss_weight", 1e2))
        c.lbs_mode = {"rows": 0, "dense": 1}[lbs_mode]
        c.reuse_entry_eval = int(bool(reuse_entry_eval))
        c.side_view_thsh = float(cfg.get("side_view_thsh", 0.0) or 0.0) if side_view else 0.0
        c.left_shoulder_idx = int(cfg.get("left_shoulder_idx", 2))
        c.right_shoulder_idx = int(cfg.get("right_shoulder_idx", 5))
        # interpenetration term (fitting.py:437-455): dense mode only, part labels from DeviceModel.set_parts
        c.interpenetration = int(bool(cfg.get("interpenetration", False)))
        c.max_collisions = int(cfg.get("max_collisions", 8))
        c.df_cone_height = float(cfg.get("df_cone_height", 0.5))
        c.penalize_outside = int(bool(cfg.get("penalize_outside", True)))
        c.point2plane = int(bool(cfg.get("point2plane", False)))
        if c.point2plane and cfg.get("interpenetration", False):
            _warn_point2plane()
        c.slots = int(slots or 0)
        # LBFGS hyper-parameters (optimizers/lbfgs_ls.py); the cfg files never set them: 0 = the reference's defaults
        # (tolerances: negative = default; an explicit 0 -- LBFGS(tolerance_grad=0): the test is disabled -- is passed through)
        tg, tc = cfg.get("lbfgs_tolerance_grad"), cfg.get("lbfgs_tolerance_change")
        c.lbfgs_tolerance_grad = -1.0 if tg is None else float(tg)
        c.lbfgs_tolerance_change = -1.0 if tc is None else float(tc)
        c.lbfgs_max_eval = int(cfg.get("lbfgs_max_eval", 0) or 0)
        c.lbfgs_history_size = int(cfg.get("lbfgs_history_size", 0) or 0)
        c.lbfgs_max_iter = int(cfg.get("lbfgs_max_iter", 0) or 0)
        # cfg float_dtype: float64 (main.py:99-105) -> the engine's high-precision mode (include/sfx.h sfx_batch_cfg.high_precision)
        c.high_precision = int(str(cfg.get("float_dtype", "float32")) == "float64" or bool(cfg.get("high_precision", False)))
        if c.interpenetration and lbs_mode != "dense":
            raise ValueError("interpenetration=True needs lbs_mode='dense' (the term reads every vertex)")
        self.use_vposer = bool(c.use_vposer)
        self.nemb = model.vposer_latent if self.use_vposer else 63
        arr = (capi.StageWeights * max(1, self.n_stages))(*stages)
        h = C.c_void_p()
        capi.check(self._lib.sfx_batch_create(model._h, C.byref(c), arr, C.byref(h)))
        self._h = h
        self.K = model.K
        self._trace_cap = 0

    # ---- data ------------------------------------------------------------------------------
    def set_frames(self, keypoints, joint_weights, cam_init_mask, focal, center, data_weight, est_tz=None,
                   cam_rot=None):
        B, K = self.B, self.K
        kp = capi.f32(keypoints).reshape(B, K, 3)
        jw = capi.f32(np.broadcast_to(np.asarray(joint_weights, np.float32), (B, K)))
        cm = capi.f32(np.broadcast_to(np.asarray(cam_init_mask, np.float32), (B, K)))
        cam = np.zeros((B, 6), np.float32)
        cam[:, 0] = cam[:, 1] = np.asarray(focal, np.float32)
        cam[:, 2:4] = np.asarray(center, np.float32).reshape(-1, 2)
        cam[:, 4] = np.asarray(data_weight, np.float32)
        if est_tz is not None:
            cam[:, 5] = np.asarray(est_tz, np.float32)
        R = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (B, 1)) if cam_rot is None else \
            capi.f32(cam_rot).reshape(B, 9)
        capi.check(self._lib.sfx_batch_set_frames(self._h, capi.fptr(kp), capi.fptr(jw), capi.fptr(cm),
                                                  capi.fptr(cam), capi.fptr(capi.f32(R))))

    def set_params(self, regression_pose=None, **p):
        B = self.B
        sizes = dict(cam_translation=3, global_orient=3, betas=self.model.num_betas,
                     left_hand_pose=self.model.num_pca, right_hand_pose=self.model.num_pca,
                     expression=self.model.num_expr, jaw_pose=3, leye_pose=3, reye_pose=3,
                     pose_embedding=self.nemb)
        args = []
        for n in PARAM_NAMES:
            v = p.get(n)
            a = None if v is None else capi.f32(np.broadcast_to(np.asarray(v, np.float32).reshape(-1, sizes[n]), (B, sizes[n])))
            args.append(a)
        reg = None if regression_pose is None else capi.f32(
            np.broadcast_to(np.asarray(regression_pose, np.float32).reshape(-1, self.nemb), (B, self.nemb)))
        capi.check(self._lib.sfx_batch_set_params(self._h, *[capi.fptr(a) for a in args], capi.fptr(reg)))

    def get_params(self):
        B = self.B
        sizes = [3, 3, self.model.num_betas, self.model.num_pca, self.model.num_pca, self.model.num_expr,
                 3, 3, 3, self.nemb, 63]
        outs = [np.zeros((B, n), np.float32) for n in sizes]
        capi.check(self._lib.sfx_batch_get_params(self._h, *[capi.fptr(a) for a in outs]))
        d = dict(zip(PARAM_NAMES + ("body_pose",), outs))
        return d

    # ---- compute ---------------------------------------------------------------------------
    def num_vars(self, stage):
        return self._lib.sfx_batch_num_vars(self._h, stage)

    def closure(self, stage):
        """(loss[B], grad[B,N]) at the current parameters; stage -1 = camera-init loss."""
        N = self.num_vars(stage)
        loss = np.zeros(self.B, np.float32)
        grad = np.zeros((self.B, N), np.float32)
        capi.check(self._lib.sfx_batch_closure(self._h, stage, capi.fptr(loss), capi.fptr(grad), None))
        return loss, grad

    def guess_init(self, pairs):
        p = capi.i32(np.asarray(pairs).reshape(-1, 2))
        capi.check(self._lib.sfx_batch_guess_init(self._h, capi.iptr(p), p.shape[0], None))

    def fit(self, first_stage=-1, last_stage=None, stream=None):
        last = self.n_stages - 1 if last_stage is None else last_stage
        capi.check(self._lib.sfx_batch_fit(self._h, first_stage, last, C.c_void_p(stream) if stream else None))

    def step(self, stage, resume):
        """One LBFGS.step for every frame; returns the entry losses [B]."""
        loss = np.zeros(self.B, np.float32)
        capi.check(self._lib.sfx_batch_step(self._h, stage, int(bool(resume)), capi.fptr(loss), None))
        return loss

    def set_gmm(self, prior):
        """Body pose prior = prior.MaxMixturePrior (body_prior_type 'gmm'): used by the closure when
        use_vposer is off and the batch has no regression pose (fitting.py:399-401)."""
        mu = np.ascontiguousarray(prior.means.detach().cpu().numpy(), np.float32)
        P = np.ascontiguousarray(prior.precisions.detach().cpu().numpy(), np.float32)
        nw = np.ascontiguousarray(prior.nll_weights.detach().cpu().numpy().reshape(-1), np.float32)
        cc = None
        if not getattr(prior, "use_merged", True):
            # MaxMixturePrior.log_likelihood (prior.py:203-225): d^T P d + 0.5 (log(det cov + eps) + D log 2 pi) per component
            import torch
            cov_term = torch.log(torch.det(prior.covs) + prior.epsilon)
            cc = np.ascontiguousarray((0.5 * (cov_term + prior.random_var_dim * prior.pi_term)).detach().cpu().numpy().reshape(-1), np.float32)
        capi.check(self._lib.sfx_batch_set_gmm_form(self._h, mu.shape[0], mu.shape[1], capi.fptr(mu), capi.fptr(P), capi.fptr(nw),
                                                    capi.fptr(cc) if cc is not None else None))

    _DEBUG_PER_FRAME = dict(verts="V3", vposed="V3", pen_dverts="V3", pen_dfeat=512, feat=512, pen_dA=55 * 12, A=12 * 55,
                            pen_loss=1)

    def debug_read(self, name):
        """Tests: one of the dense path's device buffers of the most recent evaluation as [B, per-frame]
        (sfx_batch_debug_read)."""
        per = self._DEBUG_PER_FRAME[name]
        per = self.model.V * 3 if per == "V3" else per
        out = np.zeros((self.B, per), np.float32)
        capi.check(self._lib.sfx_batch_debug_read(self._h, name.encode(), capi.fptr(out), out.size))
        return out

    def penetration_stats(self):
        """Diagnostics of the interpenetration term of the most recent evaluation (per frame):
        ordered pairs kept, partners dropped by max_collisions, grid overflow, vertices with gradient."""
        st = np.zeros((self.B, 4), np.int32)
        ext = np.zeros(self.B, np.int32)
        capi.check(self._lib.sfx_batch_pen_stats(self._h, capi.iptr(st), capi.iptr(ext)))
        return dict(pairs=st[:, 0].copy(), dropped=st[:, 1].copy(), entry_overflow=st[:, 2].copy(), walks_cut=st[:, 3].copy(), vertices=ext)

    def penetration_pairs(self, column):
        """Ordered pair list [n, 2] (receiving triangle, partner) of GEMM column `column` in the most recent evaluation
        (sfx_batch_pen_pairs): what BVH + FilterFaces hand to the loss in the reference (fitting.py:445-450)."""
        return _read_pairs(lambda cap, buf, n: self._lib.sfx_batch_pen_pairs(self._h, int(column), cap, buf, n))

    def penetration_flags(self):
        """Per frame: True when the frame's fit consumed a collision evaluation whose pair set depended on arrival order
        (a cut bucket walk; with max_collisions > 1024 also a partner list beyond 2 x max_collisions): its result is not
        reproducible run to run."""
        fl = np.zeros(self.B, np.int32)
        capi.check(self._lib.sfx_batch_pen_flags(self._h, capi.iptr(fl)))
        return fl != 0

    def penetration_launches(self):
        """Kernel launches of one interpenetration step of the fitting loop, counted on the captured graph (0 before a fit)."""
        return int(self._lib.sfx_batch_pen_launches(self._h))

    def last_grad(self, stage):
        """Gradient [B,N] of the most recent closure evaluation (what var.grad holds after step())."""
        grad = np.zeros((self.B, self.num_vars(stage)), np.float32)
        capi.check(self._lib.sfx_batch_get_grad(self._h, stage, capi.fptr(grad)))
        return grad

    def trace(self, capacity, evaluations=False):
        """Attach (capacity > 0 records per frame) or detach (0) the optimiser trace (sfx_batch_trace);
        evaluations=True adds one record per closure evaluation."""
        capi.check(self._lib.sfx_batch_trace(self._h, -int(capacity) if evaluations else int(capacity)))
        self._trace_cap = int(capacity)

    def get_trace(self):
        """Per frame: array [n, 4] of the records written since the trace was attached
        ((0, t, loss, ls_evals) | (1, entry loss, func_evals, n_iter) | (2, result, evals, stage))."""
        cap = self._trace_cap
        rec = np.zeros((self.B, cap, 4), np.float32)
        cnt = np.zeros(self.B, np.int32)
        capi.check(self._lib.sfx_batch_get_trace(self._h, capi.fptr(rec), capi.iptr(cnt)))
        if (cnt > cap).any():
            raise RuntimeError("optimiser trace overflowed: %d records, capacity %d" % (cnt.max(), cap))
        return [rec[i, :cnt[i]].copy() for i in range(self.B)]

    def stats(self):
        ns = self.n_stages + 1
        loss = np.zeros((self.B, ns), np.float32)
        ev = np.zeros((self.B, ns), np.int32)
        rev = np.zeros((self.B, ns), np.int32)
        capi.check(self._lib.sfx_batch_get_stats(self._h, capi.fptr(loss), capi.iptr(ev), capi.iptr(rev)))
        return dict(stage_loss=loss, stage_evals=ev, stage_ref_evals=rev)

    def forward(self, want_verts=True):
        """Final meshes/joints at the current parameters as torch CUDA tensors."""
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        verts = torch.empty([self.B, self.model.V, 3], dtype=torch.float32, device=dev) if want_verts else None
        joints = torch.empty([self.B, self.K, 3], dtype=torch.float32, device=dev)
        capi.check(self._lib.sfx_batch_forward(self._h, C.c_void_p(verts.data_ptr()) if want_verts else None,
                                               C.c_void_p(joints.data_ptr()), None))
        return verts, joints

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sfx_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prof_enable(on=True, every=1):
    """HIP-event timing of the named kernel launches; `every` = N times only every N-th launch of
    each name (an event pair costs two queue packets per launch)."""
    capi.load().sfx_prof_enable((int(on) & 3) | ((max(1, int(every)) & 0xffff) << 8))


def prof_reset():
    capi.load().sfx_prof_reset()


def prof_get(name):
    ms, n, u = C.c_double(), C.c_int64(), C.c_double()
    capi.load().sfx_prof_get(name.encode(), C.byref(ms), C.byref(n), C.byref(u))
    return ms.value, n.value, u.value


def loop_host_stats(reset=False):
    """Host side of the dense fitting loops since the last reset: seconds enqueueing, seconds waiting for stage flags, wall
    seconds, rounds (sfx_loop_host_stats)."""
    w = (C.c_double * 4)()
    capi.check(capi.load().sfx_loop_host_stats(w, int(bool(reset))))
    return dict(enqueue_s=w[0], wait_s=w[1], wall_s=w[2], rounds=int(w[3]))


def pen_work_reset():
    """Zero the device counters of the interpenetration term's work (sfx_pen_work_reset)."""
    capi.check(capi.load().sfx_pen_work_reset())


def pen_work_get():
    """Work of the interpenetration term since the last reset, counted on the device: grid entries and ordered pairs per
    column evaluation (a mesh that went through the broad phase), the number of those evaluations, surviving triangles."""
    w = (C.c_int64 * 6)()
    capi.check(capi.load().sfx_pen_work_get(w))
    cols = max(int(w[2]), 1)
    return dict(entries=int(w[0]), pairs=int(w[1]), columns=int(w[2]), survivors=int(w[3]),
                lists_overflowed=int(w[4]), walks_cut=int(w[5]),     # lists re-derived from the grid (exact); walks cut (> 0: pairs missing, order dependent)
                entries_per_column=w[0] / cols, pairs_per_column=w[1] / cols, survivors_per_column=w[3] / cols)


def pen_form(form=-1):
    """Debug / A-B: which form of the interpenetration term Penetration handles and FrameBatches created from now on take
    (sfx_debug_pen_form: 0 = the ten general kernels, the default; 1 = one workgroup per column behind the pair tests; 2 = form 1
    handing every column over).  LAB build only (include/sfx_lab.h).  Returns the previous setting; any other argument only queries."""
    capi.need_lab("pen_form")
    return int(capi.load().sfx_debug_pen_form(int(form)))


def pen_phase_ticks():
    """Debug: mean microseconds a column evaluation spent in the phases of k_pen_narrow since pen_work_reset()."""
    capi.need_lab("pen_phase_ticks")
    w = (C.c_int64 * 8)()
    capi.check(capi.load().sfx_debug_pen_phase_ticks(w))
    n = max(int(w[5]), 1)
    return dict(entry_us=w[0] / n / 100.0, list_us=w[1] / n / 100.0, eval_us=w[2] / n / 100.0, sums_us=w[3] / n / 100.0,
                verts_us=w[4] / n / 100.0, evaluations=int(w[5]), pairs_per_evaluation=w[6] / n)


def _warn_point2plane():
    import warnings
    warnings.warn("point2plane=True: the form built here -- Psi^2 (n_f . n_g)^2, gradient through both normals -- is assumption A6 of "
                  "oracle/penetration.py; the mesh_intersection package's own form is not available to compare with, so the numbers "
                  "may differ from the reference's (every shipped cfg leaves the flag False)", RuntimeWarning, stacklevel=3)


def _read_pairs(call):
    n = C.c_int32(0)
    capi.check(call(0, None, C.byref(n)))
    out = np.zeros((max(int(n.value), 1), 2), np.int32)
    capi.check(call(int(out.shape[0]), capi.iptr(out), C.byref(n)))
    return out[:int(n.value)].astype(np.int64)


class Penetration(object):
    """Interpenetration term on a batch of posed meshes (sfx_pen_*):
    BVH + FilterFaces + DistanceFieldPenetrationLoss of the reference's external package
    (fitting.py:437-455).  `segm` / `parents`: per-face part labels (smplx_parts_segm.pkl);
    `ign_part_pairs`: the cfg's ["9,16", ...] strings or (a, b) tuples."""

    def __init__(self, num_verts, faces, segm=None, parents=None, ign_part_pairs=None, max_collisions=128,
                 max_batch=1):
        self._lib = capi.load()
        faces = capi.i32(np.asarray(faces).astype(np.int64).reshape(-1, 3))
        self.V, self.F, self.max_batch = int(num_verts), int(faces.shape[0]), int(max_batch)
        pairs = []
        for p in ign_part_pairs or []:
            a, b = (int(x) for x in str(p).split(",")) if isinstance(p, str) else p
            pairs.append((a, b))
        ign = capi.i32(np.asarray(pairs, np.int64).reshape(-1, 2))
        sg = capi.i32(segm) if segm is not None else None
        pr = capi.i32(parents) if parents is not None else None
        h = C.c_void_p()
        capi.check(self._lib.sfx_pen_create(self.V, self.F, capi.iptr(faces), capi.iptr(sg) if sg is not None else None,
                                            capi.iptr(pr) if pr is not None else None,
                                            capi.iptr(ign) if len(pairs) else None, len(pairs), int(max_collisions),
                                            self.max_batch, C.byref(h)))
        self._h = h

    def eval(self, verts, sigma, penalize_outside=True, stream=None, point2plane=False):
        """verts: float32 CUDA tensor [B, V, 3] -> (loss [B], d loss / d verts [B, V, 3]) on the GPU.
        point2plane: DistanceFieldPenetrationLoss(point2plane=True) (include/sfx.h sfx_pen_set_point2plane)."""
        if point2plane:
            _warn_point2plane()
        import torch
        capi.check(self._lib.sfx_pen_set_point2plane(self._h, int(bool(point2plane))))
        assert verts.is_cuda and verts.dtype == torch.float32 and verts.shape[1:] == (self.V, 3)
        v = verts.contiguous()
        B = v.shape[0]
        loss = torch.empty([B], dtype=torch.float32, device=v.device)
        dv = torch.empty_like(v)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        capi.check(self._lib.sfx_pen_eval(self._h, B, C.c_void_p(v.data_ptr()), float(sigma), int(bool(penalize_outside)),
                                          C.c_void_p(loss.data_ptr()), C.c_void_p(dv.data_ptr()), C.c_void_p(s)))
        return loss, dv

    def eval_pairs(self, verts, pairs, sigma, penalize_outside=True, point2plane=False, stream=None):
        """DistanceFieldPenetrationLoss on pairs the CALLER supplies (sfx_pen_eval_pairs): verts float32 CUDA [B, V, 3], pairs
        int32 CUDA [B, n, 2] (each unordered pair once; rows with -1 are empty) -> (loss [B], d loss / d verts [B, V, 3],
        d loss / d triangle corners [B, F, 3, 3])."""
        import torch
        if point2plane:
            _warn_point2plane()
        capi.check(self._lib.sfx_pen_set_point2plane(self._h, int(bool(point2plane))))
        assert verts.is_cuda and verts.dtype == torch.float32 and verts.shape[1:] == (self.V, 3)
        assert pairs.is_cuda and pairs.dtype == torch.int32 and pairs.dim() == 3 and pairs.shape[2] == 2 and pairs.shape[0] == verts.shape[0]
        v, pr = verts.contiguous(), pairs.contiguous()
        B = v.shape[0]
        loss = torch.empty([B], dtype=torch.float32, device=v.device)
        dv = torch.empty_like(v)
        dtri = torch.empty([B, self.F, 3, 3], dtype=torch.float32, device=v.device)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        capi.check(self._lib.sfx_pen_eval_pairs(self._h, B, C.c_void_p(v.data_ptr()), C.c_void_p(pr.data_ptr()), int(pr.shape[1]), float(sigma),
                                                int(bool(penalize_outside)), C.c_void_p(loss.data_ptr()), C.c_void_p(dv.data_ptr()),
                                                C.c_void_p(dtri.data_ptr()), C.c_void_p(s)))
        return loss, dv, dtri

    def stats(self, B):
        out = np.zeros((B, 4), np.int32)
        capi.check(self._lib.sfx_pen_stats(self._h, int(B), capi.iptr(out)))
        return dict(pairs=out[:, 0].copy(), dropped=out[:, 1].copy(), entry_overflow=out[:, 2].copy(), walks_cut=out[:, 3].copy())

    def pairs(self, mesh):
        """Ordered pair list [n, 2] (receiving triangle, partner; receiver ascending, partner ascending) of mesh `mesh` of the
        most recent eval(): both orders of every colliding pair, after the part filter and the max_collisions rule."""
        return _read_pairs(lambda cap, buf, n: self._lib.sfx_pen_pairs(self._h, int(mesh), cap, buf, n))

    def phase_clocks(self, B):
        """Debug (LAB build): microseconds at the end of the broad phase's ten steps, grid entries (see sfx_pen_phase_clocks)."""
        capi.need_lab("phase_clocks")
        out = np.zeros((B, 11), np.int32)
        capi.check(self._lib.sfx_pen_phase_clocks(self._h, int(B), capi.iptr(out)))
        res = out / 100.0
        res[:, 10] = out[:, 10]
        return res

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sfx_pen_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
