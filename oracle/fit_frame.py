"""ORACLE (test infrastructure): restatement of the hot path of
smplifyx/fit_single_frame.py:119-660 (one frame, one person) on top of
oracle/body_model.py, oracle/objective.py and oracle/lbfgs_machine.py.

Pinned against the real reference by tests/golden/e2e_*.npz: tools/make_goldens.py runs
the reference's own fit_single_frame() (imported from /root/reference) with this
package's SMPLXRef body model and records per-stage losses / final parameters.

Covered: weight schedule (:133-207,330-353), regression-prior initialisation (:209-235,
given the euler triples), confidence thresholding (:276-294), camera prior / guess_init
(:358-411), camera stage (:447-496), orientation x stage loop (:527-612), result dict
(:644-660).  Not covered: VPoser encode().sample() (random, :245), interpenetration
(:300-328), visualisation and file output.
"""
import numpy as np
import torch

from . import objective as obj
from .lbfgs_machine import StageMachine

NUM_BODY_JOINTS = {"coco25": 25, "halpe": 26, "coco_wholebody": 23}


def _jaw_weights(cfg, shape_weights):
    jw = cfg.get("jaw_pose_prior_weights")
    if jw is None:
        return [[x] * 3 for x in shape_weights]
    out = []
    for e in jw:
        out.append([float(v) for v in e.split(",")] if isinstance(e, str) else [float(v) for v in e])
    return out


def build_schedule(cfg, dtype=torch.float32):
    """List of per-stage weight dicts (fit_single_frame.py:133-207,330-353); lists are
    zip-truncated to the shortest one exactly like the reference."""
    use_hands, use_face = cfg.get("use_hands", True), cfg.get("use_face", True)
    bpw = cfg.get("body_pose_prior_weights") or [4.04 * 1e2, 4.04 * 1e2, 57.4, 4.78]
    n = len(bpw)
    d = {"data_weight": cfg.get("data_weights") or [1] * n,
         "body_pose_weight": bpw,
         "shape_weight": cfg.get("shape_weights") or [1e2, 5 * 1e1, 1e1, .5 * 1e1]}
    if use_face:
        d["face_weight"] = cfg.get("face_joints_weights") or [0.0, 0.0, 0.0, 1.0]
        d["expr_prior_weight"] = cfg.get("expr_weights") or [1e2, 5 * 1e1, 1e1, .5 * 1e1]
        d["jaw_prior_weight"] = _jaw_weights(cfg, d["shape_weight"])
    if use_hands:
        d["hand_weight"] = cfg.get("hand_joints_weights") or [0.0, 0.0, 0.0, 1.0]
        d["hand_prior_weight"] = cfg.get("hand_pose_prior_weights") or [1e2, 5 * 1e1, 1e1, .5 * 1e1]
    if cfg.get("interpenetration", False):
        d["coll_loss_weight"] = cfg.get("coll_loss_weights") or [0.0] * n
    d["global_orient_weight"] = cfg.get("global_orient_weights") or [20, 10, 7.5, 5, 5]
    keys = list(d.keys())
    stages = [dict(zip(keys, vals)) for vals in zip(*(d[k] for k in keys))]
    for st in stages:
        for k in st:
            st[k] = torch.tensor(st[k], dtype=dtype)
    return stages


def rotvec_to_mat(r):
    r = np.asarray(r, np.float64).reshape(3)
    a = np.linalg.norm(r)
    if a < 1e-12:
        return np.eye(3)
    k = r / a
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def mat_to_rotvec(R):
    """cv2.Rodrigues(matrix) semantics: axis * angle, angle in [0, pi]."""
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
    """fit_single_frame.py:528-535: R(orient) . R([0, pi, 0]) back to a rotation vector."""
    return mat_to_rotvec(rotvec_to_mat(body_orient) @ rotvec_to_mat([0.0, np.pi, 0.0]))


class FrameFit(object):
    """Holds one frame's optimisation problem; `run()` executes the reference schedule."""

    def __init__(self, bm, keypoints, H, W, focal_length, cfg, joint_weights,
                 reg_pose=None, reg_global=None, cam_prior=None, vposer=None,
                 dtype=torch.float32, reuse_entry_eval=False, machine_kwargs=None, body_pose_prior=None):
        self.bm, self.cfg, self.dtype = bm, cfg, dtype
        self.body_pose_prior = body_pose_prior
        self.vposer = vposer
        self.reuse = reuse_entry_eval
        self.mk = machine_kwargs or {}
        self.np_dtype = np.float32 if dtype == torch.float32 else np.float64
        self.use_vposer = bool(cfg.get("use_vposer", True))
        self.use_hands, self.use_face = cfg.get("use_hands", True), cfg.get("use_face", True)
        self.use_conf = cfg.get("use_joints_conf", False)
        self.H, self.W, self.focal = H, W, float(focal_length)
        nb = NUM_BODY_JOINTS[cfg.get("format", "coco25")]
        self.nb = nb
        kd = torch.tensor(np.asarray(keypoints), dtype=dtype)
        self.gt = kd[:, :, :2]
        self.conf = kd[:, :, 2].reshape(1, -1)
        K = self.gt.shape[1]
        thr = np.array([cfg.get("confidence_threshold", 0)] * nb + [0] * 42 + [0] * 68)
        self.low = [i for i in range(K) if float(self.conf[0, i]) < thr[i]]
        self.jw = torch.as_tensor(np.asarray(joint_weights), dtype=dtype).reshape(1, -1).clone()
        self.jw[:, self.low] = 0
        self.init_idxs = [i for i in cfg.get("init_joints_idxs", (9, 12, 2, 5))
                          if float(self.gt[0, i, 0]) != 0 and float(self.gt[0, i, 1]) != 0
                          and i not in self.low]
        self.stages = build_schedule(cfg, dtype)
        self.regression = reg_pose is not None
        if self.regression:
            self.pose_embedding = torch.tensor(np.asarray(reg_pose), dtype=dtype).reshape(1, -1).requires_grad_(True)
            self.reg_global = torch.tensor(np.asarray(reg_global), dtype=dtype).reshape(1, 3)
        elif self.use_vposer:
            self.pose_embedding = torch.zeros([1, cfg.get("vposer_latent_dim", 32)], dtype=dtype, requires_grad=True)
        elif body_pose_prior is not None:        # fit_single_frame.py:250-252: start from the mixture's mean
            self.pose_embedding = body_pose_prior.get_mean().to(dtype).clone().detach().requires_grad_(True)
        else:
            raise ValueError("use_vposer=False needs a regression prior (the reference crashes here, "
                             "fit_single_frame.py:252 with body_prior_type 'l2')")
        self.regression_pose = self.pose_embedding.clone().detach() if self.regression else None
        new = dict(body_pose=self.pose_embedding)
        if self.regression:
            new["global_orient"] = self.reg_global
        bm.reset_params(**new)
        # camera
        self.cam_R = torch.eye(3, dtype=dtype).unsqueeze(0)
        self.cam_t = torch.zeros([1, 3], dtype=dtype, requires_grad=True)
        self.fx = torch.full([1], self.focal, dtype=dtype)
        self.fy = torch.full([1], self.focal, dtype=dtype)
        self.center = torch.zeros([1, 2], dtype=dtype)
        if cfg.get("use_camera_prior") and self.regression and cam_prior is not None:
            init_t = torch.tensor(np.asarray(cam_prior["init_t"]), dtype=dtype).reshape(1, -1)
            cx, cy = cam_prior["center"]
            with torch.no_grad():
                self.cam_t[:] = init_t
                self.center[:] = torch.tensor([cx, cy], dtype=dtype)
        else:
            with torch.no_grad():
                out = bm(body_pose=self._body_pose(), return_verts=False, return_full_pose=False)
                est = obj.guess_init_depth(out.joints, self.gt, cfg.get("body_tri_idxs", [(5, 12), (2, 9)]),
                                           self.focal)
                init_t = torch.stack([torch.zeros([1], dtype=dtype), torch.zeros([1], dtype=dtype), est], dim=1)
                self.cam_t[:] = init_t
                self.center[:] = torch.tensor([W, H], dtype=dtype) * 0.5
        self.init_t = init_t.clone().detach()
        self.data_weight = torch.tensor(1000 / H, dtype=dtype)
        self.depth_loss_weight = torch.tensor(cfg.get("depth_loss_weight", 1e2), dtype=dtype)
        self.evals = []
        self.stage_losses = []

    # ---- pieces ---------------------------------------------------------------------------
    def _body_pose(self):
        if self.use_vposer:
            return self.vposer.decode(self.pose_embedding, output_type="aa").view(1, -1)
        return self.pose_embedding.reshape(1, -1)

    def _project(self, joints):
        return obj.project(joints, self.cam_R, self.cam_t, self.fx, self.fy, self.center)

    def camera_objective(self):
        out = self.bm(return_verts=False, body_pose=self._body_pose(), return_full_pose=False)
        return obj.camera_init_loss(self._project(out.joints), self.gt, self.init_idxs,
                                    self.data_weight, self.depth_loss_weight,
                                    self.cam_t[:, 2], self.init_t[:, 2],
                                    joints_conf=self.conf,
                                    use_conf=bool(self.cfg.get("use_conf_for_camera_init")))

    def set_penetration(self, faces, segm=None, parents=None, ign_part_pairs=None):
        """Switch the interpenetration term on (fitting.py:437-455; oracle/penetration.py)."""
        self.pen = dict(faces=np.asarray(faces).astype(np.int64), segm=segm, parents=parents, ign=ign_part_pairs)

    def penetration_term(self, vertices, w):
        from . import penetration as P
        cw = float(w.get("coll_loss_weight", 0.0) or 0.0)
        if getattr(self, "pen", None) is None or cw <= 0:
            return None
        v = vertices[0]
        pairs = P.candidate_pairs(v.detach().numpy(), self.pen["faces"], self.pen["segm"], self.pen["parents"], self.pen["ign"])
        # BVH(max_collisions): a triangle keeps at most max_collisions partners (the lowest ids: oracle/penetration.py)
        opairs, self.pen_cut = P.ordered_pairs_capped(pairs, int(self.cfg.get("max_collisions", 8)))
        return cw * P.penetration_loss_ordered(v, self.pen["faces"], opairs, float(self.cfg.get("df_cone_height", 0.5)),
                                               bool(self.cfg.get("penalize_outside", True)), bool(self.cfg.get("point2plane", False)))

    def body_terms(self, stage, w, jw):
        out = self.bm(return_verts=True, body_pose=self._body_pose(), return_full_pose=True)
        terms = self._body_terms_nopen(out, stage, w, jw)
        pen = self.penetration_term(out.vertices, w)
        if pen is not None:
            terms = dict(terms)
            terms["penetration"] = pen
            terms["total"] = terms["total"] + pen
        return terms

    def _body_terms_nopen(self, out, stage, w, jw):
        return obj.smplify_terms(out, self._project(out.joints), self.gt, self.conf, jw, w,
                                 self.pose_embedding, use_vposer=self.use_vposer,
                                 regression_pose=self.regression_pose, stage=stage,
                                 num_stages=len(self.cfg.get("body_pose_prior_weights") or [0] * 4),
                                 use_joints_conf=self.use_conf, use_hands=self.use_hands,
                                 use_face=self.use_face, rho=self.cfg.get("rho", 100),
                                 body_pose_prior=None if (self.use_vposer or self.regression) else self.body_pose_prior)

    # ---- flat-vector closure -----------------------------------------------------------------
    def _make_closure(self, params, fn):
        sizes = [p.numel() for p in params]

        def closure(x):
            xt = torch.as_tensor(x, dtype=self.dtype)
            o = 0
            with torch.no_grad():
                for p, n in zip(params, sizes):
                    p.copy_(xt[o:o + n].view_as(p)); o += n
            for p in params:
                p.grad = None
            loss = fn()
            loss.backward()
            g = torch.cat([(p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), dtype=self.dtype))
                           for p in params])
            return loss.item(), g.numpy().copy()
        groups, o = [], 0
        for p, n in zip(params, sizes):
            groups.append([o, n, True]); o += n
        return closure, groups

    def _optimise(self, params, fn):
        closure, groups = self._make_closure(params, fn)
        x0 = torch.cat([p.detach().reshape(-1) for p in params]).numpy()
        # which groups ever receive a gradient (fitting.py:191 skips grad=None params)
        _, g0 = closure(x0.copy())
        for gi, p in zip(groups, params):
            gi[2] = p.grad is not None
        m = StageMachine(x0, groups=[tuple(g) for g in groups], maxiters=self.cfg.get("maxiters", 30),
                         ftol=self.cfg.get("ftol", 1e-9), gtol=self.cfg.get("gtol", 1e-9),
                         lr=self.cfg.get("lr", 1.0), dtype=self.np_dtype,
                         reuse_entry_eval=self.reuse, **self.mk)
        while not m.done:
            f, g = closure(m.x_trial.copy())
            m.feed(f, g)
        closure(m.x.copy())          # leave the parameters at the accepted point
        self.evals.append(m.evals)
        return m

    # ---- schedule ---------------------------------------------------------------------------
    def run(self):
        cfg, bm = self.cfg, self.bm
        sd = torch.dist(self.gt[:, cfg.get("left_shoulder_idx", 2)], self.gt[:, cfg.get("right_shoulder_idx", 5)])
        try_both = sd.item() < cfg.get("side_view_thsh", 25.)
        bm.global_orient.requires_grad_(True)
        m = self._optimise([self.cam_t, bm.global_orient], self.camera_objective)
        self.cam_loss = m.result
        orient0 = bm.global_orient.detach().numpy().copy()
        orients = [orient0]
        if try_both:
            orients.append(flipped_orientation(orient0.ravel()).astype(self.np_dtype).reshape(1, 3))
        results = []
        for orient in orients:
            bm.reset_params(global_orient=orient, body_pose=self.pose_embedding)
            final = 0
            losses = []
            for si, w in enumerate(self.stages):
                params = [p for p in bm.parameters() if p.requires_grad] + [self.pose_embedding]
                w = dict(w)
                w["data_weight"] = self.data_weight
                w["bending_prior_weight"] = 3.17 * w["body_pose_weight"]
                jw = self.jw
                if self.use_hands:
                    jw[:, self.nb:self.nb + 42] = w["hand_weight"]
                if self.use_face:
                    jw[:, self.nb + 42:] = w["face_weight"]
                jw[:, self.low] = 0
                m = self._optimise(params, lambda: self.body_terms(si, w, jw)["total"])
                final = m.result
                losses.append(final)
            with torch.no_grad():
                out = bm(return_verts=True, body_pose=self._body_pose(), return_full_pose=True)
            res = {"camera_rotation": self.cam_R.numpy().copy(),
                   "camera_translation": self.cam_t.detach().numpy().copy(),
                   "camera_center": self.center.numpy().copy(),
                   "H": self.H, "W": self.W, "focal_length": self.focal}
            res.update({k: v.detach().numpy().copy() for k, v in bm.named_parameters()})
            res["body_pose"] = self._body_pose().detach().numpy().copy()
            results.append({"loss": final, "result": res, "stage_losses": losses,
                            "vertices": out.vertices.numpy().copy(), "joints": out.joints.numpy().copy()})
        idx = 0
        if len(results) > 1:
            idx = 0 if results[0]["loss"] < results[1]["loss"] else 1
        best = results[idx]
        best["evals"] = list(self.evals)
        best["cam_loss"] = self.cam_loss
        best["n_orient"] = len(results)
        return best
