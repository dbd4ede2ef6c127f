"""ORACLE (test infrastructure): PyTorch-CPU restatement of the in-tree objective of the
reference's hot path.  Pinned against the real reference by tests/golden/objective_*.npz
(tools/make_goldens.py imports /root/reference/smplifyx and evaluates the originals on
the same seeded inputs).

Restates
  GMoF.forward                           smplifyx/utils.py:84-95
  PerspectiveCamera.forward              smplifyx/camera.py:93-117
  SMPLifyAnglePrior / L2Prior            smplifyx/prior.py:53-97
  SMPLifyLoss.forward                    smplifyx/fitting.py:375-461 (no interpenetration)
  SMPLifyCameraInitLoss.forward          smplifyx/fitting.py:499-520
  rel_change                             smplifyx/utils.py:60-61
  guess_init                             smplifyx/fitting.py:36-110
as plain functions of tensors (no modules, no hidden state).
"""
import numpy as np
import torch

ANGLE_IDX = (52, 55, 9, 12)         # prior.py:61 minus the 3 global-orient dims (:86)
ANGLE_SIGN = (1.0, -1.0, -1.0, -1.0)


def gmof(residual, rho):
    sq = residual ** 2
    return rho ** 2 * torch.div(sq, sq + rho ** 2)


def project(points, rotation, translation, focal_x, focal_y, center):
    """points[B,K,3] -> pixels[B,K,2];  p' = R p + t,  uv = f * p'_xy / p'_z + c."""
    B = points.shape[0]
    Rt = torch.cat([rotation, translation.unsqueeze(-1)], dim=2)              # [B,3,4]
    ph = torch.cat([points, torch.ones_like(points[..., :1])], dim=-1)
    cam = torch.einsum("bki,bji->bjk", Rt, ph)
    img = cam[:, :, :2] / cam[:, :, 2:3]
    f = torch.stack([focal_x, focal_y], dim=-1).view(B, 1, 2)
    return img * f + center.unsqueeze(1)


def angle_prior(body_pose):
    idx = torch.as_tensor(ANGLE_IDX, dtype=torch.long)
    sign = torch.as_tensor(ANGLE_SIGN, dtype=body_pose.dtype)
    return torch.exp(body_pose[:, idx] * sign).pow(2)


def rel_change(prev, cur):
    return (prev - cur) / max(abs(prev), abs(cur), 1)


def smplify_terms(out, proj, gt_joints, joints_conf, joint_weights, w, pose_embedding,
                  use_vposer=False, regression_pose=None, stage=0, num_stages=3,
                  use_joints_conf=True, use_hands=True, use_face=True, rho=100.0, body_pose_prior=None):
    """All terms of SMPLifyLoss.forward as a dict (+ 'total').  `w` holds the stage's 0-d
    weight tensors: data_weight, body_pose_weight, shape_weight, bending_prior_weight,
    hand_prior_weight, expr_prior_weight, jaw_prior_weight[3]."""
    weights = (joint_weights * joints_conf if use_joints_conf else joint_weights).unsqueeze(-1)
    terms = {}
    terms["joint"] = torch.sum(weights ** 2 * gmof(gt_joints - proj, rho)) * w["data_weight"] ** 2
    if use_vposer:
        if stage + 1 == num_stages and regression_pose is not None:
            terms["pprior"] = (pose_embedding - regression_pose).pow(2).sum() * w["body_pose_weight"] ** 2
        else:
            terms["pprior"] = pose_embedding.pow(2).sum() * w["body_pose_weight"] ** 2
    elif regression_pose is not None:
        terms["pprior"] = (pose_embedding - regression_pose).pow(2).sum() * w["body_pose_weight"] ** 2
    elif body_pose_prior is not None:   # body_prior_type 'gmm' (fitting.py:399-401, oracle/prior_gmm.py)
        terms["pprior"] = torch.sum(body_pose_prior(out.body_pose, out.betas)) * w["body_pose_weight"] ** 2
    else:   # body_prior_type 'l2' on the model's body_pose (fitting.py:399-401)
        terms["pprior"] = out.body_pose.pow(2).sum() * w["body_pose_weight"] ** 2
    terms["shape"] = out.betas.pow(2).sum() * w["shape_weight"] ** 2
    terms["angle"] = torch.sum(angle_prior(out.full_pose[:, 3:66])) * w["bending_prior_weight"]
    if use_hands:
        terms["lhand"] = out.left_hand_pose.pow(2).sum() * w["hand_prior_weight"] ** 2
        terms["rhand"] = out.right_hand_pose.pow(2).sum() * w["hand_prior_weight"] ** 2
    if use_face:
        terms["expr"] = out.expression.pow(2).sum() * w["expr_prior_weight"] ** 2
        terms["jaw"] = out.jaw_pose.mul(w["jaw_prior_weight"]).pow(2).sum()
    total = terms["joint"] + terms["pprior"] + terms["shape"] + terms["angle"]
    # summation order of fitting.py:457-460: joint + pprior + shape + angle + pen(0) + jaw + expr + lh + rh
    if use_face:
        total = total + terms["jaw"] + terms["expr"]
    if use_hands:
        total = total + terms["lhand"] + terms["rhand"]
    terms["total"] = total
    return terms


def camera_init_loss(proj, gt_joints, init_idxs, data_weight, depth_loss_weight,
                     cam_tz, est_tz, joints_conf=None, use_conf=False):
    """SMPLifyCameraInitLoss.forward incl. the use_conf broadcast quirk: the double
    unsqueeze makes the term (sum_i conf_i^2) * (sum_j err_j)  (fitting.py:509-511)."""
    idx = torch.as_tensor(np.asarray(init_idxs), dtype=torch.long)
    err = (torch.index_select(gt_joints, 1, idx) - torch.index_select(proj, 1, idx)) ** 2
    if use_conf:
        c = torch.index_select(joints_conf, 1, idx)
        joint = torch.sum(err.unsqueeze(1) * (c ** 2).view(c.shape[0], -1, 1, 1)) * data_weight ** 2
    else:
        joint = torch.sum(err) * data_weight ** 2
    depth = 0.0
    if float(depth_loss_weight) > 0 and est_tz is not None:
        depth = depth_loss_weight ** 2 * torch.sum((cam_tz - est_tz).pow(2))
    return joint + depth


def guess_init_depth(joints_3d, joints_2d, edge_idxs, focal_length):
    """t_z = f * mean||d3D|| / mean||d2D|| over limb pairs (fitting.py:87-102)."""
    d3 = torch.stack([joints_3d[:, a] - joints_3d[:, b] for a, b in edge_idxs], dim=1)
    d2 = torch.stack([joints_2d[:, a] - joints_2d[:, b] for a, b in edge_idxs], dim=1)
    l2 = d2.pow(2).sum(-1).sqrt().mean(dim=1)
    l3 = d3.pow(2).sum(-1).sqrt().mean(dim=1)
    return focal_length * (l3 / l2)
