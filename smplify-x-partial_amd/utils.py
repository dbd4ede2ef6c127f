"""Host-side helpers of the fitting path -- mirror of the reference's `utils` names that
the hot path touches (smplifyx/utils.py:60-95 rel_change/JointMapper/GMoF, :98-250
smpl_to_annotation, :306-436 _compute_euler_from_matrix).  Index tables are data; the
euler conversion is a closed form, not the reference's generic scipy-style routine.
"""
import numpy as np
import torch
import torch.nn as nn


def rel_change(prev_val, curr_val):
    """(prev - cur) / max(|prev|, |cur|, 1)  -- signed, utils.py:60-61."""
    return (prev_val - curr_val) / max([np.abs(prev_val), np.abs(curr_val), 1])


class JointMapper(nn.Module):
    """index_select(joints, 1, joint_maps); identity when no map (utils.py:68-81)."""

    def __init__(self, joint_maps=None):
        super().__init__()
        if joint_maps is None:
            self.joint_maps = None
        else:
            self.register_buffer("joint_maps", torch.as_tensor(np.asarray(joint_maps), dtype=torch.long))

    def forward(self, joints, **kwargs):
        if self.joint_maps is None:
            return joints
        return torch.index_select(joints, 1, self.joint_maps)


class GMoF(nn.Module):
    """rho^2 r^2 / (r^2 + rho^2), element-wise (utils.py:84-95).  Stand-alone module of the
    call surface; inside the fitting loop the robustifier lives in the HIP loss kernel."""

    def __init__(self, rho=1):
        super().__init__()
        self.rho = rho

    def extra_repr(self):
        return "rho = {}".format(self.rho)

    def forward(self, residual):
        sq = residual ** 2
        return self.rho ** 2 * torch.div(sq, sq + self.rho ** 2)


# ---- keypoint -> model-joint index tables (data) -------------------------------------------
# SMPL-X joint numbering: 0..54 kinematic joints, 55..75 vertex joints, 76..126 static face
# landmarks, 127..143 dynamic contour.
_SMPLX_BODY = {
    "coco25": [55, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7,
               56, 57, 58, 59, 60, 61, 62, 63, 64, 65],
    "halpe": [55, 57, 56, 59, 58, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8,
              15, 12, 0, 60, 63, 61, 64, 62, 65],
    "coco_wholebody": [55, 57, 56, 59, 58, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8,
                       60, 61, 62, 63, 64, 65],
    "coco19": [55, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 56, 57, 58, 59],
}
# wrist, then (index, middle, pinky->ring order of OpenPose) 3 joints + tip each
_FINGER_ORDER = [(37, 66), (25, 67), (28, 68), (34, 69), (31, 70)]   # thumb,index,middle,ring,pinky


def _hand(wrist, first, tip0):
    out = [wrist]
    for base, tip in _FINGER_ORDER:
        b = base + first
        out += [b, b + 1, b + 2, tip + tip0]
    return out


def smpl_to_annotation(model_type="smplx", use_hands=True, use_face=True,
                       use_face_contour=False, format="coco25"):
    """Indices that reorder SMPL-X joints into the keypoint format (utils.py:98-250).
    Only model_type 'smplx' is on the supported path."""
    if model_type != "smplx":
        raise ValueError("Unknown model type: {}".format(model_type))
    fmt = format if format == "coco19" else format.lower()
    if fmt not in _SMPLX_BODY:
        raise ValueError("Unknown joint format: {}".format(format))
    parts = [np.array(_SMPLX_BODY[fmt], dtype=np.int32)]
    shift = -6 if fmt == "coco19" else 0          # coco19 has no foot keypoints: 6 fewer vertex joints
    if use_hands:
        parts.append(np.array(_hand(20, 0, shift), dtype=np.int32))
        parts.append(np.array(_hand(21, 15, shift + 5), dtype=np.int32))
    if use_face:
        start = 76 + shift
        parts.append(np.arange(start, start + 51 + 17 * use_face_contour, dtype=np.int32))
    return np.concatenate(parts)


def euler_xyz_from_matrix(R):
    """Intrinsic x-y-z Euler angles of rotation matrices [...,3,3] -> [...,3].

    The reference converts regressor rotation matrices with a port of scipy's
    `as_euler('xyz')` for intrinsic rotations (utils.py:306-436) and then feeds the triples
    to the optimiser AS IF they were axis-angle vectors (fit_single_frame.py:212-235) -- a
    quirk this engine reproduces.  For R = Rx(a) Ry(b) Rz(c):
        b = asin(R02), a = atan2(-R12, R22), c = atan2(-R01, R00)
    which equals the reference's output away from gimbal lock (|b| = pi/2; there the
    reference sets c = 0 and a = atan2(R10 -/+ R01, R00 +/- R11))."""
    R = np.asarray(R, np.float64)
    r02 = np.clip(R[..., 0, 2], -1.0, 1.0)
    b = np.arcsin(r02)
    a = np.arctan2(-R[..., 1, 2], R[..., 2, 2])
    c = np.arctan2(-R[..., 0, 1], R[..., 0, 0])
    lock = np.abs(np.abs(b) - np.pi / 2) < 1e-7
    if np.any(lock):
        pos = r02 > 0
        # gimbal lock: third angle 0, first angle = the remaining in-plane rotation
        a_l = np.where(pos, np.arctan2(R[..., 1, 0], R[..., 1, 1]),
                       np.arctan2(-R[..., 1, 0], R[..., 1, 1]))
        a = np.where(lock, a_l, a)
        c = np.where(lock, 0.0, c)
    return np.stack([a, b, c], axis=-1)


def regression_prior_pose(regression_prior, expose=None, pixie=None, pare=None):
    """Initial body pose [63] and global orientation [3] from regressor outputs
    (fit_single_frame.py:209-235): every 3x3 rotation -> xyz-euler triple; 'combined' takes
    ExPose joints 0..18 and PIXIE joints 19..20, global orientation from ExPose."""
    if regression_prior in ("PIXIE", "combined"):
        pixie_pose = euler_xyz_from_matrix(np.asarray(pixie["body_pose"]))
        glob = euler_xyz_from_matrix(np.asarray(pixie["global_pose"]))[0]
    if regression_prior in ("ExPose", "combined"):
        expose_pose = euler_xyz_from_matrix(np.asarray(expose["body_pose"]))
        glob = euler_xyz_from_matrix(np.asarray(expose["global_orient"]))[0]
    if regression_prior == "PARE":
        pp = np.asarray(pare["pred_pose"])
        pare_pose = euler_xyz_from_matrix(pp[0, 1:22])
        glob = euler_xyz_from_matrix(pp[0, :1])[0]
    if regression_prior == "PIXIE":
        pose = pixie_pose
    elif regression_prior == "ExPose":
        pose = expose_pose
    elif regression_prior == "PARE":
        pose = pare_pose
    elif regression_prior == "combined":
        pose = np.concatenate([expose_pose[:19], pixie_pose[19:]])
    else:
        raise ValueError("Unknown regression prior: {}".format(regression_prior))
    return pose.reshape(-1).astype(np.float32), glob.astype(np.float32)
