"""ORACLE (test infrastructure, never shipped as the product path): PyTorch-CPU
restatement of the SMPL-X forward the reference calls through the un-vendored
``smplx`` package (fork github.com/xiyichen/smplx, un-pinned HEAD; reference call
sites smplifyx/main.py:123-127, smplifyx/fitting.py:82,248,
smplifyx/fit_single_frame.py:274,551,611, smplifyx/camera.py:27).

PARITY UNPINNED for this file: the ``smplx`` source and the licensed model files are
absent from /root/reference and from this image, and the reference has no tests, so
there is no golden vector for LBS itself.  The algorithm below follows the published
``smplx.lbs`` / ``smplx.body_models.SMPLX.forward`` as summarised in SURVEY.md 3.4 and
appendix A.1/A.2, and is cross-checked in tests/ against an independent fp64 numpy
implementation (oracle/lbs_numpy.py), finite differences and invariants.
Everything that IS in the reference tree (loss, camera, priors, L-BFGS, schedule) is
pinned against the real reference through tests/golden (tools/make_goldens.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

ModelOutput = namedtuple(
    "ModelOutput",
    ["vertices", "joints", "full_pose", "betas", "global_orient", "body_pose",
     "expression", "left_hand_pose", "right_hand_pose", "jaw_pose"])
ModelOutput.__new__.__defaults__ = (None,) * len(ModelOutput._fields)

NECK_CHAIN = (12, 9, 6, 3, 0)   # neck -> root (appendix A.2 find_dynamic_lmk_idx_and_bcoords)


def transform_mat(R, t):
    """[[R, t], [0 0 0 1]] -- smplx.lbs.transform_mat (used by smplifyx/camera.py:102)."""
    return torch.cat([torch.nn.functional.pad(R, [0, 0, 0, 1]),
                      torch.nn.functional.pad(t, [0, 0, 0, 1], value=1)], dim=2)


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """Appendix A.2: angle = ||theta + 1e-8|| (eps added to every component first)."""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + epsilon, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle).unsqueeze(1)
    sin = torch.sin(angle).unsqueeze(1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=rot_vecs.dtype, device=rot_vecs.device)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(n, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device).unsqueeze(0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def rigid_transform_chain(rot_mats, joints, parents):
    """batch_rigid_transform of appendix A.2 -> (posed_joints[B,J,3], A[B,J,4,4])."""
    B, J = joints.shape[:2]
    joints = joints.unsqueeze(-1)
    rel = joints.clone()
    rel[:, 1:] = rel[:, 1:] - joints[:, parents[1:]]
    M = transform_mat(rot_mats.reshape(-1, 3, 3), rel.reshape(-1, 3, 1)).reshape(B, J, 4, 4)
    chain = [M[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[int(parents[i])], M[:, i]))
    G = torch.stack(chain, dim=1)
    posed = G[:, :, :3, 3]
    jh = torch.nn.functional.pad(joints, [0, 0, 0, 1])
    A = G - torch.nn.functional.pad(torch.matmul(G, jh), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed, A


def rot_mat_to_euler(R):
    sy = torch.sqrt(R[:, 0, 0] * R[:, 0, 0] + R[:, 1, 0] * R[:, 1, 0])
    return torch.atan2(-R[:, 2, 0], sy)


def dynamic_lmk_rows(full_pose, dtype):
    """LUT row (0..78) of the dynamic face contour; no gradient (appendix A.2)."""
    B = full_pose.shape[0]
    chain = torch.as_tensor(NECK_CHAIN, dtype=torch.long, device=full_pose.device)
    aa = torch.index_select(full_pose.view(B, -1, 3), 1, chain)
    R = batch_rodrigues(aa.reshape(-1, 3)).view(B, -1, 3, 3)
    rel = torch.eye(3, dtype=dtype, device=full_pose.device).unsqueeze(0).expand(B, -1, -1)
    for k in range(len(NECK_CHAIN)):
        rel = torch.bmm(R[:, k], rel)
    y = torch.round(torch.clamp(-rot_mat_to_euler(rel) * 180.0 / np.pi, max=39)).to(torch.long)
    neg = y.lt(0).to(torch.long)
    big = y.lt(-39).to(torch.long)
    neg_vals = big * 78 + (1 - big) * (39 - y)
    return neg * neg_vals + (1 - neg) * y


class SMPLXRef(nn.Module):
    """nn.Module with the surface fit_single_frame()/fitting.py touch:
    parameters (names = result-pkl keys), reset_params(**d), forward(return_verts,
    body_pose, return_full_pose) -> ModelOutput, faces_tensor, faces."""

    def __init__(self, model, joint_map=None, num_betas=10, num_expression_coeffs=10,
                 num_pca_comps=12, use_pca=True, flat_hand_mean=False,
                 use_face_contour=True, create_body_pose=True, batch_size=1,
                 extra_vertex_ids=None, dtype=torch.float32):
        super().__init__()
        self.dtype = dtype
        self.use_pca = use_pca
        self.use_face_contour = use_face_contour
        self.num_pca_comps = num_pca_comps
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
        sd = np.asarray(model["shapedirs"])
        nb, ne = num_betas, num_expression_coeffs
        expr_start = 300 if sd.shape[-1] >= 400 else 10
        sd = np.concatenate([sd[:, :, :nb], sd[:, :, expr_start:expr_start + ne]], axis=-1)
        self.num_betas, self.num_expr = nb, ne
        V = model["v_template"].shape[0]
        self.register_buffer("v_template", t(model["v_template"]))
        self.register_buffer("shapedirs", t(sd))
        pd = np.asarray(model["posedirs"])
        self.register_buffer("posedirs", t(pd.reshape(-1, pd.shape[-1]).T))   # [486, 3V]
        self.register_buffer("J_regressor", t(model["J_regressor"]))
        self.register_buffer("lbs_weights", t(model["weights"]))
        parents = np.asarray(model["kintree_table"])[0].astype(np.int64).copy()
        parents[0] = -1
        self.register_buffer("parents", torch.as_tensor(parents))
        faces = np.asarray(model["f"]).astype(np.int64)
        self.faces = faces
        self.register_buffer("faces_tensor", torch.as_tensor(faces))
        lc = np.asarray(model["hands_componentsl"])[:num_pca_comps]
        rc = np.asarray(model["hands_componentsr"])[:num_pca_comps]
        self.register_buffer("left_hand_components", t(lc))
        self.register_buffer("right_hand_components", t(rc))
        lm = np.zeros(45) if flat_hand_mean else np.asarray(model["hands_meanl"])
        rm = np.zeros(45) if flat_hand_mean else np.asarray(model["hands_meanr"])
        pose_mean = np.concatenate([np.zeros(3 + 63 + 9), lm, rm])
        self.register_buffer("pose_mean", t(pose_mean))
        self.register_buffer("lmk_faces_idx", torch.as_tensor(np.asarray(model["lmk_faces_idx"]).astype(np.int64)))
        self.register_buffer("lmk_bary_coords", t(model["lmk_bary_coords"]))
        self.register_buffer("dynamic_lmk_faces_idx",
                             torch.as_tensor(np.asarray(model["dynamic_lmk_faces_idx"]).astype(np.int64)))
        self.register_buffer("dynamic_lmk_bary_coords", t(model["dynamic_lmk_bary_coords"]))
        if extra_vertex_ids is None:
            extra_vertex_ids = model["extra_vertex_ids"]
        self.register_buffer("extra_vertex_ids", torch.as_tensor(np.asarray(extra_vertex_ids).astype(np.int64)))
        if joint_map is not None:
            self.register_buffer("joint_map", torch.as_tensor(np.asarray(joint_map).astype(np.int64)))
        else:
            self.joint_map = None
        B = batch_size
        z = lambda n: nn.Parameter(torch.zeros([B, n], dtype=dtype))
        # registration order of smplx.SMPL/SMPLH/SMPLX.__init__ [external]
        self.betas = z(nb)
        self.global_orient = z(3)
        if create_body_pose:
            self.body_pose = z(63)
        self.left_hand_pose = z(num_pca_comps if use_pca else 45)
        self.right_hand_pose = z(num_pca_comps if use_pca else 45)
        self.jaw_pose = z(3)
        self.leye_pose = z(3)
        self.reye_pose = z(3)
        self.expression = z(ne)

    @torch.no_grad()
    def reset_params(self, **params_dict):
        for name, p in self.named_parameters():
            if name in params_dict:
                p[:] = torch.as_tensor(np.asarray(params_dict[name].detach().cpu())
                                       if torch.is_tensor(params_dict[name])
                                       else params_dict[name], dtype=p.dtype).reshape(p.shape)
            else:
                p.fill_(0)

    def forward(self, return_verts=True, body_pose=None, return_full_pose=False, **kw):
        go, betas, expr = self.global_orient, self.betas, self.expression
        if body_pose is None:
            body_pose = self.body_pose
        lh, rh = self.left_hand_pose, self.right_hand_pose
        if self.use_pca:
            lh = torch.einsum("bi,ij->bj", lh, self.left_hand_components)
            rh = torch.einsum("bi,ij->bj", rh, self.right_hand_components)
        full_pose = torch.cat([go, body_pose, self.jaw_pose, self.leye_pose,
                               self.reye_pose, lh, rh], dim=1)
        full_pose = full_pose + self.pose_mean
        B = full_pose.shape[0]
        coeff = torch.cat([betas, expr], dim=1)
        v_shaped = self.v_template + torch.einsum("bl,mkl->bmk", coeff, self.shapedirs)
        J = torch.einsum("bik,ji->bjk", v_shaped, self.J_regressor)
        R = batch_rodrigues(full_pose.view(-1, 3)).view(B, -1, 3, 3)
        ident = torch.eye(3, dtype=self.dtype, device=R.device)
        pose_feature = (R[:, 1:] - ident).reshape(B, -1)
        v_posed = v_shaped + torch.matmul(pose_feature, self.posedirs).view(B, -1, 3)
        posed_J, A = rigid_transform_chain(R, J, self.parents)
        W = self.lbs_weights.unsqueeze(0).expand(B, -1, -1)
        T = torch.matmul(W, A.view(B, -1, 16)).view(B, -1, 4, 4)
        vh = torch.cat([v_posed, torch.ones([B, v_posed.shape[1], 1], dtype=self.dtype,
                                           device=R.device)], dim=2)
        verts = torch.matmul(T, vh.unsqueeze(-1))[:, :, :3, 0]

        lmk_faces = self.lmk_faces_idx.unsqueeze(0).expand(B, -1)
        lmk_bary = self.lmk_bary_coords.unsqueeze(0).expand(B, -1, -1)
        if self.use_face_contour:
            rows = dynamic_lmk_rows(full_pose.detach(), self.dtype)
            lmk_faces = torch.cat([lmk_faces, self.dynamic_lmk_faces_idx[rows]], dim=1)
            lmk_bary = torch.cat([lmk_bary, self.dynamic_lmk_bary_coords[rows]], dim=1)
        tri = self.faces_tensor[lmk_faces]                                   # [B,L,3]
        off = (torch.arange(B, device=verts.device) * verts.shape[1]).view(B, 1, 1)
        lv = verts.reshape(-1, 3)[tri + off]                                 # [B,L,3,3]
        landmarks = torch.einsum("blfi,blf->bli", lv, lmk_bary)
        joints = torch.cat([posed_J, verts[:, self.extra_vertex_ids], landmarks], dim=1)
        if self.joint_map is not None:
            joints = torch.index_select(joints, 1, self.joint_map)
        return ModelOutput(vertices=verts if return_verts else None, joints=joints,
                           full_pose=full_pose if return_full_pose else None,
                           betas=betas, global_orient=go, body_pose=body_pose,
                           expression=expr, left_hand_pose=lh, right_hand_pose=rh,
                           jaw_pose=self.jaw_pose)
