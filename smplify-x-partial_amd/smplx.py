"""Drop-in for the part of the external `smplx` package the reference's hot path uses
(reference call sites smplifyx/main.py:109-127, fitting.py:82,248, fit_single_frame.py:274,
551,611, camera.py:27): `create()`, `SMPLX` with `.parameters()`, `.reset_params()`,
`.forward(return_verts, body_pose, return_full_pose)`, `.faces_tensor`, `.faces`, and
`lbs.transform_mat`.  The forward is the HIP dense-LBS path of libsfx.so; torch tensors are
containers only (no autograd graph: gradients come from the engine's hand-written adjoint,
see fitting.py).  Model file keys: SURVEY.md appendix A.1.
"""
import os
import types
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from . import engine
from .synthetic import SMPLX_EXTRA_VERTEX_IDS

_ModelOutput = namedtuple("ModelOutput",
                          ["vertices", "joints", "full_pose", "betas", "global_orient", "body_pose",
                           "expression", "left_hand_pose", "right_hand_pose", "jaw_pose"])
_ModelOutput.__new__.__defaults__ = (None,) * len(_ModelOutput._fields)


class ModelOutput(_ModelOutput):
    """smplx.ModelOutput (a namedtuple).  Outputs of SMPLX.forward below also remember the model and the exact parameter
    tensors they were made from (`_model`, `_inputs`): the stand-alone SMPLifyLoss.forward evaluates the device objective there."""


SMPLXOutput = ModelOutput


def _transform_mat(R, t):
    """[[R, t], [0 0 0 1]] (smplx.lbs.transform_mat)."""
    return torch.cat([torch.nn.functional.pad(R, [0, 0, 0, 1]),
                      torch.nn.functional.pad(t, [0, 0, 0, 1], value=1)], dim=2)


lbs = types.SimpleNamespace(transform_mat=_transform_mat)


class SMPLX(nn.Module):
    NUM_JOINTS = 54
    NUM_BODY_JOINTS = 21

    def __init__(self, model_path, joint_mapper=None, create_global_orient=True, create_body_pose=True,
                 create_betas=True, create_left_hand_pose=True, create_right_hand_pose=True,
                 create_expression=True, create_jaw_pose=True, create_leye_pose=True, create_reye_pose=True,
                 create_transl=False, dtype=torch.float32, batch_size=1, use_pca=True, num_pca_comps=6,
                 flat_hand_mean=False, num_betas=10, num_expression_coeffs=10, use_face_contour=False,
                 gender="neutral", ext="npz", vposer=None, **kwargs):
        super().__init__()
        if dtype not in (torch.float32, torch.float64):
            raise ValueError("Unknown float type {}".format(dtype))
        # dtype float64 (cfg float_dtype, main.py:99-105): parameters and outputs are float64 CONTAINERS; the engine's
        # arithmetic is its own (fp32 parameters / reverse sweep / optimiser, fp64 keypoint forward, and -- in batches created
        # with float_dtype float64 -- fp64 projection as well: sfx_batch_cfg.high_precision)
        if not use_pca:
            num_pca_comps = 45          # the hand pose parameters are the 45 axis-angle values (cmd_parser.py:127)
        if create_transl:
            raise NotImplementedError("create_transl=True is not used by the reference (main.py:120)")
        if isinstance(model_path, dict):
            data = model_path
        else:
            fn = model_path
            if os.path.isdir(fn):
                fn = os.path.join(model_path, "SMPLX_{}.{}".format(gender.upper(), ext))
            data = dict(np.load(fn, allow_pickle=True, encoding="latin1"))
        self._model_data = data
        self.dtype = dtype
        self.batch_size = batch_size
        self.use_pca, self.num_pca_comps = use_pca, num_pca_comps
        self.num_betas, self.num_expression_coeffs = num_betas, num_expression_coeffs
        self.use_face_contour, self.flat_hand_mean = use_face_contour, flat_hand_mean
        self.joint_mapper = joint_mapper
        jm = None
        if joint_mapper is not None and getattr(joint_mapper, "joint_maps", None) is not None:
            jm = joint_mapper.joint_maps.detach().cpu().numpy()
        self._cfg = dict(joint_map=jm, num_betas=num_betas, num_expression_coeffs=num_expression_coeffs,
                         num_pca_comps=num_pca_comps, flat_hand_mean=flat_hand_mean, use_face_contour=use_face_contour,
                         extra_vertex_ids=data.get("extra_vertex_ids", SMPLX_EXTRA_VERTEX_IDS), vposer=vposer, use_pca=use_pca)
        self._dm = None
        self.faces = np.asarray(data["f"]).astype(np.int64)
        self.register_buffer("faces_tensor", torch.as_tensor(self.faces, dtype=torch.long))
        B = batch_size
        z = lambda n: nn.Parameter(torch.zeros([B, n], dtype=dtype), requires_grad=True)
        # registration order of smplx.SMPL / SMPLH / SMPLX.__init__  [external]
        if create_betas: self.betas = z(num_betas)
        if create_global_orient: self.global_orient = z(3)
        if create_body_pose: self.body_pose = z(63)
        if create_left_hand_pose: self.left_hand_pose = z(num_pca_comps)
        if create_right_hand_pose: self.right_hand_pose = z(num_pca_comps)
        if create_jaw_pose: self.jaw_pose = z(3)
        if create_leye_pose: self.leye_pose = z(3)
        if create_reye_pose: self.reye_pose = z(3)
        if create_expression: self.expression = z(num_expression_coeffs)

    # ---- engine handle (one per process/GPU, created on first use) ----------------------------
    @property
    def device_model(self):
        if self._dm is None:
            if not torch.cuda.is_available():
                raise RuntimeError("SMPLX.forward needs a GPU: the HIP path has no CPU fallback")
            self._dm = engine.DeviceModel(self._model_data, **self._cfg)
        return self._dm

    @torch.no_grad()
    def reset_params(self, **params_dict):
        """Copy params_dict[name] where given, zero-fill every other parameter."""
        for name, p in self.named_parameters():
            if name in params_dict:
                v = params_dict[name]
                v = v.detach() if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))
                p[:] = v.to(device=p.device, dtype=p.dtype).reshape(p.shape)
            else:
                p.fill_(0)

    def _p(self, name, n, dev):
        p = getattr(self, name, None)
        return p.detach() if p is not None else torch.zeros([self.batch_size, n], dtype=self.dtype, device=dev)

    def forward(self, betas=None, global_orient=None, body_pose=None, left_hand_pose=None, right_hand_pose=None,
                expression=None, jaw_pose=None, leye_pose=None, reye_pose=None, return_verts=True,
                return_full_pose=False, **kwargs):
        dm = self.device_model
        dev = self.faces_tensor.device
        if dev.type != "cuda":
            raise RuntimeError("move the model to the GPU first (.to('cuda')): no CPU fallback")
        pick = lambda v, name, n: (v.detach() if v is not None else self._p(name, n, dev)).to(dev, torch.float32)
        go = pick(global_orient, "global_orient", 3)
        bp = pick(body_pose, "body_pose", 63).reshape(go.shape[0], -1)
        be = pick(betas, "betas", self.num_betas)
        ex = pick(expression, "expression", self.num_expression_coeffs)
        jw, le, re = pick(jaw_pose, "jaw_pose", 3), pick(leye_pose, "leye_pose", 3), pick(reye_pose, "reye_pose", 3)
        lh = pick(left_hand_pose, "left_hand_pose", self.num_pca_comps)
        rh = pick(right_hand_pose, "right_hand_pose", self.num_pca_comps)
        verts, joints, full_pose = dm.lbs_forward(go, bp, be, ex, jw, le, re, lh, rh, return_verts=return_verts,
                                                  return_full_pose=True)
        if self.dtype != torch.float32:         # float64 containers
            cast = lambda x: x.to(self.dtype) if x is not None else None
            verts, joints, full_pose, be, go, bp, ex, jw = (cast(x) for x in (verts, joints, full_pose, be, go, bp, ex, jw))
        inputs = dict(global_orient=go, body_pose=bp, betas=be, expression=ex, jaw_pose=jw, leye_pose=le, reye_pose=re,
                      left_hand_pose=lh, right_hand_pose=rh)
        out = ModelOutput(vertices=verts if return_verts else None, joints=joints,
                           full_pose=full_pose if return_full_pose else None, betas=be, global_orient=go,
                           body_pose=bp, expression=ex,
                           left_hand_pose=full_pose[:, 75:120] - torch.as_tensor(dm_pose_mean(self, 75), device=dev, dtype=full_pose.dtype),
                           right_hand_pose=full_pose[:, 120:165] - torch.as_tensor(dm_pose_mean(self, 120), device=dev, dtype=full_pose.dtype),
                           jaw_pose=jw)
        out._model, out._inputs = self, inputs
        return out


def dm_pose_mean(model, start):
    d = model._model_data
    if model.flat_hand_mean:
        return np.zeros(45, np.float32)
    return np.asarray(d["hands_meanl" if start == 75 else "hands_meanr"], np.float32)


def create(model_path, model_type="smplx", **kwargs):
    """smplx.create: `model_path` is the models folder (file model_path/smplx/SMPLX_{GENDER}.npz,
    main.py:264), a direct .npz path, or an in-memory model dict (synthetic)."""
    if model_type.lower() != "smplx":
        raise ValueError("Unknown model type {}, exiting!".format(model_type))
    if isinstance(model_path, str) and os.path.isdir(model_path):
        sub = os.path.join(model_path, "smplx")
        if os.path.isdir(sub):
            model_path = sub
    return SMPLX(model_path, **kwargs)
