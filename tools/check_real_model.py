#!/usr/bin/env python
"""Optional external anchor for the SMPL-X forward (skipped by default: needs a licensed model file).

The reference's demo/ExPose_results/*/*_params.npz hold ExPose's own evaluation of SMPL-X: shape / expression
coefficients, rotation matrices of every joint, and the resulting `vertices [10475,3]` and `joints [144,3]` -- the only
known-answer vector for smplx.lbs that exists in the reference tree.  Given a user's SMPLX_{NEUTRAL,MALE,FEMALE}.npz this
script feeds those parameters to sfx_lbs_forward (the HIP path, through the C ABI) and reports max |delta vertices| and
max |delta joints|.  ExPose does not record which model / gender produced its results, so a mismatch of centimetres
means "other model file", agreement to ~1e-5 m pins the LBS restatement (row a6 of SURVEY.md 8).

    python tools/check_real_model.py --model /path/to/models/smplx/SMPLX_NEUTRAL.npz \\
                                     [--expose-dir /root/reference/demo/ExPose_results]

Needs a GPU (the product path has no CPU fallback).  Nothing here is imported by the package, the tests or bench.py.
"""
import argparse
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rotmat_to_aa(R):
    """Log map of rotation matrices [..., 3, 3] -> axis-angle [..., 3] (angle in [0, pi])."""
    R = np.asarray(R, np.float64)
    c = np.clip((np.trace(R, axis1=-2, axis2=-1) - 1.0) / 2.0, -1.0, 1.0)
    ang = np.arccos(c)
    v = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1)
    s = np.linalg.norm(v, axis=-1, keepdims=True)
    out = np.where(s > 1e-12, v / np.maximum(s, 1e-300) * ang[..., None], 0.0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True, help="SMPLX_{NEUTRAL,MALE,FEMALE}.npz of the licensed SMPL-X release")
    ap.add_argument("--expose-dir", default=os.path.join(os.environ.get("SFX_REFERENCE_ROOT", "/root/reference"), "demo", "ExPose_results"))
    args = ap.parse_args()
    import torch
    from smplifyx_amd import engine
    if not torch.cuda.is_available():
        sys.exit("no GPU visible: sfx_lbs_forward has no CPU fallback")
    files = sorted(glob.glob(os.path.join(args.expose_dir, "*", "*_params.npz")))
    if not files:
        sys.exit("no *_params.npz under %s" % args.expose_dir)
    model = dict(np.load(args.model, allow_pickle=True))
    # ExPose stores full 15-joint hand rotations: identity "PCA" basis of 45 components, flat hand mean
    model["hands_componentsl"] = np.eye(45, dtype=np.float32)
    model["hands_componentsr"] = np.eye(45, dtype=np.float32)
    dm = engine.DeviceModel(model, num_betas=10, num_expression_coeffs=10, num_pca_comps=45, flat_hand_mean=True,
                            use_face_contour=True)
    dev = torch.device("cuda")
    t = lambda a: torch.tensor(np.asarray(a, np.float32).reshape(1, -1), device=dev)
    worst_v = worst_j = 0.0
    for f in files:
        d = np.load(f, allow_pickle=True)
        aa = lambda k: rotmat_to_aa(d[k]).reshape(-1)
        verts, joints, _ = dm.lbs_forward(t(aa("global_orient")), t(aa("body_pose")), t(d["betas"]), t(d["expression"]),
                                          t(aa("jaw_pose")), t(np.zeros(3)), t(np.zeros(3)), t(aa("left_hand_pose")),
                                          t(aa("right_hand_pose")))
        v, j = verts[0].cpu().numpy().astype(np.float64), joints[0].cpu().numpy().astype(np.float64)
        ev, ej = np.asarray(d["vertices"], np.float64), np.asarray(d["joints"], np.float64)
        n = min(len(j), len(ej))
        dv, dj = np.abs(v - ev).max(), np.abs(j[:n] - ej[:n]).max()
        # ExPose may report the mesh relative to another origin: also after aligning the pelvis joints
        dv_al = np.abs((v - j[0]) - (ev - ej[0])).max()
        dj_al = np.abs((j[:n] - j[0]) - (ej[:n] - ej[0])).max()
        print("%s: max |d vertices| %.3e m (pelvis-aligned %.3e), max |d joints| %.3e m (pelvis-aligned %.3e), %d joints compared"
              % (os.path.basename(f), dv, dv_al, dj, dj_al, n))
        worst_v, worst_j = max(worst_v, min(dv, dv_al)), max(worst_j, min(dj, dj_al))
    print("worst: vertices %.3e m, joints %.3e m -> %s" % (
        worst_v, worst_j, "LBS restatement PINNED by ExPose's own evaluation" if max(worst_v, worst_j) < 1e-4
        else "differs (another model file / gender than ExPose used, or a real discrepancy: compare several model files)"))


if __name__ == "__main__":
    main()
