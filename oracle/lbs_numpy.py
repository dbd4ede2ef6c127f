"""ORACLE (test infrastructure): independent fp64 numpy implementation of the SMPL-X
forward, written joint-by-joint with explicit loops so that it shares no code path with
oracle/body_model.py.  Used only to cross-check that restatement (the LBS itself is
"parity unpinned": the `smplx` package is absent from /root/reference).
Algorithm: SURVEY.md 3.4 / appendix A.2.
"""
import numpy as np

NECK_CHAIN = (12, 9, 6, 3, 0)


def rodrigues(theta, eps=1e-8):
    th = np.asarray(theta, np.float64)
    a = np.sqrt(((th + eps) ** 2).sum())
    d = th / a
    K = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def forward(model, params, num_betas=10, num_expr=10, num_pca=12, use_face_contour=True,
            flat_hand_mean=False, joint_map=None):
    """params: dict of 1-D arrays (global_orient, body_pose, betas, expression, jaw_pose,
    leye_pose, reye_pose, left_hand_pose[num_pca], right_hand_pose[num_pca]).
    Returns dict(vertices[V,3], joints[K,3], full_pose[165], posed_joints[55,3], lut_row)."""
    g = lambda k, n: np.asarray(params.get(k, np.zeros(n)), np.float64).reshape(-1)
    lh = g("left_hand_pose", num_pca) @ np.asarray(model["hands_componentsl"], np.float64)[:num_pca]
    rh = g("right_hand_pose", num_pca) @ np.asarray(model["hands_componentsr"], np.float64)[:num_pca]
    full = np.concatenate([g("global_orient", 3), g("body_pose", 63), g("jaw_pose", 3),
                           g("leye_pose", 3), g("reye_pose", 3), lh, rh])
    if not flat_hand_mean:
        full[75:120] += np.asarray(model["hands_meanl"], np.float64)
        full[120:165] += np.asarray(model["hands_meanr"], np.float64)
    sd = np.asarray(model["shapedirs"], np.float64)
    es = 300 if sd.shape[-1] >= 400 else 10
    coeff = np.concatenate([g("betas", num_betas), g("expression", num_expr)])
    sdirs = np.concatenate([sd[:, :, :num_betas], sd[:, :, es:es + num_expr]], -1)
    v_shaped = np.asarray(model["v_template"], np.float64) + sdirs @ coeff
    J = np.asarray(model["J_regressor"], np.float64) @ v_shaped
    nj = J.shape[0]
    R = [rodrigues(full[3 * i:3 * i + 3]) for i in range(nj)]
    feat = np.concatenate([(R[i] - np.eye(3)).reshape(-1) for i in range(1, nj)])
    pd = np.asarray(model["posedirs"], np.float64)           # [V,3,486]
    v_posed = v_shaped + pd @ feat
    parents = np.asarray(model["kintree_table"])[0].astype(int)
    G = [None] * nj
    for i in range(nj):
        M = np.eye(4)
        M[:3, :3] = R[i]
        M[:3, 3] = J[i] - (J[parents[i]] if i > 0 else 0)
        G[i] = M if i == 0 else G[parents[i]] @ M
    posed = np.stack([Gi[:3, 3] for Gi in G])
    A = []
    for i in range(nj):
        Ai = G[i].copy()
        Ai[:3, 3] = G[i][:3, 3] - G[i][:3, :3] @ J[i]
        A.append(Ai)
    A = np.stack(A)                                           # [55,4,4]
    W = np.asarray(model["weights"], np.float64)
    T = np.tensordot(W, A, axes=(1, 0))                       # [V,4,4]
    verts = np.einsum("vij,vj->vi", T[:, :3, :3], v_posed) + T[:, :3, 3]
    faces = np.asarray(model["f"]).astype(np.int64)
    lf = list(np.asarray(model["lmk_faces_idx"]).astype(int))
    lb = [np.asarray(b, np.float64) for b in np.asarray(model["lmk_bary_coords"])]
    row = -1
    if use_face_contour:
        rel = np.eye(3)
        for j in NECK_CHAIN:
            rel = rodrigues(full[3 * j:3 * j + 3]) @ rel
        ang = -np.arctan2(-rel[2, 0], np.sqrt(rel[0, 0] ** 2 + rel[1, 0] ** 2)) * 180.0 / np.pi
        y = int(np.round(min(ang, 39.0)))                     # numpy rounds half to even
        if y < -39:
            row = 78
        elif y < 0:
            row = 39 - y
        else:
            row = y
        lf += list(np.asarray(model["dynamic_lmk_faces_idx"])[row].astype(int))
        lb += [np.asarray(b, np.float64) for b in np.asarray(model["dynamic_lmk_bary_coords"])[row]]
    lm = np.stack([sum(lb[l][k] * verts[faces[lf[l], k]] for k in range(3)) for l in range(len(lf))])
    joints = np.concatenate([posed, verts[np.asarray(model["extra_vertex_ids"]).astype(int)], lm])
    if joint_map is not None:
        joints = joints[np.asarray(joint_map).astype(int)]
    return dict(vertices=verts, joints=joints, full_pose=full, posed_joints=posed, lut_row=row,
                v_posed=v_posed, A=A, J=J)
