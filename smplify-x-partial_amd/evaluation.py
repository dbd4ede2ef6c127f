"""Accuracy metrics either side of the fitting path: the alignment / error
utilities of smplifyx/utils.py:540-801 and `compute_v2v` of smplifyx/eval.py:13-45, used to score
fitted meshes against ground truth (EHF).  Host numpy; nothing here runs inside the optimisation.
Same class and function names and return structures as the reference.  The F-score uses a k-d tree
(scipy) for the nearest-neighbour distances the reference obtains from open3d.
"""
from collections import defaultdict

import numpy as np


def _as_np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def mpjpe(input_joints, target_joints):
    """Per-joint Euclidean error, [..., J] (utils.py:597-611)."""
    d = np.asarray(input_joints) - np.asarray(target_joints)
    return np.sqrt((d * d).sum(axis=-1))


def vertex_to_vertex_error(input_vertices, target_vertices):
    return mpjpe(input_vertices, target_vertices)


def point_fscore(pred, gt, thresh):
    """Precision / recall / F of two point sets at a distance threshold (utils.py:622-648)."""
    from scipy.spatial import cKDTree
    pred, gt = _as_np(pred), _as_np(gt)
    gt_to_pred = cKDTree(pred).query(gt)[0]
    pred_to_gt = cKDTree(gt).query(pred)[0]
    recall = (pred_to_gt < thresh).sum() / len(pred_to_gt)
    precision = (gt_to_pred < thresh).sum() / len(gt_to_pred)
    f = 2 * recall * precision / (recall + precision) if recall + precision > 0.0 else 0.0
    return {"fscore": f, "precision": precision, "recall": recall}


def _cols(S1, S2):
    """Points as columns [d, N]; remembers whether the inputs were [N, d]."""
    S1, S2 = np.asarray(S1), np.asarray(S2)
    flip = S1.shape[0] not in (2, 3)
    if flip:
        S1, S2 = S1.T, S2.T
    assert S2.shape[1] == S1.shape[1]
    return S1, S2, flip


class ProcrustesAlignment(object):
    """Similarity transform (scale, rotation, translation) of S1 closest to S2 in the least-squares
    sense (orthogonal Procrustes / Umeyama; utils.py:540-595); returns the transformed S1."""

    def __repr__(self):
        return "ProcrustesAlignment"

    def __call__(self, S1, S2):
        S1, S2, flip = _cols(S1, S2)
        mu1, mu2 = S1.mean(axis=1, keepdims=True), S2.mean(axis=1, keepdims=True)
        X1, X2 = S1 - mu1, S2 - mu2
        var1 = np.sum(X1 ** 2)
        K = X1 @ X2.T
        U, _, Vh = np.linalg.svd(K)
        Z = np.eye(U.shape[0])
        Z[-1, -1] *= np.sign(np.linalg.det(U @ Vh))          # det(R) = +1
        R = Vh.T @ Z @ U.T
        scale = np.trace(R @ K) / var1
        out = scale * (R @ S1) + (mu2 - scale * (R @ mu1))
        return out.T if flip else out


class ScaleAlignment(object):
    """Isotropic scale + translation only (utils.py:729-772)."""

    def __repr__(self):
        return "ScaleAlignment"

    def __call__(self, S1, S2):
        S1, S2, flip = _cols(S1, S2)
        mu1, mu2 = S1.mean(axis=1, keepdims=True), S2.mean(axis=1, keepdims=True)
        scale = np.sqrt(np.sum((S2 - mu2) ** 2) / np.sum((S1 - mu1) ** 2))
        out = scale * S1 + (mu2 - scale * mu1)
        return out.T if flip else out


class PelvisAlignment(object):
    """Subtract the mean of the hip joints (utils.py:650-668)."""

    def __init__(self, hips_idxs=None):
        self.hips_idxs = [2, 3] if hips_idxs is None else hips_idxs

    def align_by_pelvis(self, joints):
        pelvis = joints[self.hips_idxs, :].mean(axis=0, keepdims=True)
        return {"joints": joints - pelvis, "pelvis": pelvis}

    def __call__(self, gt, est):
        return self.align_by_pelvis(gt)["joints"], self.align_by_pelvis(est)["joints"]


class PelvisAlignmentMPJPE(PelvisAlignment):
    def __init__(self, fscore_thresholds=None):
        super(PelvisAlignmentMPJPE, self).__init__()
        self.fscore_thresholds = fscore_thresholds

    def __call__(self, est_points, gt_points):
        gt_al, est_al = super(PelvisAlignmentMPJPE, self).__call__(gt_points, est_points)
        fscore = {t: point_fscore(est_al, gt_points, t) for t in (self.fscore_thresholds or [])}
        return {"point": mpjpe(est_al, gt_al), "fscore": fscore}


class ProcrustesAlignmentMPJPE(ProcrustesAlignment):
    def __init__(self, fscore_thresholds=None):
        super(ProcrustesAlignmentMPJPE, self).__init__()
        self.fscore_thresholds = fscore_thresholds

    def __call__(self, est_points, gt_points):
        aligned = super(ProcrustesAlignmentMPJPE, self).__call__(est_points, gt_points)
        fscore = {t: point_fscore(aligned, gt_points, t) for t in (self.fscore_thresholds or [])}
        return {"point": vertex_to_vertex_error(aligned, gt_points), "fscore": fscore}


def compute_v2v(vertices_fitted, vertices_target, alignments, vids=None):
    """Per-vertex errors of a batch under each alignment (eval.py:13-45):
    {'point': {name: [B, V]}, 'fscore': {name: {thresh: [B]}}}."""
    fitted, target = _as_np(vertices_fitted), _as_np(vertices_target)
    if vids is not None:
        fitted, target = fitted[:, vids], target[:, vids]
    err, fs = {}, {}
    for name, align in alignments.items():
        rows, f = [], defaultdict(list)
        for b in range(target.shape[0]):
            out = align(fitted[b], target[b])
            rows.append(out["point"])
            for t, v in out["fscore"].items():
                f[t].append(np.asarray(v["fscore"]).copy())
        err[name] = np.stack(rows)
        fs[name] = {t: np.stack(v) for t, v in f.items()}
    return {"point": err, "fscore": fs}


def read_ply_vertices(path):
    """xyz of the first element of a binary-little-endian or ascii .ply (the vertices.ply written
    by fit_single_frame, fit_single_frame.py:671-677)."""
    with open(path, "rb") as fh:
        fmt, n, props = None, 0, 0
        first = True
        while True:
            line = fh.readline().decode("ascii").strip()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element") and first:
                n = int(line.split()[2]); first = False
            elif line.startswith("element"):
                first = None
            elif line.startswith("property") and first is False:
                props += 1
            elif line == "end_header":
                break
        if fmt == "binary_little_endian":
            return np.frombuffer(fh.read(n * props * 4), "<f4").reshape(n, props)[:, :3].copy()
        if fmt == "ascii":
            return np.array([[float(v) for v in fh.readline().split()[:3]] for _ in range(n)], np.float32)
        raise ValueError("unsupported ply format %r" % fmt)
