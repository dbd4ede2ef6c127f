"""Evaluation utilities against the reference's (tests/golden/evaluation.npz, made by
tools/make_goldens.py eval)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from smplifyx_amd import evaluation as E

G = np.load(os.path.join(ROOT, "tests", "golden", "evaluation.npz"))


def test_alignments_and_errors_match_reference():
    for tag in ("j", "v"):
        gt, est = G[tag + "_gt"], G[tag + "_est"]
        assert np.allclose(E.ProcrustesAlignment()(est, gt), G[tag + "_procrustes"], rtol=1e-10, atol=1e-12)
        assert np.allclose(E.ProcrustesAlignment()(est.T.copy(), gt.T.copy()), G[tag + "_procrustes_cols"], rtol=1e-10, atol=1e-12)
        assert np.allclose(E.ScaleAlignment()(est, gt), G[tag + "_scale"], rtol=1e-12)
        a, b = E.PelvisAlignment()(gt, est)
        assert np.array_equal(a, G[tag + "_pelvis_gt"]) and np.array_equal(b, G[tag + "_pelvis_est"])
        assert np.array_equal(E.mpjpe(est, gt), G[tag + "_mpjpe"])
        assert np.array_equal(E.vertex_to_vertex_error(est, gt), G[tag + "_v2v"])
        assert np.allclose(E.PelvisAlignmentMPJPE()(est, gt)["point"], G[tag + "_pelvis_mpjpe"], rtol=1e-12)
        assert np.allclose(E.ProcrustesAlignmentMPJPE()(est, gt)["point"], G[tag + "_procrustes_mpjpe"], rtol=1e-9, atol=1e-12)
        # Procrustes recovers the similarity up to the injected noise
        assert E.ProcrustesAlignmentMPJPE()(est, gt)["point"].mean() < 0.12


def test_compute_v2v_fscore_and_ply(tmp_path):
    gt, est = G["v_gt"], G["v_est"]
    al = {"procrustes": E.ProcrustesAlignmentMPJPE(fscore_thresholds=[0.05, 0.2]), "pelvis": E.PelvisAlignmentMPJPE()}
    out = E.compute_v2v(np.stack([est, est]), np.stack([gt, gt]), al, vids=np.arange(0, 400, 2))
    assert out["point"]["procrustes"].shape == (2, 200) and out["point"]["pelvis"].shape == (2, 200)
    f = out["fscore"]["procrustes"]
    assert set(f) == {0.05, 0.2} and f[0.2].shape == (2,) and np.all(f[0.2] >= f[0.05]) and np.all(f[0.2] <= 1.0)
    assert out["fscore"]["pelvis"] == {}
    same = E.point_fscore(gt, gt, 1e-6)
    assert same == {"fscore": 1.0, "precision": 1.0, "recall": 1.0}
    from smplifyx_amd.fit_single_frame import _write_ply
    _write_ply(str(tmp_path / "v.ply"), gt.astype(np.float32))
    assert np.array_equal(E.read_ply_vertices(str(tmp_path / "v.ply")), gt.astype(np.float32))
