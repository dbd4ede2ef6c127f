"""Shared builders for the parity tests (oracle side = checker, engine side = product)."""
import os

import numpy as np
import torch

from smplifyx_amd import cmd_parser, synthetic, utils as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def has_lab():
    """Is the library under test the LAB build (include/sfx_lab.h: A/B forms, environment switches, phase clocks)?  The product
    library (libsfx.so) has none of them; tools/run_gpu_suite.sh runs the suite once on each."""
    from smplifyx_amd import _capi
    try:
        return _capi.has_lab()
    except Exception:
        return False


def requires_lab():
    import pytest
    return pytest.mark.skipif(not has_lab(), reason="needs the lab build (SFX_LAB=1 csrc/build.sh; SFX_LIB=.../libsfx_lab.so; "
                                                    "tools/run_gpu_suite.sh)")


CFG_DIR = os.path.join(ROOT, "cfg_files")


def load_cfg(name, **over):
    base = dict(visualize=False, interactive=False, interpenetration=False, save_vertices=False,
                use_gender_classifier=False)
    base.update(over)
    return cmd_parser.load_config(os.path.join(CFG_DIR, name), base)


def joint_map_for(cfg):
    return U.smpl_to_annotation("smplx", use_hands=cfg["use_hands"], use_face=cfg["use_face"],
                                use_face_contour=cfg["use_face_contour"], format=cfg["format"])


def base_joint_weights(cfg, K):
    """COCO25.get_joint_weights (smplifyx/data_parser.py:159-171)."""
    w = np.ones(K, np.float32)
    ign = cfg.get("joints_to_ign")
    if ign is not None and -1 not in ign:
        w[ign] = 0.0
    return w


def random_params(rng, B, scale=1.0, nb=10, ne=10, npca=12):
    return dict(
        global_orient=(0.4 * scale * rng.normal(size=(B, 3))).astype(np.float32),
        pose_embedding=(0.3 * scale * rng.normal(size=(B, 63))).astype(np.float32),
        betas=(scale * rng.normal(size=(B, nb))).astype(np.float32),
        expression=(scale * rng.normal(size=(B, ne))).astype(np.float32),
        jaw_pose=(0.2 * scale * rng.normal(size=(B, 3))).astype(np.float32),
        leye_pose=(0.1 * scale * rng.normal(size=(B, 3))).astype(np.float32),
        reye_pose=(0.1 * scale * rng.normal(size=(B, 3))).astype(np.float32),
        left_hand_pose=(scale * rng.normal(size=(B, npca))).astype(np.float32),
        right_hand_pose=(scale * rng.normal(size=(B, npca))).astype(np.float32),
    )


def oracle_model(model, cfg, dtype=torch.float32):
    from oracle.body_model import SMPLXRef
    return SMPLXRef(model, joint_map=joint_map_for(cfg), num_betas=cfg["num_betas"],
                    num_expression_coeffs=cfg["num_expression_coeffs"], num_pca_comps=cfg["num_pca_comps"],
                    use_face_contour=cfg["use_face_contour"], create_body_pose=not cfg["use_vposer"], dtype=dtype)


def oracle_joints_fn(model, cfg):
    bm = oracle_model(model, cfg, torch.float64)

    def fn(P):
        n = P["global_orient"].shape[0]
        out = []
        for i in range(n):
            bm.reset_params(global_orient=P["global_orient"][i:i + 1], betas=P["betas"][i:i + 1])
            with torch.no_grad():
                o = bm(return_verts=False, body_pose=torch.tensor(P["body_pose"][i:i + 1], dtype=torch.float64))
            out.append(o.joints[0].numpy())
        return np.stack(out)
    return fn


def oracle_frame_fit(model, cfg, frames, i, dtype=torch.float32, **kw):
    """oracle.fit_frame.FrameFit for synthetic frame i (regression prior = noisy truth; with
    use_vposer: zero latent, synthetic VPoser decoder, no regression prior)."""
    from oracle.fit_frame import FrameFit
    bm = oracle_model(model, cfg, dtype)
    K = frames["keypoints"].shape[1]
    if cfg.get("use_vposer"):
        from oracle.vposer import VPoserRef
        vp = VPoserRef(synthetic.make_synthetic_vposer(0), dtype)
        return FrameFit(bm, frames["keypoints"][i:i + 1], frames["H"], frames["W"], frames["focal"], cfg,
                        base_joint_weights(cfg, K), vposer=vp, dtype=dtype, **kw)
    return FrameFit(bm, frames["keypoints"][i:i + 1], frames["H"], frames["W"], frames["focal"], cfg,
                    base_joint_weights(cfg, K), reg_pose=frames["reg_pose"][i], reg_global=frames["reg_global"][i],
                    dtype=dtype, **kw)


def engine_batch_from_frames(dm, cfg, frames, idx, lbs_mode="rows", reuse=False):
    """FrameBatch prepared the way fit_single_frame.py:209-294,358-411 prepares one frame."""
    from smplifyx_amd import engine
    idx = list(idx)
    B = len(idx)
    kp = frames["keypoints"][idx]
    K = kp.shape[1]
    nb = engine.NUM_BODY_JOINTS[cfg["format"]]
    thr = np.array([cfg.get("confidence_threshold", 0)] * nb + [0] * 110)[:K]
    jw = np.tile(base_joint_weights(cfg, K), (B, 1))
    low = kp[:, :, 2] < thr[None, :]
    jw[low] = 0
    cmask = np.zeros((B, K), np.float32)
    for b in range(B):
        for j in cfg["init_joints_idxs"]:
            if kp[b, j, 0] != 0 and kp[b, j, 1] != 0 and not low[b, j]:
                cmask[b, j] = 1
    vp = bool(cfg.get("use_vposer"))
    fb = engine.FrameBatch(dm, B, cfg, lbs_mode=lbs_mode, reuse_entry_eval=reuse, has_regression_pose=not vp)
    H, W = frames["H"], frames["W"]
    fb.set_frames(kp, jw, cmask, frames["focal"], np.tile([W * 0.5, H * 0.5], (B, 1)), 1000.0 / H)
    if vp:
        fb.set_params(pose_embedding=np.zeros((B, fb.nemb), np.float32), global_orient=np.zeros((B, 3), np.float32),
                      cam_translation=np.zeros((B, 3), np.float32))
    else:
        fb.set_params(regression_pose=frames["reg_pose"][idx], global_orient=frames["reg_global"][idx],
                      pose_embedding=frames["reg_pose"][idx], cam_translation=np.zeros((B, 3), np.float32))
    return fb


# ---------------------------------------------------------------------------------------------------------
# Closure-level parity bounds (loss 1e-5; north_star: 1e-4 on the gradient), shared by every
# closure-vs-oracle comparison of the GPU suite, __graft_entry__.smoke() and bench.py's closure_parity object.
# Every comparison is recorded in PARITY_LOG; conftest prints the observed maxima per (label, stage) at the end
# of a test session, so the log of a run shows the measurement next to the bound that protects it.
# Measured on MI355X (round 3, every closure test of the suite): loss <= 3.2e-7, gradient <= 5.2e-7.  The asserted bounds are
# ~10 x those maxima -- far inside SURVEY 8(d)'s 1e-5 and north_star's 1e-4, which an assertion must not merely repeat.
CLOSURE_LOSS_TOL = 4e-6
CLOSURE_GRAD_TOL = 6e-6
PARITY_LOG = {}


def closure_errors(loss, lo, grad, go):
    """(relative loss error, relative gradient error in the 2-norm) of one frame's closure result against the oracle's."""
    le = abs(float(loss) - float(lo)) / max(abs(float(lo)), 1e-30)
    ge = float(np.linalg.norm(np.asarray(grad, np.float64) - np.asarray(go, np.float64)) / max(np.linalg.norm(go), 1e-30))
    return le, ge


def check_closure(label, stage, loss, lo, grad, go, loss_tol=CLOSURE_LOSS_TOL, grad_tol=CLOSURE_GRAD_TOL):
    """Record and assert one closure comparison (HIP result vs fp64 autograd of the oracle)."""
    le, ge = closure_errors(loss, lo, grad, go)
    e = PARITY_LOG.setdefault((label, int(stage)), [0.0, 0.0, 0, loss_tol, grad_tol])
    e[0] = max(e[0], le); e[1] = max(e[1], ge); e[2] += 1
    assert le <= loss_tol, ("closure loss", label, stage, float(loss), float(lo), le, loss_tol)
    assert ge <= grad_tol, ("closure gradient", label, stage, ge, grad_tol)
    return le, ge


TERM_LOG = {}


def check_bound(label, what, err, bound):
    """Record and assert one relative error of the interpenetration term (or any other quantity outside check_closure's
    loss / gradient pair): the session summary prints the observed maximum next to the bound that protects it."""
    e = TERM_LOG.setdefault((label, what), [0.0, 0, bound])
    e[0] = max(e[0], float(err)); e[1] += 1
    assert err <= bound, (label, what, float(err), bound)
    return err


def parity_log_lines():
    out = []
    for (label, what), (err, n, bound) in sorted(TERM_LOG.items()):
        out.append("term parity    %-28s %-34s rel err max %.2e (bound %.0e)  [%d checks]" % (label, what, err, bound, n))
    for (label, stage), (le, ge, n, lt, gt) in sorted(PARITY_LOG.items()):
        out.append("closure parity %-28s stage %2d: loss rel err max %.2e (bound %.0e)  gradient rel err max %.2e (bound %.0e)  [%d frames]"
                   % (label, stage, le, lt, ge, gt, n))
    return out
