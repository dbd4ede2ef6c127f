"""Diagnostic (GPU box): the SECOND orientation pass of a side-view frame of the configs[4] job, by hand (fit_single_frame.py:527-551):
pass 0 from the camera-stage orientation, pass 1 from that orientation turned by pi about y with the embedding pass 0 left behind.
At the start of every body stage of pass 1 the device closure (loss, gradient, term's share, pair statistics) is compared with the
oracle's closure at the SAME parameters (fp64, with and without the term) -- where do the two part ways?

usage: pen_orient_probe.py [frame of tests/golden/e2e_pen_set.npz, default 92]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench as BB, helpers as H, test_gpu_parity as T
from smplifyx_amd import engine, synthetic, driver

frame = int(sys.argv[1]) if len(sys.argv) > 1 else 92
g = BB.load_pen_golden()
q = int(np.flatnonzero(g["frames"] == frame)[0])
cfg = BB.build_cfg("pen")
model = synthetic.make_topology_model(0)
parts = synthetic.topology_parts()
dm = T._dm(model, cfg)
dm.set_parts(parts["segm"], parts["parents"], cfg["ign_part_pairs"])
faces = np.asarray(model["f"]).astype(np.int64)
kp = g["keypoints"][q:q + 1]
K = kp.shape[1]
jw = np.ones(K, np.float32); jw[cfg["joints_to_ign"]] = 0.0
prep = driver.prepare_frames(cfg, kp, jw)
cam_t = g["cam_prior_t"][q:q + 1].astype(np.float32)
center = np.array([[400.0, 300.0]], np.float32)
frames = dict(keypoints=kp, reg_pose=g["reg_pose"][q:q + 1], reg_global=g["reg_global"][q:q + 1], H=600, W=800, focal=5000.0,
              cam_t=cam_t)


def batch():
    fb = engine.FrameBatch(dm, 1, cfg, lbs_mode="dense", reuse_entry_eval=True, has_regression_pose=True, side_view=False)
    fb.set_frames(prep["keypoints"], prep["jw"], prep["cmask"], 5000.0, center, 1000.0 / 600.0, est_tz=cam_t[:, 2])
    return fb


def oracle(P, stage, with_pen):
    ff_make = H.oracle_frame_fit

    def patched(model_, c, fr, idx, dtype=torch.float64):
        ff = ff_make(model_, c, fr, idx, dtype=dtype)
        if with_pen:
            ff.set_penetration(faces, parts["segm"], parts["parents"], cfg["ign_part_pairs"])
        return ff
    H.oracle_frame_fit = patched
    try:
        return T._oracle_closure(model, cfg, frames, 0, P, stage)
    finally:
        H.oracle_frame_fit = ff_make


np.set_printoptions(linewidth=200, precision=6)
fb = batch()
fb.set_params(regression_pose=frames["reg_pose"], global_orient=frames["reg_global"], pose_embedding=frames["reg_pose"], cam_translation=cam_t)
fb.fit(first_stage=-1, last_stage=-1)
go_cam = fb.get_params()["global_orient"].copy()
fb.fit(first_stage=0, last_stage=fb.n_stages - 1)
p0, s0 = fb.get_params(), fb.stats()
print("pass 0 stage losses", s0["stage_loss"][0], "evals", s0["stage_evals"][0])
print("reference pass 0 / pass 1 (fp32):", np.load(os.path.join(ROOT, "tests/golden/e2e_pen_set.npz"))["f%d_f32_losses_all" % frame])
flip = driver.flipped_orientation(go_cam[0]).astype(np.float32)[None]
fb2 = batch()
fb2.set_params(regression_pose=frames["reg_pose"], global_orient=flip, pose_embedding=p0["pose_embedding"], cam_translation=p0["cam_translation"])
for stage in range(fb2.n_stages):
    P = {k: v.copy() for k, v in fb2.get_params().items() if k != "body_pose"}
    P["est_tz"] = cam_t[:, 2].copy()
    loss, grad = fb2.closure(stage)
    st = fb2.penetration_stats()
    pl = fb2.debug_read("pen_loss")[:, 0]
    lo, go = oracle(P, stage, True)
    lo0, go0 = oracle(P, stage, False)
    print("pass 1, start of stage %d: device loss %.4f (term x weight share %.4f, raw term %.6g) | oracle %.4f (without the term %.4f) | "
          "rel err loss %.2e grad %.2e | |grad| device %.4g oracle %.4g | pairs %s dropped %s overflow %s cut walks %s" %
          (stage, loss[0], loss[0] - lo0, pl[0], lo, lo0, abs(loss[0] - lo) / abs(lo), np.linalg.norm(grad[0] - go) / np.linalg.norm(go),
           np.linalg.norm(grad[0]), np.linalg.norm(go), st["pairs"], st["dropped"], st["entry_overflow"], st["walks_cut"]))
    fb2.trace(20000, evaluations=True)
    fb2.fit(first_stage=stage, last_stage=stage)
    s1 = fb2.stats()
    rec = fb2.get_trace()[0]
    ev = rec[rec[:, 0] == 0]
    print("   stage %d ends at %.4f after %d evaluations; finite %s; trial-loss range in the stage: min %.4g max %.4g; non-finite trial losses %d" %
          (stage, s1["stage_loss"][0, stage + 1], s1["stage_evals"][0, stage + 1], np.isfinite(s1["stage_loss"][0, stage + 1]),
           np.nanmin(ev[:, 2]) if len(ev) else float("nan"), np.nanmax(ev[:, 2]) if len(ev) else float("nan"), int((~np.isfinite(ev[:, 2])).sum())))
    fb2.trace(0)
