"""Diagnostic: a frame whose fit turned non-finite -- where does the first NaN / Inf appear?"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench as BB, _frames as FR
from smplifyx_amd import synthetic
frame = int(sys.argv[1]) if len(sys.argv) > 1 else 198
cfg = BB.build_cfg("body"); m = synthetic.make_synthetic_model(0)
dm, jm = FR.device_model(m, cfg)
fr = FR.frames(dm, jm, frame + 1)
fb = FR.batch(dm, cfg, fr, [frame], lbs_mode="rows", reuse=True)
kp = fr["keypoints"][frame]
print("keypoints with conf > 0:", int((kp[:, 2] > 0).sum()), "of", len(kp), "; init joints", [(j, float(kp[j, 2])) for j in cfg["init_joints_idxs"]])
fb.trace(40000, evaluations=True)
for stage in range(-1, fb.n_stages):
    fb.fit(first_stage=stage, last_stage=stage)
    P = fb.get_params()
    st = fb.stats()
    fin = all(np.isfinite(P[k]).all() for k in P)
    print("stage", stage, "loss", st["stage_loss"][0, stage + 1], "evals", st["stage_evals"][0, stage + 1], "params finite", fin,
          "| max |pose|", np.abs(P["pose_embedding"]).max(), "max |betas|", np.abs(P["betas"]).max(), "cam_t", P["cam_translation"][0], "go", P["global_orient"][0])
    if not fin:
        break
rec = fb.get_trace()[0]
bad = np.flatnonzero(~np.isfinite(rec).all(1))
print("records", len(rec), "first non-finite record", bad[:3])
lo = max(0, (bad[0] if len(bad) else len(rec)) - 40)
np.set_printoptions(precision=6, suppress=False, linewidth=200)
print(rec[lo:lo + 50])
