"""Diagnostic: closure evaluations per frame of the benchmark sequence against the number of camera-init (torso) keypoints
the synthetic detector dropped -- which frames make the long chains that a lock-step batch (and a rank) waits for."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H, test_gpu_parity as T, bench as BB
from smplifyx_amd import synthetic, driver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
min_cam = int(sys.argv[2]) if len(sys.argv) > 2 else None        # bench.MIN_CAMERA_KEYPOINTS, or nothing: the raw detector
cfg = BB.build_cfg("body")
model = synthetic.make_synthetic_model(0)
dm = T._dm(model, cfg)
jm = H.joint_map_for(cfg)
dev = torch.device("cuda")
def joints_fn(P):
    B = len(P["betas"]); z = lambda k: torch.zeros([B, k], device=dev); t = lambda a: torch.tensor(a, device=dev)
    _, j, _ = dm.lbs_forward(t(P["global_orient"]), t(P["body_pose"]), t(P["betas"]), z(10), z(3), z(3), z(3), z(12), z(12),
                             return_verts=False, return_full_pose=False)
    return j.cpu().numpy()
fr = synthetic.make_frames(n, joints_fn, len(jm), start=0, min_camera_keypoints=min_cam, camera_keypoints=cfg.get("init_joints_idxs", (9, 12, 2, 5)))
jw = np.ones(len(jm), np.float32); jw[cfg["joints_to_ign"]] = 0.0
res = driver.fit_frames(dm, cfg, fr["keypoints"], jw, fr["H"], fr["W"], fr["focal"], reg_pose=fr["reg_pose"], reg_global=fr["reg_global"],
                        lbs_mode="rows")
ev = res["stage_evals"].sum(1)
tri = sorted(set(int(i) for pair in cfg["body_tri_idxs"] for i in pair)) if "body_tri_idxs" in cfg else [2, 5, 9, 12]
init = cfg.get("init_joints_idxs", tri)
present = (fr["keypoints"][:, init, 2] > 0).sum(1)
print("camera-init keypoints", list(init))
for k in range(len(init) + 1):
    m = present == k
    if m.any(): print("  %d present: %4d frames, evaluations mean %.0f  p90 %.0f  max %d" % (k, m.sum(), ev[m].mean(), np.percentile(ev[m], 90), ev[m].max()))
full = present == len(init)
for r in range(n // 256):
    s = slice(r * 256, (r + 1) * 256)
    print("  block %d: max %5d (all frames)  %5d (frames with every camera-init keypoint)  %5d (at most one missing)" % (
        r, ev[s].max(), ev[s][full[s]].max(), ev[s][present[s] >= len(init) - 1].max()))
