"""Mean closure evaluations per stage of a bench workload (which stages the time goes to).  usage: stage_evals.py [body|full|pen] [B]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from smplifyx_amd import engine, synthetic, utils as U, driver
which = sys.argv[1] if len(sys.argv) > 1 else 'full'; B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = bench.build_cfg(which); full = which == 'full'; pen = which == 'pen'
model = synthetic.make_synthetic_model(0, surface=pen)
jm = U.smpl_to_annotation("smplx", use_hands=cfg["use_hands"], use_face=cfg["use_face"], use_face_contour=cfg["use_face_contour"], format=cfg["format"])
dm = engine.DeviceModel(model, joint_map=jm, num_betas=cfg["num_betas"], num_expression_coeffs=cfg["num_expression_coeffs"],
                        num_pca_comps=cfg["num_pca_comps"], use_face_contour=cfg["use_face_contour"], vposer=synthetic.make_synthetic_vposer(0) if full else None)
if pen:
    parts = synthetic.make_synthetic_parts(model); dm.set_parts(parts["segm"], parts["parents"], cfg["ign_part_pairs"])
dev = torch.device("cuda")
def joints_fn(P):
    z = lambda n: torch.zeros([B, n], device=dev); t = lambda a: torch.tensor(a, device=dev)
    _, j, _ = dm.lbs_forward(t(P["global_orient"]), t(P["body_pose"]), t(P["betas"]), z(10), z(3), z(3), z(3), z(12), z(12), return_verts=False, return_full_pose=False)
    return j.cpu().numpy()
fr = synthetic.make_frames(B, joints_fn, len(jm), focal=float(cfg.get("focal_length") or 5000.0))
jw = np.ones(len(jm), np.float32); jw[cfg["joints_to_ign"]] = 0
ct = cc = None
if pen:
    rng = np.random.RandomState(1000); ct = (fr["cam_t"] + 0.05 * rng.normal(size=fr["cam_t"].shape)).astype(np.float32)
    cc = np.tile(np.array([fr["W"] * 0.5, fr["H"] * 0.5], np.float32), (B, 1))
res = driver.fit_frames(dm, cfg, fr["keypoints"], jw, fr["H"], fr["W"], fr["focal"], reg_pose=None if full else fr["reg_pose"],
                        reg_global=None if full else fr["reg_global"], cam_prior_t=ct, cam_prior_center=cc, lbs_mode="dense", reuse_entry_eval=True)
ev = res["stage_evals"]
print(which, "B", B, "mean evals per stage (camera, body stages...):", ev.mean(0).round(1).tolist(), "max per stage", ev.max(0).tolist(), "total mean", ev.sum(1).mean(), "max", ev.sum(1).max())

# lock-step occupancy by stage: in round r frame f is in the stage its cumulative evaluation count has reached
cum = np.cumsum(ev, 1)
R = int(cum[:, -1].max())
ns = ev.shape[1]
any_in = np.zeros(ns, np.int64); frames_in = np.zeros(ns, np.float64); act = 0.0
for r in range(R):
    st = (cum <= r).sum(1)              # stage index of each frame in round r (ns = finished)
    for q in range(ns):
        n = int((st == q).sum())
        any_in[q] += n > 0; frames_in[q] += n
    act += (st < ns).sum()
print("rounds", R, "mean active frames/round %.1f" % (act / R), "| rounds with >= 1 frame in stage q:", any_in.tolist(),
      "| mean frames in stage q over those rounds:", [round(frames_in[q] / max(any_in[q], 1), 1) for q in range(ns)])
# how many rounds run with how many frames left (the per-round floors of both kernels are paid by every round)
tot = cum[:, -1]
n_act = np.array([(tot > r).sum() for r in range(R)])
edges = [1, 8, 16, 32, 64, 96, 128, 192, 256, 512, 1024, 1 << 30]
print("rounds by active frames:", {"%d-%d" % (lo, hi - 1): int(((n_act >= lo) & (n_act < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:]) if ((n_act >= lo) & (n_act < hi)).any()})
