"""Side measurement (BASELINE configs[4]-like): cfg_files/fit_smplx_combined_halpe.yaml with the
interpenetration term, synthetic frames / model / part labels.  Not the bench headline."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import _frames as FR
from smplifyx_amd import driver, synthetic, engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
body_only = (len(sys.argv) > 2 and sys.argv[2] == "body")
over = dict(interpenetration=True)
if body_only: over.update(use_hands=False, use_face=False)
cfg = FR.load_cfg("fit_smplx_combined_halpe.yaml", **over)
cfg["use_camera_prior"] = False
m = synthetic.make_synthetic_model(0, surface=("soup" not in sys.argv))
parts = synthetic.make_synthetic_parts(m)
dm, jm = FR.device_model(m, cfg)
dm.set_parts(parts["segm"], parts["parents"], cfg["ign_part_pairs"])
K = len(jm)
frames = FR.frames(dm, jm, B)
jw = FR.joint_weights(cfg, K)
for pen in (False, True):
    c = dict(cfg); c["interpenetration"] = pen
    engine.prof_enable(True, every=4); engine.prof_reset()
    torch.cuda.synchronize(); t0 = time.time()
    res = driver.fit_frames(dm, c, frames["keypoints"], jw, 600, 800, 5000.0, reg_pose=frames["reg_pose"], reg_global=frames["reg_global"], lbs_mode="dense")
    torch.cuda.synchronize(); dt = time.time() - t0
    engine.prof_enable(False)
    print("interpenetration=%s  B=%d K=%d: %.2f s -> %.1f frames/s; evals/frame mean %.0f; kernels us: lbs %.0f tick %.0f pen %.0f" % (
        pen, B, K, dt, B / dt, res["stage_evals"].sum(1).mean(),
        1e3 * engine.prof_get("lbs_dense")[0] / max(1, engine.prof_get("lbs_dense")[1]),
        1e3 * engine.prof_get("tick")[0] / max(1, engine.prof_get("tick")[1]),
        1e3 * engine.prof_get("penetration")[0] / max(1, engine.prof_get("penetration")[1])))
