"""Diagnostic: trace of one frame of the benchmark set through the stages (every evaluation)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H, test_gpu_parity as T, bench as BB
from smplifyx_amd import synthetic
i = int(sys.argv[1]) if len(sys.argv) > 1 else 51
g = T._golden("e2e_bench")
cfg = BB.build_cfg("body")
model = synthetic.make_synthetic_model(0)
dm = T._dm(model, cfg)
kp = g["keypoints"].copy(); kp[32:, :, :2] += 0.37
frames = dict(keypoints=kp, reg_pose=g["reg_pose"], reg_global=g["reg_global"], H=600, W=800, focal=5000.0)
fb = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="dense", reuse=True)
fb.guess_init(cfg["body_tri_idxs"])
fb.trace(60000, evaluations=True)
fb.fit()
rec = fb.get_trace()[0]
np.set_printoptions(precision=6, linewidth=200, suppress=False)
print("stage losses", fb.stats()["stage_loss"][0], "evals", fb.stats()["stage_evals"][0])
st = np.flatnonzero(rec[:, 0] == 2)
print("stage records", rec[st])
bad = np.flatnonzero(~np.isfinite(rec).all(1))
print("first non-finite records", bad[:5])
if len(bad):
    print(rec[max(0, bad[0] - 25):bad[0] + 5])
else:
    print(rec[st[-3] - 30:st[-3] + 5])
P = fb.get_params()
print({k: (float(np.abs(v).max()), bool(np.isfinite(v).all())) for k, v in P.items()})
