"""Diagnostic (GPU box): frame `i` of the `--workload pen` sequence (frames made with the device forward, as bench.py makes them)
fitted alone with the optimiser trace attached: where does the first non-finite number appear, and what preceded it?
usage: pen_nan_probe.py [frame, default 92]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import _frames as TF
from smplifyx_amd import engine, synthetic, driver
i = int(sys.argv[1]) if len(sys.argv) > 1 else 92
cfg = TF.load_cfg("fit_smplx_combined_halpe.yaml", interpenetration=True)
model = synthetic.make_topology_model(0); parts = synthetic.topology_parts()
dm, jm = TF.device_model(model, cfg)
dm.set_parts(parts["segm"], parts["parents"], cfg["ign_part_pairs"])
fr = TF.frames(dm, jm, i + 1)
jw = TF.joint_weights(cfg, len(jm))
cam_t = (fr["cam_t"] + 0.05 * np.random.RandomState(1000).normal(size=fr["cam_t"].shape)).astype(np.float32)
sel = slice(i, i + 1)
fb, prep = driver._make_batch(dm, cfg, fr["keypoints"][sel], jw, fr["H"], fr["W"], fr["focal"], fr["reg_pose"][sel], fr["reg_global"][sel],
                              cam_t[sel], np.array([[400.0, 300.0]], np.float32), "dense", True)
fb.trace(60000, evaluations=True)
fb.fit(first_stage=-1, last_stage=fb.n_stages - 1)
st = fb.stats()
print("stage losses", st["stage_loss"][0], "evals", st["stage_evals"][0], "flags", fb.penetration_flags())
rec = fb.get_trace()[0]
np.set_printoptions(linewidth=200, precision=6, suppress=False)
print("records", len(rec), "by type", {int(t): int((rec[:, 0] == t).sum()) for t in np.unique(rec[:, 0])})
bad = np.flatnonzero(~np.isfinite(rec).all(1))
print("first non-finite records at", bad[:5])
if len(bad):
    lo = max(0, bad[0] - 30)
    print(rec[lo:bad[0] + 8])
# stage results
print("stage records (type 2):"); print(rec[rec[:, 0] == 2])
P = fb.get_params()
print("final params finite:", {k: bool(np.isfinite(v).all()) for k, v in P.items()})
