"""Per-frame relative stage-loss differences of the 16 VPoser fits against the reference's fp32 fits (the statistic of
tests/test_gpu_parity.py::test_vposer_set_matches_reference), frame by frame.  usage: vposer_set_probe.py [rows|dense]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import test_gpu_parity as T
from smplifyx_amd import synthetic
mode = sys.argv[1] if len(sys.argv) > 1 else "rows"
model = synthetic.make_synthetic_model(0)
g = T._golden("e2e_vposer_set")
cfg = H.load_cfg("fit_smplx_smplifyx.yaml")
dm = T._dm(model, cfg, vposer=synthetic.make_synthetic_vposer(0))
n = g["keypoints"].shape[0]
frames = dict(keypoints=g["keypoints"], H=600, W=800, focal=5000.0)
fb = H.engine_batch_from_frames(dm, cfg, frames, range(n), lbs_mode=mode, reuse=True)
fb.guess_init(cfg["body_tri_idxs"]); fb.fit()
st = fb.stats()
ours = st["stage_loss"].astype(np.float64)
r32 = np.stack([g["f%d_f32_losses" % i] for i in range(n)]); r64 = np.stack([g["f%d_f64_losses" % i] for i in range(n)])
d = (ours - r32) / np.abs(r32); y = (r64 - r32) / np.abs(r32)
np.set_printoptions(precision=4, suppress=True, linewidth=250)
for k in range(d.shape[1]):
    print(mode, "stage", k, "ours-ref32:", d[:, k], "| mean %.4f  mean|.| %.4f  median %.4f" % (d[:, k].mean(), np.abs(d[:, k]).mean(), np.median(d[:, k])))
    print(mode, "stage", k, "ref64-ref32:", y[:, k], "| mean %.4f  mean|.| %.4f" % (y[:, k].mean(), np.abs(y[:, k]).mean()))
print("evals", st["stage_evals"].sum(1))
