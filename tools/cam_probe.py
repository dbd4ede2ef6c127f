"""Diagnostic: camera stage of one frame of the VPoser reference set, device trace next to the oracle machine."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H, test_gpu_parity as T
from smplifyx_amd import synthetic
i = int(sys.argv[1]) if len(sys.argv) > 1 else 14
g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_vposer_set.npz"))
cfg = H.load_cfg("fit_smplx_smplifyx.yaml")
model = synthetic.make_synthetic_model(0)
dm = T._dm(model, cfg, vposer=synthetic.make_synthetic_vposer(0))
frames = dict(keypoints=g["keypoints"], H=600, W=800, focal=5000.0)
np.set_printoptions(precision=7, linewidth=200, suppress=False)
for reuse in (True, False):
    fb = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=reuse)
    fb.guess_init(cfg["body_tri_idxs"])
    P0 = fb.get_params()
    print("init cam_t", P0["cam_translation"][0], "go", P0["global_orient"][0])
    fb.trace(4096, evaluations=True)
    fb.fit(first_stage=-1, last_stage=-1)
    rec = fb.get_trace()[0]
    print("reuse", reuse, "result", fb.stats()["stage_loss"][0, 0], "evals", fb.stats()["stage_evals"][0, 0], "ref", g["f%d_f32_losses" % i][0])
    print(rec[:60])
    print("final", fb.get_params()["cam_translation"][0], fb.get_params()["global_orient"][0])

# gradient check at the point where the device's camera stage got stuck
P = fb.get_params()
loss, grad = fb.closure(-1)
Q = {k: v for k, v in P.items() if k != "body_pose"}
Q["est_tz"] = np.array([P0["cam_translation"][0, 2]], np.float32)
fr1 = {k: (v[i:i + 1] if isinstance(v, np.ndarray) else v) for k, v in frames.items()}
lo, go_ = T._oracle_closure(model, cfg, fr1, 0, Q, -1, dtype=torch.float64)
print("HIP closure at its final point: loss", loss[0], "grad", grad[0])
print("oracle fp64 at the same point : loss", lo, "grad", go_)
# finite differences of the HIP loss along each variable
for k, (name, j) in enumerate([("cam_translation", 0), ("cam_translation", 1), ("cam_translation", 2), ("global_orient", 0), ("global_orient", 1), ("global_orient", 2)]):
    h = 1e-3
    vals = []
    for sgn in (+1, -1):
        Pp = {kk: vv.copy() for kk, vv in P.items() if kk != "body_pose"}
        Pp[name][0, j] += sgn * h
        fb.set_params(**Pp)
        vals.append(float(fb.closure(-1)[0][0]))
    print("  d/d %s[%d]: finite difference %.4f   HIP gradient %.4f   oracle %.4f" % (name, j, (vals[0] - vals[1]) / (2 * h), grad[0][k], go_[k]))

# the specification machine (oracle/lbfgs_machine.py) on the SAME HIP closure from the same start
import test_gpu_optimizer_steps as S
for reuse in (True,):
    fc = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=reuse)
    fc.guess_init(cfg["body_tri_idxs"])
    m = S._machine(np.concatenate([P0["cam_translation"][0], P0["global_orient"][0]]), cfg, [(0, 3, True), (3, 3, True)], reuse)
    hist = []
    while not m.done:
        x = m.x_trial
        fc.set_params(cam_translation=x[None, :3], global_orient=x[None, 3:], pose_embedding=P0["pose_embedding"])
        f, gr = fc.closure(-1)
        hist.append((float(f[0]), float(np.abs(gr[0]).max())))
        m.feed(f[0], gr[0])
    mac = np.array(m.records)
    print("machine on the HIP closure: result", mac[-1], "evaluations", len(hist))
    print(mac[:60])
