import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np, ctypes as C
import _frames as TF
from smplifyx_amd import engine, synthetic, driver, _capi
B=256
cfg = TF.load_cfg("fit_smplx_combined_halpe.yaml", interpenetration=True)
model = synthetic.make_topology_model(0); parts = synthetic.topology_parts()
dm, jm = TF.device_model(model, cfg); dm.set_parts(parts["segm"], parts["parents"], cfg["ign_part_pairs"])
fr = TF.frames(dm, jm, B); jw = TF.joint_weights(cfg, len(jm))
rngc = np.random.RandomState(1000)
cam_t = (fr["cam_t"] + 0.05 * rngc.normal(size=fr["cam_t"].shape)).astype(np.float32)
cam_c = np.tile(np.array([fr["W"] * 0.5, fr["H"] * 0.5], np.float32), (B, 1))
for it in range(2):
    engine.pen_work_reset()
    r = driver.fit_frames(dm, cfg, fr["keypoints"], jw, fr["H"], fr["W"], fr["focal"], reg_pose=fr["reg_pose"], reg_global=fr["reg_global"], cam_prior_t=cam_t, cam_prior_center=cam_c, lbs_mode="dense", reuse_entry_eval=True)
w = (C.c_int64 * 8)(); _capi.check(_capi.load().sfx_debug_pen_phase_ticks(w))
n=max(w[0],1)
print("first loads of the blocks %.1f us, short lists %.1f us" % (w[6]/n/100, w[7]/n/100)); print("wavefronts of k_pen_rank beyond 40 us: %d; mean us: main pass %.1f (of which long lists %.1f), queue section %.1f; long lists per such wavefront %.1f, re-derived %.1f" % (w[0], w[1]/n/100, w[2]/n/100, w[3]/n/100, w[4]/n, w[5]/n))
