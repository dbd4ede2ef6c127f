import os, sys, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, _frames as FR
from smplifyx_amd import synthetic, _capi
m = synthetic.make_synthetic_model(0)
cfg = FR.load_cfg("fit_smplx_smplifyx.yaml", use_camera_prior=False)
dm, jm = FR.device_model(m, cfg, vposer=synthetic.make_synthetic_vposer(0)); B=256
fr = FR.frames(dm, jm, 3)
idx = [i%3 for i in range(B)]
fb = FR.batch(dm, cfg, fr, idx, lbs_mode="rows")
out = (C.c_int64*32)()
for st in (-1, 1):
    _capi.check(_capi.load().sfx_debug_phase_clocks(fb._h, st, out))
    t = np.array(list(out)[:17], np.float64); d = np.diff(t)
    print('vposer full stage', st, 'total', t[16]-t[0], 'wall us', (out[18]-out[17])*0.01, d.astype(int).tolist())
