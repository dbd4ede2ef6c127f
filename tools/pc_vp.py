import sys, ctypes as C; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, helpers as H, test_gpu_parity as T
from smplifyx_amd import synthetic, _capi
m = synthetic.make_synthetic_model(0)
cfg = H.load_cfg("fit_smplx_smplifyx.yaml")
dm = T._dm(m, cfg, vposer=synthetic.make_synthetic_vposer(0)); B=256
fr = T.synth_frames(m, cfg, 3)
idx = [i%3 for i in range(B)]
fb = H.engine_batch_from_frames(dm, cfg, fr, idx, lbs_mode="rows")
fb.guess_init(cfg["body_tri_idxs"])
out = (C.c_int64*32)()
for st in (-1, 1):
    _capi.check(_capi.load().sfx_debug_phase_clocks(fb._h, st, out))
    t = np.array(list(out)[:17], np.float64); d = np.diff(t)
    print('vposer full stage', st, 'total', t[16]-t[0], 'wall us', (out[18]-out[17])*0.01, d.astype(int).tolist())
