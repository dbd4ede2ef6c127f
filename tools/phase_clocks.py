import os, sys, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, _frames as FR
from smplifyx_amd import synthetic, _capi
m = synthetic.make_synthetic_model(0)
for which in ('body','full'):
    cfg = FR.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=(which=='full'), use_face=(which=='full'), use_camera_prior=False)
    dm, jm = FR.device_model(m, cfg); B=256
    fr = FR.frames(dm, jm, 3)
    idx = [i%3 for i in range(B)]
    fb = FR.batch(dm, cfg, fr, idx, lbs_mode="rows")
    out = (C.c_int64*32)()
    for st in (-1, 1):
        _capi.check(_capi.load().sfx_debug_phase_clocks(fb._h, st, out))
        t = np.array(list(out)[:17], np.float64); d = np.diff(t)
        wall = (out[18] - out[17]) * 0.01   # 100 MHz constant clock -> us
        print('   loss sub-phases (8->20->21->22->23->9):', [int(out[20]-out[8]), int(out[21]-out[20]), int(out[22]-out[21]), int(out[23]-out[22]), int(out[9]-out[23])])
        print('   chain sub-phases (3->27->28->4):', [int(out[27]-out[3]), int(out[28]-out[27]), int(out[4]-out[28])])
        print(which, 'stage', st, 'total cycles', t[16]-t[0], 'wall us', wall, 'phases', d.astype(int).tolist())

    # optimiser tick profile of frame 0 over a whole fit (rows path)
    fb = FR.batch(dm, cfg, fr, idx, lbs_mode="rows")
    lib = _capi.load()
    _capi.check(lib.sfx_debug_clocks(fb._h, 1, None))
    fb.fit()
    o64 = (C.c_int64*64)()
    _capi.check(lib.sfx_debug_clocks(fb._h, 0, o64))
    o = np.array(list(o64), np.float64); n = max(o[63], 1)
    print(which, 'ticks', int(o[63]), 'avg cycles/tick between marks [load, consume, act->hist, loop1, loop2, rest, store]:',
          (o[33:40] / n).astype(int).tolist(), 'sum', int(o[33:40].sum() / n))

    # dense path: one k_tick_dense launch = [loss+adjoint of eval i] -> [tick] -> [pose/FK/export of eval i+1]
    fb = FR.batch(dm, cfg, fr, idx, lbs_mode="dense")
    _capi.check(lib.sfx_debug_clocks(fb._h, 400, None))        # stamps of launch 400: first body stage, history full
    fb.fit(first_stage=-1, last_stage=0)
    _capi.check(lib.sfx_debug_clocks(fb._h, 0, o64))
    o = np.array(list(o64), np.float64)
    post = np.diff(o[40:57]); pre = np.diff(o[0:9])
    print(which, 'dense tick kernel (last launch): total', int(o[26]-o[24]), 'post-closure+tick', int(o[25]-o[24]), 'pre/export', int(o[26]-o[25]))
    print('    post phases', post.astype(int).tolist()); print('    pre phases', pre.astype(int).tolist())
    rel = lambda a: [int(x - o[24]) if 0 <= x - o[24] < 1e6 else None for x in a]
    print('    marks of the loss/adjoint pass relative to kernel entry (cycles; None = not passed in this launch):', rel(o[40:57]))
    print('    marks of the next-pose pass:', rel(o[0:5]), 'export (feat row, AT, state)', rel(o[17:20]), 'after tick', rel([o[25]]), 'exit', rel([o[26]]))
    print('    mean per workgroup: loss + adjoint %.1f us, optimiser tick %.1f us, rest (next pose / chain / export, entry, exit) %.1f us' % (
        o[29] * 0.01 / max(o[60], 1), o[30] * 0.01 / max(o[60], 1), (o[59] - o[29] - o[30]) * 0.01 / max(o[60], 1)))
    print('    per-workgroup duration of k_tick_dense over the fit: max %.1f us, mean %.1f us over %d workgroup launches' % (o[58] * 0.01, o[59] * 0.01 / max(o[60], 1), o[60]))
