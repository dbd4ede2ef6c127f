"""Cycle stamps of one mid-fit k_tick_dense launch (block 0) for the bench workloads: body / full (VPoser).
usage: python tools/phase_dense.py [body|full|fullreg] [B] [stage] [launch]"""
import os, sys, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, _frames as FR
from smplifyx_amd import synthetic, _capi
which = sys.argv[1] if len(sys.argv) > 1 else 'full'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = synthetic.make_synthetic_model(0)
full = which != 'body'
vp = which == 'full'
cfg = FR.load_cfg("fit_smplx_smplifyx.yaml", use_hands=full, use_face=full, use_vposer=vp, use_camera_prior=False)
dm, jm = FR.device_model(m, cfg, **({'vposer': synthetic.make_synthetic_vposer(0)} if vp else {}))
fr = FR.frames(dm, jm, 8)
idx = [i % 8 for i in range(B)]
lib = _capi.load()
o64 = (C.c_int64 * 64)()
fb = FR.batch(dm, cfg, fr, idx, lbs_mode="dense")
stage = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # body stage whose launch number `launch` is stamped
launch = int(sys.argv[4]) if len(sys.argv) > 4 else (400 if stage == 0 else 100)
if stage > 0:
    fb.fit(first_stage=-1, last_stage=stage - 1)
_capi.check(lib.sfx_debug_clocks(fb._h, launch, None))
fb.fit(first_stage=-1 if stage == 0 else stage, last_stage=stage)
_capi.check(lib.sfx_debug_clocks(fb._h, 0, o64))
o = np.array(list(o64), np.float64)
rel = lambda a: [int(x - o[24]) if 0 <= x - o[24] < 1e7 else None for x in a]
print(which, 'B', B, 'stage', stage, 'dense tick kernel (launch %d, block 0): total cycles' % launch, int(o[26] - o[24]), 'loss+adjoint+tick', int(o[25] - o[24]), 'next pose/export', int(o[26] - o[25]))
print('  marks 0..16 of the loss/adjoint pass rel. to entry:', rel(o[40:57]))
print('  loss sub marks 20..23:', rel(o[20:24]), ' chain 27,28:', rel(o[27:29]))
print('  next-pose marks 0..5:', rel(o[0:6]), 'export 17..19', rel(o[17:20]))
print('  mean per workgroup: loss+adjoint %.1f us, tick %.1f us, rest %.1f us; wg duration max %.1f mean %.1f us over %d' % (
    o[29] * 0.01 / max(o[60], 1), o[30] * 0.01 / max(o[60], 1), (o[59] - o[29] - o[30]) * 0.01 / max(o[60], 1), o[58] * 0.01, o[59] * 0.01 / max(o[60], 1), o[60]))
nt = max(o[63], 1)
print('  optimiser tick of frame 0, mean cycles between marks over %d ticks: fetch %.0f  consume %.0f  line-search end + iteration head + pair push %.0f  two-loop %.0f  rest-of-machine %.0f  write-back %.0f' % (
    nt, o[33] / nt, o[34] / nt, o[35] / nt, o[37] / nt, o[38] / nt, o[39] / nt))
