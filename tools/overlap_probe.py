"""Experiment: what would a dense round cost if the GEMM of an evaluation ran next to its loss / adjoint pass?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, _frames as FR
from smplifyx_amd import synthetic, _capi
import bench as BB


def main():
    m = synthetic.make_synthetic_model(0)
    cfg = BB.build_cfg("body")
    dm, jm = FR.device_model(m, cfg)
    for B in (32, 128, 256):
        fr = FR.frames(dm, jm, B)
        for mode in (0, 1, 0, 1):
            fb = FR.batch(dm, cfg, fr, range(B), lbs_mode="dense", reuse=True)
            fb.fit(first_stage=-1, last_stage=-1)
            ms = C.c_double()
            _capi.check(_capi.load().sfx_debug_overlap_test(fb._h, 300, mode, C.byref(ms)))
            print("B %d mode %d: %.1f us per round" % (B, mode, 1e3 * ms.value / 300))


if __name__ == "__main__":
    main()
