"""Micro-benchmark of k_lbs_dense alone (HIP events inside libsfx): time vs frames per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplifyx_amd import engine, synthetic
m = synthetic.make_synthetic_model(0)
dm = engine.DeviceModel(m)
dev = torch.device("cuda")
BS = [int(x) for x in sys.argv[1:]] or [1, 8, 16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224, 240, 256, 384, 512, 1024]
for B in BS:
    t = lambda n, s=0.3: (s * torch.randn([B, n], device=dev)).contiguous()
    args = [t(3), t(63), t(10, 1.0), t(10, 1.0), t(3), t(3), t(3), t(12, 1.0), t(12, 1.0)]
    for _ in range(3):
        dm.lbs_forward(*args)
    engine.prof_enable(True); engine.prof_reset()
    for _ in range(20):
        dm.lbs_forward(*args)
    engine.prof_enable(False)
    ms, n, u = engine.prof_get("lbs_dense")
    us = 1e3 * ms / n
    fl = B * (2.0 * 506 * 3 * 10475 + 2.0 * 55 * 12 * 10475 + 21 * 10475)
    print("B=%4d  %7.1f us  %6.1f TFLOP/s  (%.1f%% of 157.3)" % (B, us, fl / us / 1e6, 100 * fl / us / 1e6 / 157.3))
