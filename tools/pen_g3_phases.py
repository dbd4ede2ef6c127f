"""Diagnostic (PEN_COUNT build: tools/build_variant.sh count collide -DPEN_COUNT; SFX_LIB=.../libsfx_count.so): per-phase cycles of
k_pen_g3 and the walk's candidate counters for B posed frames (random poses of the synthetic surface model)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplifyx_amd import engine, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = synthetic.make_synthetic_model(0, surface=True)
parts = synthetic.make_synthetic_parts(m)
ign = ["9,16", "9,17", "6,16", "6,17", "1,2", "12,22"]
v = np.asarray(m["v_template"], np.float32); f = np.asarray(m["f"]).astype(np.int64)
pen = engine.Penetration(len(v), f, parts["segm"], parts["parents"], ign, max_collisions=128, max_batch=B)
rng = np.random.default_rng(0)
vb = torch.tensor(np.stack([v + rng.normal(0, 1e-3, v.shape).astype(np.float32) for _ in range(B)]), device="cuda")
for _ in range(3): pen.eval(vb, 1e-4)
torch.cuda.synchronize()
pen.phase_clocks(B)      # (the PEN_COUNT build prints its counters to stderr here)
print("entries per frame:", pen.phase_clocks(B)[:4, 10])
