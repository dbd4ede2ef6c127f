import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, time
from smplifyx_amd import engine, synthetic
m = synthetic.make_synthetic_model(0, surface=("soup" not in sys.argv))
parts = synthetic.make_synthetic_parts(m)
ign = ["9,16", "9,17", "6,16", "6,17", "1,2", "12,22"]
v = np.asarray(m["v_template"], np.float32); f = np.asarray(m["f"]).astype(np.int64)
for B in (1, 64, 256):
    pen = engine.Penetration(len(v), f, parts["segm"], parts["parents"], ign, max_collisions=128, max_batch=B)
    vb = torch.tensor(np.stack([v]*B), device="cuda")
    for _ in range(3): pen.eval(vb, 1e-4)
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(10): pen.eval(vb, 1e-4)
    torch.cuda.synchronize(); dt=(time.time()-t0)/10
    st = pen.stats(B)
    print("B=%d  %.1f us per eval, pairs/frame %d, walks cut %d" % (B, dt*1e6, st["pairs"][0], st["walks_cut"][0]))
    pen.close()
