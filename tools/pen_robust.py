import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import test_gpu_penetration as TP
from smplifyx_amd import engine
verts, faces, segm, parents = TP._two_spheres(0.13)
pen = engine.Penetration(len(verts), faces, segm, parents, max_collisions=64, max_batch=4)
for name, mod in (("nan", lambda v: (v.__setitem__((0, 5, 1), np.nan), v)[1]),
                  ("inf", lambda v: (v.__setitem__((1, 7, 0), np.inf), v)[1]),
                  ("huge", lambda v: (v.__setitem__((2, 9, 2), 1e30), v)[1]),
                  ("allnan", lambda v: (v.__setitem__((3, slice(None), slice(None)), np.nan), v)[1])):
    vb = np.stack([verts] * 4).astype(np.float32)
    vb = mod(vb)
    loss, dv = pen.eval(torch.tensor(vb, device="cuda"), 0.5)
    torch.cuda.synchronize()
    print(name, loss.cpu().numpy(), pen.stats(4)["pairs"], flush=True)
