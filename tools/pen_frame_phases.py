"""Measurement tool (GPU box): where the per-frame kernel of the interpenetration term (k_pen_frame) spends its time -- wall-clock
stamps at the ends of its phases (sfx_pen_phase_clocks), on bodies posed like the benchmark's frames (the real SMPL-X surface,
halpe cfg), for a few batch sizes; and the operator's time per evaluation in the three forms (sfx_debug_pen_form)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import _frames as TF
from smplifyx_amd import engine, synthetic

cfg = TF.load_cfg("fit_smplx_combined_halpe.yaml", interpenetration=True)
model = synthetic.make_topology_model(0)
parts = synthetic.topology_parts()
dm, jm = TF.device_model(model, cfg)
N = 256
tr = [synthetic.make_frame_truth(i) for i in range(N)]
dev = torch.device("cuda")
t = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)
z = lambda k: torch.zeros([N, k], device=dev)
rng = np.random.RandomState(3)
verts, _, _ = dm.lbs_forward(t([x["global_orient"] for x in tr]), t([x["body_pose"] for x in tr]), t([x["betas"] for x in tr]), z(dm.num_expr),
                             z(3), z(3), z(3), t(0.5 * rng.normal(size=(N, dm.num_pca))), t(0.5 * rng.normal(size=(N, dm.num_pca))))
verts = verts.contiguous()
faces = np.asarray(model["f"]).astype(np.int64)
names = ["A cull", "B grid", "C pairs", "D list", "E eval", "F sums", "G verts"]
for B in (1, 16, 64, 256):
    for form in (1, 3, 0):
        engine.pen_form(form)
        pen = engine.Penetration(verts.shape[1], faces, parts["segm"], parts["parents"], cfg["ign_part_pairs"], max_collisions=128, max_batch=B)
        vb = verts[:B]
        for _ in range(3): pen.eval(vb, 1e-4)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): pen.eval(vb, 1e-4)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 20
        st = pen.stats(B)
        line = "B=%3d form %d: %.1f us per evaluation; ordered pairs per body p50 %d max %d" % (B, form, dt * 1e6, np.median(st["pairs"]), st["pairs"].max())
        if form == 3:
            pca = pen.phase_clocks(B)
            line += "\n      inside C: first tile in LDS at %.1f us, wavefront 0 through its blocks at %.1f us" % (pca[:, 7].mean(), pca[:, 8].mean())
            pc = pca[:, :7]
            d = np.diff(np.concatenate([np.zeros((B, 1)), pc], 1), axis=1)
            line += "\n      phases (us, mean over the bodies; end of G mean %.1f max %.1f): " % (pc[:, 6].mean(), pc[:, 6].max()) + \
                    ", ".join("%s %.1f" % (n, v) for n, v in zip(names, d.mean(0))) + "; grid entries mean %d" % pen.phase_clocks(B)[:, 10].mean()
        print(line)
        pen.close()
engine.pen_form(1)
