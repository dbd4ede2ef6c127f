"""Diagnostic: cost / pair statistics of the penetration operator on POSED synthetic meshes (the
truth poses of the synthetic frames), surface-like mesh."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import _frames as FR
from smplifyx_amd import engine, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = FR.load_cfg("fit_smplx_combined_halpe.yaml", interpenetration=True, use_hands=False, use_face=False)
m = synthetic.make_synthetic_model(0, surface=("soup" not in sys.argv))
parts = synthetic.make_synthetic_parts(m)
dm, jm = FR.device_model(m, cfg)
fr = FR.frames(dm, jm, B)
P = fr["truth"]
dev = torch.device("cuda")
z = lambda k: torch.zeros([B, k], device=dev)
t = lambda a: torch.tensor(a, device=dev)
for scale in (0.0, 0.5, 1.0):
    v, j, _ = dm.lbs_forward(t(P["global_orient"]), t(P["body_pose"] * scale), t(P["betas"]), z(dm.num_expr), z(3), z(3), z(3),
                             z(dm.num_pca), z(dm.num_pca), return_verts=True, return_full_pose=False)
    f = np.asarray(m["f"]).astype(np.int64)
    pen = engine.Penetration(dm.V, f, parts["segm"], parts["parents"], cfg["ign_part_pairs"],
                             max_collisions=int(cfg["max_collisions"]), max_batch=B)
    v = v.contiguous()
    for _ in range(2): pen.eval(v, float(cfg["df_cone_height"]))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): loss, g = pen.eval(v, float(cfg["df_cone_height"]))
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    st = pen.stats(B)
    print("pose scale %.1f: %.0f us/eval; pairs/frame mean %.0f max %d; %s; loss mean %.3g" % (
        scale, dt * 1e6, st["pairs"].mean(), st["pairs"].max(), {k: (int(v_.max()) if hasattr(v_, "max") else v_) for k, v_ in st.items()}, float(loss.mean())))
    print("   broad-phase steps end at (us, frame 0 / max):", pen.phase_clocks(B)[0], pen.phase_clocks(B).max(0))
    pen.close()
