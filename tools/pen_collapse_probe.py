"""Measurement tool (GPU box): WHICH frames of the `--workload pen` job (cfg_files/fit_smplx_combined_halpe.yaml verbatim on
synthetic.make_topology_model) collapse or end non-finite in the stages that carry the interpenetration term, and what the same
frames do without the term.  Per frame: stage losses, closure evaluations, the order-dependence flag (a cut bucket walk), the
pair count of the stand-alone operator on the FITTED body.  Writes gpurun_out/pen_collapse_probe.npz; the frames it names go
into tools/make_goldens.py e2e_pen_set (reference-driven fits of the same frames).

usage: pen_collapse_probe.py [n_frames] [keypoints.npz]      (keypoints.npz: fit THESE frames -- keys keypoints, reg_pose,
reg_global, cam_t -- instead of frames made with the device forward)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import _frames as TF
from smplifyx_amd import engine, synthetic, driver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = TF.load_cfg("fit_smplx_combined_halpe.yaml", interpenetration=True)
model = synthetic.make_topology_model(0)
parts = synthetic.topology_parts()
dm, jm = TF.device_model(model, cfg)
dm.set_parts(parts["segm"], parts["parents"], cfg["ign_part_pairs"])
if len(sys.argv) > 2:
    g = np.load(sys.argv[2])
    fr = dict(keypoints=g["keypoints"][:B], reg_pose=g["reg_pose"][:B], reg_global=g["reg_global"][:B], cam_t=g["cam_t"][:B],
              H=600, W=800, focal=5000.0)
    B = fr["keypoints"].shape[0]
else:
    fr = TF.frames(dm, jm, B)
jw = TF.joint_weights(cfg, len(jm))
rngc = np.random.RandomState(1000)
cam_t = (fr["cam_t"] + 0.05 * rngc.normal(size=fr["cam_t"].shape)).astype(np.float32)
cam_c = np.tile(np.array([fr["W"] * 0.5, fr["H"] * 0.5], np.float32), (B, 1))


def fit(d, c, want_vertices=False):
    import warnings
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        engine.pen_work_reset()
        r = driver.fit_frames(d, c, fr["keypoints"], jw, fr["H"], fr["W"], fr["focal"], reg_pose=fr["reg_pose"],
                              reg_global=fr["reg_global"], cam_prior_t=cam_t, cam_prior_center=cam_c, lbs_mode="dense",
                              reuse_entry_eval=True, want_vertices=want_vertices)
    for w in wl:
        print("warning:", str(w.message)[:400])
    return r, engine.pen_work_get()


r, w = fit(dm, cfg, True)
print("work", w)
cfg0 = dict(cfg); cfg0["interpenetration"] = False
dm0, _ = TF.device_model(model, cfg0)
r0, _ = fit(dm0, cfg0, False)
v = np.asarray(r["vertices"], np.float32)
ok = np.isfinite(v).all((1, 2))
pairs = np.full(B, -1, np.int64); dropped = np.full(B, -1, np.int64)
if ok.any():
    vv = v[ok]
    pen = engine.Penetration(vv.shape[1], np.asarray(model["f"]).astype(np.int64), parts["segm"], parts["parents"], cfg["ign_part_pairs"],
                             max_collisions=128, max_batch=len(vv))
    pen.eval(torch.tensor(vv, device="cuda"), 1e-4)
    st = pen.stats(len(vv))
    pairs[ok] = st["pairs"]; dropped[ok] = st["dropped"]
sl, sl0 = r["stage_loss"], r0["stage_loss"]
ev, ev0 = r["stage_evals"], r0["stage_evals"]
flag = r["pen_order_dependent"]
np.set_printoptions(linewidth=200, precision=5)
nonfin = np.flatnonzero(~np.isfinite(sl).all(1))
print("non-finite with the term:", nonfin.tolist(), "| without:", np.flatnonzero(~np.isfinite(sl0).all(1)).tolist())
print("order-dependent (cut walk) frames:", np.flatnonzero(flag).tolist())
ratio = sl[:, -1] / sl0[:, -1]
print("final loss with / without the term: median %.4f p90 %.4f p99 %.4f max %.4f" % tuple(np.nanpercentile(ratio, [50, 90, 99, 100])))
sus = np.argsort(-np.nan_to_num(ratio, nan=1e30))[:16]
print("largest ratios (frame, with, without, evals with, evals without, fitted-body ordered pairs, flag):")
for i in sus:
    print("  %4d  %12.4f %12.4f  %s %s  pairs %d dropped %d flag %d n_orient %d" % (i, sl[i, -1], sl0[i, -1], ev[i].tolist(), ev0[i].tolist(), pairs[i], dropped[i], flag[i], r["n_orient"][i]))
print("fitted-body ordered pairs: p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(pairs[pairs >= 0], [50, 90, 99, 100]).astype(int)))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/pen_collapse_probe.npz", stage_loss=sl, stage_loss_noterm=sl0, stage_evals=ev, stage_evals_noterm=ev0,
                    flag=flag, pairs=pairs, dropped=dropped, n_orient=r["n_orient"], keypoints=fr["keypoints"], reg_pose=fr["reg_pose"],
                    reg_global=fr["reg_global"], cam_t=fr["cam_t"], cam_prior_t=cam_t)
print("wrote gpurun_out/pen_collapse_probe.npz")
