"""Measurement tool (GPU box): what the interpenetration term's work looks like on the real SMPL-X surface
(synthetic.make_topology_model) under cfg_files/fit_smplx_combined_halpe.yaml verbatim -- the bench's `--workload pen`
job: device work counters of a whole fit, non-finite frames (and whether they are non-finite without the term), and
per-mesh counts of the stand-alone operator on the FITTED bodies."""
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
fr = TF.frames(dm, jm, B)
jw = TF.joint_weights(cfg, len(jm))
rngc = np.random.RandomState(1000)
cam_t = (fr["cam_t"] + 0.05 * rngc.normal(size=fr["cam_t"].shape)).astype(np.float32)
cam_c = np.tile(np.array([fr["W"] * 0.5, fr["H"] * 0.5], np.float32), (B, 1))


def fit(c, want_vertices=False):
    engine.pen_work_reset()
    t0 = time.time()
    r = driver.fit_frames(dm, c, fr["keypoints"], jw, fr["H"], fr["W"], fr["focal"], reg_pose=fr["reg_pose"],
                          reg_global=fr["reg_global"], cam_prior_t=cam_t, cam_prior_center=cam_c, lbs_mode="dense",
                          reuse_entry_eval=True, want_vertices=want_vertices)
    return r, engine.pen_work_get(), time.time() - t0

r, w, dt = fit(cfg, True)
r, w, dt = fit(cfg, True)
print("fit with the term: %.3f s, %.1f frames/s; work %s" % (dt, B / dt, w))
print("k_pen_narrow phases per column evaluation:", engine.pen_phase_ticks())
ev = r["stage_evals"].sum(1)
print("evaluations per frame: mean %.1f max %d; final loss median %.1f" % (ev.mean(), ev.max(), np.median(r["final_loss"])))
bad = np.flatnonzero(~np.isfinite(r["stage_loss"]).all(1))
print("non-finite frames:", bad.tolist(), "stage losses:", r["stage_loss"][bad].tolist(), "evals", r["stage_evals"][bad].tolist())
cfg0 = dict(cfg); cfg0["interpenetration"] = False
dm0, _ = TF.device_model(model, cfg0)
saved = dm
dm = dm0
r0, _, dt0 = fit(cfg0)
r0, _, dt0 = fit(cfg0)
dm = saved
print("fit without the term: %.3f s, %.1f frames/s; non-finite %s" % (dt0, B / dt0, np.flatnonzero(~np.isfinite(r0["stage_loss"]).all(1)).tolist()))
# the stand-alone operator on the fitted bodies
v = np.asarray(r["vertices"], np.float32)
ok = np.isfinite(v).all((1, 2))
v = v[ok]
pen = engine.Penetration(v.shape[1], np.asarray(model["f"]).astype(np.int64), parts["segm"], parts["parents"], cfg["ign_part_pairs"],
                         max_collisions=128, max_batch=len(v))
vt = torch.tensor(v, device="cuda")
engine.pen_work_reset()
pen.eval(vt, 1e-4)
w1 = engine.pen_work_get()
st = pen.stats(len(v))
print("operator on %d fitted bodies: %s" % (len(v), w1))
print("ordered pairs per body: p10 %d p50 %d p90 %d max %d; dropped max %d" % (*np.percentile(st["pairs"], [10, 50, 90]).astype(int), st["pairs"].max(), st["dropped"].max()))
for _ in range(3): pen.eval(vt, 1e-4)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): pen.eval(vt, 1e-4)
torch.cuda.synchronize()
print("operator: %.1f us per evaluation of %d bodies" % ((time.time() - t0) / 10 * 1e6, len(v)))
pen.phase_clocks(len(v))
