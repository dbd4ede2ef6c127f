"""Where a step of the halpe workload spends its time outside the fitting loop: batch creation (device allocations of the collision
buffers), the fit, collection of the results, destruction.  usage: pen_step_phases.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from smplifyx_amd import engine, synthetic, utils as U, driver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = bench.build_cfg("pen")
model = synthetic.make_synthetic_model(0, surface=True)
jm = U.smpl_to_annotation("smplx", use_hands=cfg["use_hands"], use_face=cfg["use_face"], use_face_contour=cfg["use_face_contour"], format=cfg["format"])
dm = engine.DeviceModel(model, joint_map=jm, num_betas=cfg["num_betas"], num_expression_coeffs=cfg["num_expression_coeffs"],
                        num_pca_comps=cfg["num_pca_comps"], use_face_contour=cfg["use_face_contour"])
parts = synthetic.make_synthetic_parts(model); dm.set_parts(parts["segm"], parts["parents"], cfg["ign_part_pairs"])
dev = torch.device("cuda")
def joints_fn(P):
    z = lambda n: torch.zeros([B, n], device=dev); t = lambda a: torch.tensor(a, device=dev)
    _, j, _ = dm.lbs_forward(t(P["global_orient"]), t(P["body_pose"]), t(P["betas"]), z(10), z(3), z(3), z(3), z(12), z(12), return_verts=False, return_full_pose=False)
    return j.cpu().numpy()
fr = synthetic.make_frames(B, joints_fn, len(jm), focal=float(cfg.get("focal_length") or 5000.0))
jw = np.ones(len(jm), np.float32); jw[cfg["joints_to_ign"]] = 0
rng = np.random.RandomState(1000); ct = (fr["cam_t"] + 0.05 * rng.normal(size=fr["cam_t"].shape)).astype(np.float32)
cc = np.tile(np.array([fr["W"] * 0.5, fr["H"] * 0.5], np.float32), (B, 1))
sync = torch.cuda.synchronize
for rep in range(3):
    sync(); t0 = time.perf_counter()
    fb, prep = driver._make_batch(dm, cfg, fr["keypoints"], jw, fr["H"], fr["W"], fr["focal"], fr["reg_pose"], fr["reg_global"], ct, cc, "dense", True)
    sync(); t1 = time.perf_counter()
    fb.fit(first_stage=-1, last_stage=fb.n_stages - 1)
    sync(); t2 = time.perf_counter()
    res = driver._collect(fb, prep, False)
    sync(); t3 = time.perf_counter()
    fb.close()
    sync(); t4 = time.perf_counter()
    print("rep %d: create %.1f ms, fit %.1f ms (host stats %s), collect %.1f ms, destroy %.1f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, engine.loop_host_stats() if hasattr(engine, "loop_host_stats") else "", (t3 - t2) * 1e3, (t4 - t3) * 1e3))
