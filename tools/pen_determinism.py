"""Diagnostic: is a fit with the interpenetration term run-to-run deterministic?  Fits the same frames twice (resident) and
once through a column pool, lists the frames whose results differ and their collision diagnostics at the final parameters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, _frames as FR
from smplifyx_amd import synthetic, driver


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
    maxiters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    model = synthetic.make_synthetic_model(0, surface=True)
    cfg = FR.load_cfg("fit_smplx_combined_halpe.yaml", use_hands=False, use_face=False, interpenetration=True, use_camera_prior=False,
                      maxiters=maxiters)
    dm, jm = FR.device_model(model, cfg)
    parts = synthetic.make_synthetic_parts(model)
    dm.set_parts(parts["segm"], parts["parents"], cfg["ign_part_pairs"])
    fr = FR.frames(dm, jm, B)
    jw = FR.joint_weights(cfg, len(jm))
    kw = dict(reg_pose=fr["reg_pose"], reg_global=fr["reg_global"], lbs_mode="dense")
    from smplifyx_amd import engine
    runs = []
    for sl in (0, 0, 32):
        engine.pen_work_reset()
        runs.append(driver.fit_frames(dm, cfg, fr["keypoints"], jw, 600, 800, 5000.0, slots=sl, **kw))
        print("slots", sl, "work", engine.pen_work_get())
    for name, r in (("again", runs[1]), ("pool", runs[2])):
        diff = [i for i in range(B) if not all(np.array_equal(r[k][i], runs[0][k][i], equal_nan=True)
                                               for k in ("stage_loss", "pose_embedding", "betas", "cam_translation"))]
        print(name, "frames that differ:", diff)
        for i in diff[:10]:
            print("   frame", i, "stage_loss", runs[0]["stage_loss"][i], r["stage_loss"][i], "evals", runs[0]["stage_evals"][i], r["stage_evals"][i])
    # diagnostics at the final parameters of run 0
    fb, _ = driver._make_batch(dm, cfg, fr["keypoints"], jw, 600, 800, 5000.0, fr["reg_pose"], fr["reg_global"], None, None, "dense", True)
    P = {k: runs[0][k] for k in ("cam_translation", "global_orient", "betas", "left_hand_pose", "right_hand_pose", "expression", "jaw_pose",
                                 "leye_pose", "reye_pose", "pose_embedding")}
    fb.set_params(regression_pose=fr["reg_pose"], **P)
    for rep in range(2):
        loss, grad = fb.closure(fb.n_stages - 1)
        st = fb.penetration_stats()
        print("closure rep", rep, "loss sum %.6f" % float(np.nansum(loss)), "pairs max", st["pairs"].max(), "dropped max", st["dropped"].max(),
              "overflow", st["entry_overflow"].max(), "frames with dropped", np.flatnonzero(st["dropped"]).tolist()[:20])
        if rep == 0:
            l0, g0 = loss.copy(), grad.copy()
        else:
            print("closure repeat differs in frames:", [i for i in range(B) if not (np.array_equal(l0[i], loss[i], equal_nan=True) and np.array_equal(g0[i], grad[i], equal_nan=True))])


if __name__ == "__main__":
    main()
