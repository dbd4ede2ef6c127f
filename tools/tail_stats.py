import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tools')
import bench as BB, _frames as FR
from smplifyx_amd import synthetic, driver
cfg = BB.build_cfg("body"); m = synthetic.make_synthetic_model(0)
dm, jm = FR.device_model(m, cfg)
fr = FR.frames(dm, jm, 1024)
jw = FR.joint_weights(cfg, len(jm))
kp = fr["keypoints"]
ninit = np.array([sum(kp[b, j, 2] > 0 for j in cfg["init_joints_idxs"]) for b in range(len(kp))])
for mode in ("rows","dense"):
    res = driver.fit_frames(dm, cfg, kp, jw, fr["H"], fr["W"], fr["focal"], reg_pose=fr["reg_pose"], reg_global=fr["reg_global"], lbs_mode=mode)
    fl = res["stage_loss"][:,-1]; bad = np.flatnonzero(~np.isfinite(fl))
    med = np.nanmedian(fl)
    print(mode, "non-finite", bad, "valid init joints of those", ninit[bad], "| median", med, "mean(finite)", np.nanmean(fl), "> 3x median:", int((fl > 3*med).sum()), "> 10x:", int((fl > 10*med).sum()), "evals mean", res["stage_evals"].sum(1).mean())
    print("   outliers (>10x) init joints", ninit[np.flatnonzero(fl > 10*med)], "frames with < 4 valid init joints:", int((ninit < 4).sum()), "their median loss", np.nanmedian(fl[ninit < 4]), "others", np.nanmedian(fl[ninit == 4]))
