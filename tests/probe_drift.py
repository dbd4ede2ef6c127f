#!/usr/bin/env python
"""Diagnostic (GPU box; test infrastructure, not collected by pytest): how accurate are the HIP closure's
loss and gradient near the optimum of the benchmark configuration, next to torch fp32 (the arithmetic the
reference runs), both measured against the oracle in fp64?

The last two stages of the benchmark schedule end when one whole LBFGS.step makes no progress
(fitting.py:185-189 with ftol 1e-9 = "the fp32 loss did not decrease"): a closure whose gradient is noisier
than autograd's stalls earlier.  Points probed: the REFERENCE's own final fp32 parameters of the golden
frames (tests/golden/e2e_bench.npz), evaluated with the weights of every body stage.

    python tests/probe_drift.py [--frames 8] [--out gpurun_out/probe_drift.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import helpers as H                      # noqa: E402
import test_gpu_parity as T              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--at", default="ref", choices=["ref", "hip"])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "probe_drift.json"))
    args = ap.parse_args()
    import bench as BB
    from smplifyx_amd import synthetic
    g = np.load(os.path.join(HERE, "golden", "e2e_bench.npz"))
    n = min(args.frames, g["keypoints"].shape[0])
    cfg = BB.build_cfg("body")
    model = synthetic.make_synthetic_model(0)
    dm = T._dm(model, cfg)
    frames = dict(keypoints=g["keypoints"][:n], reg_pose=g["reg_pose"][:n], reg_global=g["reg_global"][:n], H=600, W=800,
                  focal=5000.0)
    names = ("camera_translation", "global_orient", "betas", "body_pose", "expression", "jaw_pose", "leye_pose", "reye_pose",
             "left_hand_pose", "right_hand_pose")
    if args.at == "ref" and "f0_f32_expression" in g:
        P = {k: np.concatenate([g["f%d_f32_%s" % (i, k)].reshape(1, -1) for i in range(n)]).astype(np.float32) for k in names}
        P["cam_translation"] = P.pop("camera_translation"); P["pose_embedding"] = P.pop("body_pose")
    else:      # the engine's own final point
        fb = H.engine_batch_from_frames(dm, cfg, frames, range(n), lbs_mode="rows", reuse=True)
        fb.guess_init(cfg["body_tri_idxs"])
        fb.fit()
        P = fb.get_params(); P.pop("body_pose")
        print("engine final losses", fb.stats()["stage_loss"][:, -1])
    est = P["cam_translation"][:, 2].copy()
    out = {"frames": n, "stages": {}}
    fbs = {}
    for mode in ("rows", "dense"):
        fb = H.engine_batch_from_frames(dm, cfg, frames, range(n), lbs_mode=mode)
        fb.set_params(regression_pose=frames["reg_pose"], **P)
        fbs[mode] = fb
    Q = dict(P); Q["est_tz"] = est
    # forward accuracy: mapped joints of the dense path vs the oracle in fp64, next to torch fp32
    _, jh = fbs["dense"].forward(want_verts=False)
    jh = jh.cpu().numpy().astype(np.float64)
    Pf = {k: v for k, v in P.items() if k != "cam_translation"}
    _, j64, _ = T._oracle_forward(model, cfg, Pf, torch.float64)
    _, j32, _ = T._oracle_forward(model, cfg, Pf, torch.float32)
    _, jr = fbs["rows"].forward(want_verts=False)
    jr = jr.cpu().numpy().astype(np.float64)
    er = np.abs(jr - j64).max(-1)
    eh = np.abs(jh - j64).max(-1); e32 = np.abs(j32 - j64).max(-1)          # [n, K]
    np.set_printoptions(precision=1, linewidth=200)
    print("joint |err| vs fp64 [1e-8 m], per keypoint, median over frames\n  hip    ", np.median(eh, 0) * 1e8,
          "\n  rows   ", np.median(er, 0) * 1e8, "\n  torch32", np.median(e32, 0) * 1e8)
    print("  overall rms: hip %.2e  torch32 %.2e" % (np.sqrt((eh ** 2).mean()), np.sqrt((e32 ** 2).mean())))
    out["joint_err"] = dict(hip=eh.tolist(), torch32=e32.tolist())
    # where the forward error enters: skinning transforms A, all vertices
    from oracle import body_model as OB
    Ah = fbs["dense"].debug_read("A").reshape(n, 12, 55).astype(np.float64)
    vh = fbs["dense"].debug_read("verts").reshape(n, -1, 3).astype(np.float64)

    def internals(dtype):
        bm = H.oracle_model(model, cfg, dtype)
        As, Vs = [], []
        for i in range(n):
            bm.reset_params(**{k: v[i:i + 1] for k, v in Pf.items() if k != "pose_embedding"})
            with torch.no_grad():
                bp = torch.tensor(Pf["pose_embedding"][i:i + 1], dtype=dtype)
                lh = torch.einsum("bi,ij->bj", bm.left_hand_pose, bm.left_hand_components)
                rh = torch.einsum("bi,ij->bj", bm.right_hand_pose, bm.right_hand_components)
                fp = torch.cat([bm.global_orient, bp, bm.jaw_pose, bm.leye_pose, bm.reye_pose, lh, rh], 1) + bm.pose_mean
                coeff = torch.cat([bm.betas, bm.expression], 1)
                v_shaped = bm.v_template + torch.einsum("bl,mkl->bmk", coeff, bm.shapedirs)
                J = torch.einsum("bik,ji->bjk", v_shaped, bm.J_regressor)
                R = OB.batch_rodrigues(fp.view(-1, 3)).view(1, -1, 3, 3)
                _, A = OB.rigid_transform_chain(R, J, bm.parents)
                o = bm(return_verts=True, body_pose=bp)
            As.append(A[0, :, :3, :].reshape(55, 12).T.numpy().astype(np.float64)); Vs.append(o.vertices[0].numpy().astype(np.float64))
        return np.stack(As), np.stack(Vs)
    A64, V64 = internals(torch.float64)
    A32, V32 = internals(torch.float32)
    tr = [3, 7, 11]; ro = [0, 1, 2, 4, 5, 6, 8, 9, 10]
    print("A translation |err| rms: hip %.2e torch32 %.2e ; rotation: hip %.2e torch32 %.2e" % (
        np.sqrt(((Ah - A64)[:, tr] ** 2).mean()), np.sqrt(((A32 - A64)[:, tr] ** 2).mean()),
        np.sqrt(((Ah - A64)[:, ro] ** 2).mean()), np.sqrt(((A32 - A64)[:, ro] ** 2).mean())))
    print("A translation err per joint (hip, 1e-8):", (np.abs(Ah - A64)[:, tr].max(1).mean(0) * 1e8))
    print("A translation err per joint (t32, 1e-8):", (np.abs(A32 - A64)[:, tr].max(1).mean(0) * 1e8))
    print("all vertices |err| rms: hip %.2e torch32 %.2e ; max hip %.2e torch32 %.2e" % (
        np.sqrt(((vh - V64) ** 2).mean()), np.sqrt(((V32 - V64) ** 2).mean()), np.abs(vh - V64).max(), np.abs(V32 - V64).max()))
    for stage in range(fbs["rows"].n_stages):
        rec = {"f64": [], "gnorm64": [], "ginf64": []}
        hip = {m: fbs[m].closure(stage) for m in fbs}
        for key in ("rows", "dense", "torch32"):
            rec[key] = {"f_rel": [], "g_rel": [], "g_abs_inf": []}
        for i in range(n):
            f64, g64 = T._oracle_closure(model, cfg, frames, i, Q, stage, dtype=torch.float64)
            f32, g32 = T._oracle_closure(model, cfg, frames, i, Q, stage, dtype=torch.float32)
            rec["f64"].append(f64); rec["gnorm64"].append(float(np.linalg.norm(g64))); rec["ginf64"].append(float(np.abs(g64).max()))
            for key, (f, gg) in (("rows", (hip["rows"][0][i], hip["rows"][1][i])),
                                 ("dense", (hip["dense"][0][i], hip["dense"][1][i])), ("torch32", (f32, g32))):
                rec[key]["f_rel"].append(float(abs(f - f64) / abs(f64)))
                rec[key]["g_rel"].append(float(np.linalg.norm(gg - g64) / np.linalg.norm(g64)))
                rec[key]["g_abs_inf"].append(float(np.abs(gg - g64).max()))
        out["stages"][str(stage)] = rec
        print("stage %d  |g|64 median %.3g" % (stage, np.median(rec["gnorm64"])))
        for key in ("rows", "dense", "torch32"):
            print("   %-8s f_rel median %.2e max %.2e   g_rel median %.2e max %.2e   g_abs_inf median %.2e" % (
                key, np.median(rec[key]["f_rel"]), np.max(rec[key]["f_rel"]), np.median(rec[key]["g_rel"]),
                np.max(rec[key]["g_rel"]), np.median(rec[key]["g_abs_inf"])))
    # forward noise vs backward noise: gradient of the fp64 objective whose joints are shifted by a CONSTANT
    # (given joints - fp64 joints): the exact Jacobian applied to a noisy forward
    def grad_with_joint_shift(i, stage, delta):
        ff = H.oracle_frame_fit(model, cfg, frames, i, dtype=torch.float64)
        bm = ff.bm
        with torch.no_grad():
            for k, v in Q.items():
                if k == "est_tz":
                    continue
                if k == "pose_embedding":
                    ff.pose_embedding.copy_(torch.tensor(v[i:i + 1], dtype=torch.float64))
                elif k == "cam_translation":
                    ff.cam_t.copy_(torch.tensor(v[i:i + 1], dtype=torch.float64))
                else:
                    getattr(bm, k).copy_(torch.tensor(v[i:i + 1], dtype=torch.float64))
        ps = [p_ for p_ in bm.parameters() if p_.requires_grad] + [ff.pose_embedding]
        w = dict(ff.stages[stage]); w["data_weight"] = ff.data_weight
        w["bending_prior_weight"] = 3.17 * w["body_pose_weight"]
        jw = ff.jw.clone(); jw[:, ff.low] = 0
        for p_ in ps:
            p_.grad = None
        o = bm(return_verts=True, body_pose=ff._body_pose(), return_full_pose=True)
        o = o._replace(joints=o.joints + torch.tensor(delta, dtype=torch.float64)[None])
        loss = ff._body_terms_nopen(o, stage, w, jw)["total"]
        loss.backward()
        return torch.cat([(p_.grad.reshape(-1) if p_.grad is not None else torch.zeros(p_.numel(), dtype=torch.float64)) for p_ in ps]).numpy()
    stage = fbs["rows"].n_stages - 1
    gh_all = fbs["rows"].closure(stage)[1]
    gd_all = fbs["dense"].closure(stage)[1]
    out["grads"] = []
    for i in range(n):
        f64, g64 = T._oracle_closure(model, cfg, frames, i, Q, stage, dtype=torch.float64)
        f32, g32 = T._oracle_closure(model, cfg, frames, i, Q, stage, dtype=torch.float32)
        g_fr = grad_with_joint_shift(i, stage, jr[i] - j64[i])
        g_fd = grad_with_joint_shift(i, stage, jh[i] - j64[i])
        g_ft = grad_with_joint_shift(i, stage, j32[i] - j64[i])
        nrm = lambda a: float(np.linalg.norm(a))
        print("frame %d |g64| %.3g: err rows %.3f dense %.3f torch32 %.3f | forward-noise only: rows %.3f dense %.3f torch32 %.3f" % (
            i, nrm(g64), nrm(gh_all[i] - g64), nrm(gd_all[i] - g64), nrm(g32 - g64), nrm(g_fr - g64), nrm(g_fd - g64), nrm(g_ft - g64)))
        out["grads"].append(dict(g64=g64.tolist(), rows=gh_all[i].tolist(), dense=gd_all[i].tolist(), torch32=g32.tolist(),
                                 fwd_rows=g_fr.tolist(), fwd_dense=g_fd.tolist(), fwd_torch32=g_ft.tolist()))
    # which gradient blocks carry the error (last stage, frame 0): betas 0:10 | go 10:13 | dead 13:76 | hands 76:100 |
    # jaw/eyes 100:109 | expr 109:119 | pose 119:182
    stage = fbs["rows"].n_stages - 1
    f64, g64 = T._oracle_closure(model, cfg, frames, 0, Q, stage, dtype=torch.float64)
    f32, g32 = T._oracle_closure(model, cfg, frames, 0, Q, stage, dtype=torch.float32)
    gh = fbs["rows"].closure(stage)[1][0]
    blocks = dict(betas=(0, 10), go=(10, 13), hands=(76, 100), jaw_eyes=(100, 109), expr=(109, 119), pose=(119, 182))
    out["blocks_frame0"] = {}
    for k, (a, b) in blocks.items():
        out["blocks_frame0"][k] = dict(g64=float(np.linalg.norm(g64[a:b])), hip_err=float(np.linalg.norm(gh[a:b] - g64[a:b])),
                                       torch32_err=float(np.linalg.norm(g32[a:b] - g64[a:b])))
        print("   block %-8s |g64| %.3e  hip err %.3e  torch32 err %.3e" % (k, out["blocks_frame0"][k]["g64"],
                                                                            out["blocks_frame0"][k]["hip_err"],
                                                                            out["blocks_frame0"][k]["torch32_err"]))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"))


if __name__ == "__main__":
    main()
