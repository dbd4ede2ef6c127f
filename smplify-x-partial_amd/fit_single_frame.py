"""Drop-in for smplifyx/fit_single_frame.py:59-677 -- same signature, same result pickle
(keys and order of :644-657, protocol 2), same optional vertices.ply -- with the whole
optimisation executed by the MI355X engine (driver.fit_frames with a batch of one).
Not reproduced: visualisation (`visualize=True`).  `interpenetration=True` runs the penetration
term of csrc/collide.hip in the stages with coll_loss_weight > 0 (always through the dense path).
`vposer.encode(prior).sample()` (:245, random in the reference) uses the posterior mean unless
`vposer_sample_seed` is given.  kwargs['lbs_mode'] = 'dense' evaluates all 10475 vertices in
every closure call as the reference does; the default 'rows' evaluates the rows the loss reads
(same objective, same optimiser; the mesh is produced once at the end).
"""
import os
import pickle

import warnings

import numpy as np
import torch

from . import driver
from . import utils
from . import vposer as vposer_host


def _write_ply(path, vertices):
    v = np.asarray(vertices, "<f4")
    with open(path, "wb") as fh:
        fh.write(("ply\nformat binary_little_endian 1.0\nelement vertices %d\nproperty float x\n"
                  "property float y\nproperty float z\nend_header\n" % v.shape[0]).encode("ascii"))
        fh.write(v.tobytes())


def setup_interpenetration(dm, part_segm_fn, ign_part_pairs):
    """fit_single_frame.py:316-328: per-face part labels from `part_segm_fn` (pickle with 'segm' and
    'parents') and the ignored part pairs -> the device model.  Without a file no pair is filtered
    by part, as in the reference (filter_faces = None)."""
    if getattr(dm, "has_parts", False) or not part_segm_fn:
        return
    with open(os.path.expandvars(part_segm_fn), "rb") as fh:
        data = pickle.load(fh, encoding="latin1")
    dm.set_parts(data["segm"], data["parents"], ign_part_pairs)


def fit_single_frame(img, keypoints, body_model, camera, joint_weights, body_pose_prior, jaw_prior,
                     left_hand_prior, right_hand_prior, shape_prior, expr_prior, angle_prior,
                     result_fn="out.pkl", mesh_fn="out.obj", loss_type="smplify", use_cuda=True,
                     init_joints_idxs=(9, 12, 2, 5), use_face=True, use_hands=True, data_weights=None,
                     body_pose_prior_weights=None, hand_pose_prior_weights=None, jaw_pose_prior_weights=None,
                     shape_weights=None, expr_weights=None, hand_joints_weights=None, face_joints_weights=None,
                     global_orient_weights=None, depth_loss_weight=1e2, interpenetration=True, coll_loss_weights=None,
                     df_cone_height=0.5, penalize_outside=True, max_collisions=8, point2plane=False, part_segm_fn="",
                     focal_length=5000., side_view_thsh=25., rho=100, vposer_latent_dim=32, vposer_ckpt="",
                     use_joints_conf=False, interactive=True, visualize=False, degrees=None, batch_size=1,
                     dtype=torch.float32, ign_part_pairs=None, left_shoulder_idx=2, right_shoulder_idx=5,
                     result_folder=".", img_name="", pixie_results=None, expose_results=None, pare_results=None,
                     regression_prior=None, format="coco25", smplx_path="", curr_img_folder=".", **kwargs):
    assert batch_size == 1, "fit_single_frame handles one frame; use driver.fit_frames for batches"
    if visualize:      # (every shipped cfg sets it) rendering is outside the fitting path: the fit runs, nothing is drawn
        warnings.warn("visualize=True: no images are rendered by this engine; the fit itself is unaffected")
    if not use_cuda:
        raise RuntimeError("use_cuda=False: this engine has no CPU path")
    H, W, _ = np.asarray(img).shape
    use_vposer = kwargs.get("use_vposer", True)
    cfg = dict(kwargs)
    cfg.update(init_joints_idxs=list(init_joints_idxs), use_face=use_face, use_hands=use_hands,
               data_weights=data_weights, body_pose_prior_weights=body_pose_prior_weights,
               hand_pose_prior_weights=hand_pose_prior_weights, jaw_pose_prior_weights=jaw_pose_prior_weights,
               shape_weights=shape_weights, expr_weights=expr_weights, hand_joints_weights=hand_joints_weights,
               face_joints_weights=face_joints_weights, global_orient_weights=global_orient_weights,
               depth_loss_weight=depth_loss_weight, interpenetration=bool(interpenetration),
               coll_loss_weights=coll_loss_weights, max_collisions=max_collisions, df_cone_height=df_cone_height,
               penalize_outside=penalize_outside, point2plane=bool(point2plane),
               side_view_thsh=side_view_thsh, rho=rho, use_joints_conf=use_joints_conf, format=format,
               left_shoulder_idx=left_shoulder_idx, right_shoulder_idx=right_shoulder_idx, use_vposer=use_vposer)
    if not use_joints_conf:
        raise NameError("name 'joints_conf' is not defined")   # the reference fails here (fit_single_frame.py:286)
    dm = body_model.device_model
    if interpenetration:
        setup_interpenetration(dm, part_segm_fn, ign_part_pairs)
    if use_vposer and not dm.vposer_latent:         # load_vposer(vposer_ckpt, vp_model='snapshot') (:239-242)
        dm.set_vposer(vposer_host.load_vposer(vposer_ckpt))
    reg_pose = reg_glob = cam_t = cam_c = None
    if regression_prior:
        reg_pose, reg_glob = utils.regression_prior_pose(regression_prior, expose=expose_results, pixie=pixie_results,
                                                         pare=pare_results)
        if use_vposer:
            # pose_embedding = vposer.encode(full_pose_prior).sample() (:245); the same latent is the
            # regression target of the last stage (:442, fitting.py:391-393).  The posterior MEAN is
            # used unless kwargs['vposer_sample_seed'] asks for a (seeded) sample.
            seed = kwargs.get("vposer_sample_seed")
            gen = np.random.default_rng(seed) if seed is not None else None
            reg_pose = vposer_host.encode(dm.vposer_weights, reg_pose, generator=gen)
        if kwargs.get("use_camera_prior"):
            if regression_prior in ("ExPose", "combined"):
                cam_c = np.asarray(expose_results["center"], np.float32)
                cam_t = np.array(expose_results["transl"], np.float64)
                cam_t[-1] /= (5000 / focal_length)
            elif regression_prior == "PIXIE":
                left, top, right, bottom = pixie_results["bbox"]
                size = int(max(right - left, bottom - top) * 1.1)
                ctr = np.array([right - (right - left) / 2.0, bottom - (bottom - top) / 2.0])
                cam_c = np.array([ctr[0], ctr[1]], np.float32)
                pc = pixie_results["body_cam"]
                cam_t = np.array([pc[1], pc[2], 2 * focal_length / (pc[0] * size + 1e-9)])
            elif regression_prior == "PARE":
                cx, cy, bb, _ = pare_results["bboxes"][0]
                pc = pare_results["pred_cam"][0]
                cam_c = np.array([cx, cy], np.float32)
                cam_t = np.array([pc[1], pc[2], (2 * focal_length) / (bb * pc[0])])
    kp = np.asarray(keypoints, np.float32).reshape(1, dm.K, 3)
    jw = joint_weights.detach().cpu().numpy() if torch.is_tensor(joint_weights) else np.asarray(joint_weights)
    want_v = bool(kwargs.get("save_vertices"))
    res = driver.fit_frames(dm, cfg, kp, jw.reshape(1, -1), H, W, focal_length, reg_pose=reg_pose, reg_global=reg_glob,
                            cam_prior_t=cam_t, cam_prior_center=cam_c,
                            lbs_mode="dense" if interpenetration else kwargs.get("lbs_mode", "rows"),
                            reuse_entry_eval=True, want_vertices=want_v,
                            body_pose_prior=body_pose_prior if hasattr(body_pose_prior, "get_mean") else None)
    # write the fitted values back into the caller's modules, as the reference leaves them
    with torch.no_grad():
        camera.translation[:] = torch.as_tensor(res["cam_translation"]).to(camera.translation)
        if cam_c is not None:
            camera.center[:] = torch.as_tensor(cam_c).to(camera.center)
        else:
            camera.center[:] = torch.tensor([W, H], dtype=camera.center.dtype, device=camera.center.device) * 0.5
        for name, p in body_model.named_parameters():
            if name in res:
                p[:] = torch.as_tensor(res[name]).to(p)
    result = {"camera_rotation": camera.rotation.detach().cpu().numpy(),
              "camera_translation": res["cam_translation"], "camera_center": camera.center.detach().cpu().numpy(),
              "H": H, "W": W, "focal_length": focal_length}
    for name, p in body_model.named_parameters():
        result[name] = res[name] if name in res else p.detach().cpu().numpy()
    result["body_pose"] = res["body_pose"]
    with open(result_fn, "wb") as fh:
        pickle.dump(result, fh, protocol=2)
    if want_v:
        _write_ply(os.path.join(result_folder, "vertices.ply"), res["vertices"][0])
    return result, float(res["final_loss"][0])
