"""Batched front-end with the interface and the output layout of smplifyx/main.py:50-323.

The reference walks the dataset and calls fit_single_frame() per image and person, one at a time
(main.py:207-318).  Here every (image, person) becomes one frame of a batch; frames that share a
body model (gender) and an image size are fitted together by driver.fit_frames -- the whole
schedule of all of them runs on the GPU at once -- and the same files are written:

    <output_folder>/conf.yaml
    <output_folder>/<result_folder>/<fn>/<person:03d>.pkl      result dict of fit_single_frame
    <output_folder>/<result_folder>/<fn>/vertices.ply          (save_vertices)
    <output_folder>/<mesh_folder>/<fn>/                        (created, as in the reference)
    <output_folder>/images/<fn>/<person:03d>/                  (created, as in the reference)

With torch.distributed initialised (one process per GPU), rank r takes the frames
dist.shard_by_cost gives it (known stragglers dealt round-robin); there is no collective in the data path (each rank writes its own
result files).
"""
import os
import warnings
import os.path as osp
import pickle
import shutil
import sys
import time

import numpy as np
import torch
import yaml

from . import dist as sdist
from . import driver, utils
from . import smplx
from . import vposer as vposer_host
from .data_parser import create_dataset
from .fit_single_frame import _write_ply, setup_interpenetration


def _load_regression(args, img_name):
    """main.py:277-289: per-image regression results (joblib / npz files)."""
    pixie = expose = pare = None
    if args.get("regression_prior"):
        if args.get("pixie_results_directory"):
            import joblib
            pixie = joblib.load(osp.join(args["pixie_results_directory"], img_name, img_name + "_param.pkl"))
        if args.get("expose_results_directory"):
            expose = np.load(osp.join(args["expose_results_directory"], img_name + ".jpg",
                                      img_name + ".jpg" + "_params.npz"), allow_pickle=True)
        if args.get("pare_results_directory"):
            import joblib
            pare = joblib.load(osp.join(args["pare_results_directory"], img_name + ".pkl"))
    return pixie, expose, pare


def _camera_prior(regression_prior, focal_length, pixie, expose, pare):
    """fit_single_frame.py:359-398 -> (translation[3], center[2])."""
    if regression_prior in ("ExPose", "combined"):
        t = np.array(expose["transl"], np.float64)
        t[-1] /= (5000 / focal_length)
        return t, np.asarray(expose["center"], np.float32)
    if regression_prior == "PIXIE":
        left, top, right, bottom = pixie["bbox"]
        size = int(max(right - left, bottom - top) * 1.1)
        pc = pixie["body_cam"]
        return (np.array([pc[1], pc[2], 2 * focal_length / (pc[0] * size + 1e-9)]),
                np.array([right - (right - left) / 2.0, bottom - (bottom - top) / 2.0], np.float32))
    if regression_prior == "PARE":
        cx, cy, bb, _ = pare["bboxes"][0]
        pc = pare["pred_cam"][0]
        return np.array([pc[1], pc[2], (2 * focal_length) / (bb * pc[0])]), np.array([cx, cy], np.float32)
    raise ValueError("unknown regression prior %r" % regression_prior)


def rank_share(items, cfg, joint_weights, rank, world):
    """This rank's frames of a multi-GPU job: the frames known to be long before the fit (driver.predicted_cost: side views,
    under-determined cameras) are dealt over the ranks back and forth (dist.shard_by_cost) -- a contiguous block
    (dist.shard_range) can hand one GPU all the side views of a sequence.  Every frame goes to exactly one rank; a rank walks
    its frames in input order."""
    if world <= 1 or not items:
        return items
    kp_all = np.stack([it["keypoints"] for it in items]).astype(np.float32)
    cost = driver.predicted_cost(cfg, driver.prepare_frames(cfg, kp_all, joint_weights))
    return [items[i] for i in sdist.shard_by_cost(cost, rank, world)]


def main(**args):
    output_folder = osp.expandvars(args.pop("output_folder"))
    rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) \
        if torch.distributed.is_available() and torch.distributed.is_initialized() else (0, 1)
    if rank == 0:
        if osp.exists(output_folder):
            shutil.rmtree(output_folder)
        os.makedirs(output_folder)
        with open(osp.join(output_folder, "conf.yaml"), "w") as fh:
            yaml.dump(args, fh)
    if world > 1:
        torch.distributed.barrier()
    result_folder = osp.join(output_folder, args.pop("result_folder", "results"))
    mesh_folder = osp.join(output_folder, args.pop("mesh_folder", "meshes"))
    for d in (result_folder, mesh_folder, osp.join(output_folder, "images")):
        os.makedirs(d, exist_ok=True)
    if args.get("float_dtype", "float32") not in ("float32", "float64"):
        raise ValueError("Unknown float type {}, exiting!".format(args.get("float_dtype")))      # main.py:99-105
    if args.get("float_dtype", "float32") == "float64":
        warnings.warn("float_dtype: float64 selects the engine's high-precision mode (keypoint forward AND projection in fp64; "
                      "parameters, reverse sweep and optimiser stay fp32): the fits behave like the reference's float64 run "
                      "(LAB_NOTES.md §3.1), they are not an end-to-end float64 evaluation")
    if args.get("use_cuda", True) and not torch.cuda.is_available():
        print("CUDA is not available, exiting!")
        sys.exit(-1)
    interpenetration = bool(args.get("interpenetration", True))
    # flags every shipped cfg sets that lie outside the fitting path: they must not stop a run of the unmodified cfg
    if args.get("use_gender_classifier", False):
        # main.py:197-200,258-262: the external homogenus network picks the model's gender per image; without it the
        # cfg's `gender` is used, which is also what the reference does when the flag is off
        warnings.warn("use_gender_classifier: the homogenus classifier is not part of this engine; using cfg gender %r"
                      % args.get("gender", "neutral"))
    if args.get("visualize", False) or args.get("interactive", False):
        warnings.warn("visualize / interactive: rendering and progress output are outside the fitting path; the fit itself "
                      "is unaffected")
    args["visualize"] = False
    if not args.get("use_joints_conf", False):
        raise NameError("name 'joints_conf' is not defined")   # the reference fails here (fit_single_frame.py:286)

    img_folder = args.pop("img_folder", "images")
    dataset_obj = create_dataset(img_folder=img_folder, **args)
    start = time.time()
    input_gender = args.pop("gender", "neutral")
    max_persons = args.pop("max_persons", -1)
    joint_map = dataset_obj.get_model2data()
    joint_weights = dataset_obj.get_joint_weights().numpy()
    use_vposer = bool(args.get("use_vposer", True))
    vpw = vposer_host.load_vposer(args["vposer_ckpt"]) if use_vposer else None
    regression_prior = args.get("regression_prior", None)

    # ---- gather frames: (image, person) pairs of this rank, grouped by (gender, H, W) ------------
    items = []
    for data in dataset_obj:
        if not data:
            continue
        H_, W_, _ = data["img"].shape
        keypoints = data["keypoints"]
        for person_id in range(keypoints.shape[0]):
            if (person_id >= max_persons and max_persons > 0) or person_id > 0:      # main.py:244-247
                continue
            img_name = data["img_path"].split("images")[-1].split(".")[0].lstrip("/\\")
            items.append(dict(fn=data["fn"], person=person_id, H=H_, W=W_, img_name=img_name,
                              keypoints=keypoints[person_id], gender=input_gender))
    items = rank_share(items, args, joint_weights, rank, world)
    groups = {}
    for it in items:
        groups.setdefault((it["gender"], it["H"], it["W"]), []).append(it)

    models = {}
    n_done = 0
    for (gender, H_, W_), its in groups.items():
        if gender not in models:
            models[gender] = smplx.create(model_path=args.get("model_folder"), gender=gender,
                                          joint_mapper=utils.JointMapper(joint_map), create_global_orient=True,
                                          create_body_pose=not use_vposer, create_betas=True,
                                          create_left_hand_pose=True, create_right_hand_pose=True,
                                          create_expression=True, create_jaw_pose=True, create_leye_pose=True,
                                          create_reye_pose=True, create_transl=False, dtype=torch.float32,
                                          vposer=vpw, **{k: v for k, v in args.items() if k not in ("model_path", "dtype")})
        bm = models[gender]
        dm = bm.device_model
        if interpenetration:
            setup_interpenetration(dm, args.get("part_segm_fn", ""), args.get("ign_part_pairs"))
        focal = args.get("focal_length", None)
        if focal is None:
            focal = (W_ ** 2 + H_ ** 2) ** 0.5                                   # main.py:213-214
        B = len(its)
        kp = np.stack([it["keypoints"] for it in its]).astype(np.float32)
        reg_pose = reg_glob = cam_t = cam_c = None
        if regression_prior:
            rp, rg, ct, cc = [], [], [], []
            for it in its:
                pixie, expose, pare = _load_regression(args, it["img_name"])
                p, g_ = utils.regression_prior_pose(regression_prior, expose=expose, pixie=pixie, pare=pare)
                rp.append(np.asarray(p, np.float32).reshape(-1)); rg.append(np.asarray(g_, np.float32).reshape(-1))
                if args.get("use_camera_prior"):
                    t, c = _camera_prior(regression_prior, focal, pixie, expose, pare)
                    ct.append(t); cc.append(c)
            reg_pose, reg_glob = np.stack(rp), np.stack(rg)
            if use_vposer:
                seed = args.get("vposer_sample_seed")
                reg_pose = vposer_host.encode(vpw, reg_pose, generator=None if seed is None else np.random.default_rng(seed))
            if ct:
                cam_t, cam_c = np.stack(ct), np.stack(cc)
        cfg = dict(args)
        cfg.update(focal_length=focal, left_shoulder_idx=dataset_obj.get_left_shoulder(),
                   right_shoulder_idx=dataset_obj.get_right_shoulder())
        res = driver.fit_frames(dm, cfg, kp, joint_weights, H_, W_, focal, reg_pose=reg_pose, reg_global=reg_glob,
                                cam_prior_t=cam_t, cam_prior_center=cam_c,
                                lbs_mode="dense" if interpenetration else args.get("lbs_mode", "rows"),
                                reuse_entry_eval=True, want_vertices=bool(args.get("save_vertices")))
        names = [n for n, _ in bm.named_parameters()]
        for b, it in enumerate(its):
            curr_result_folder = osp.join(result_folder, it["fn"])
            for d in (curr_result_folder, osp.join(mesh_folder, it["fn"]),
                      osp.join(output_folder, "images", it["fn"], "{:03d}".format(it["person"]))):
                os.makedirs(d, exist_ok=True)
            center = cam_c[b] if cam_c is not None else np.array([W_, H_], np.float32) * 0.5
            result = {"camera_rotation": np.eye(3, dtype=np.float32)[None], "camera_translation": res["cam_translation"][b:b + 1],
                      "camera_center": np.asarray(center, np.float32)[None], "H": H_, "W": W_, "focal_length": focal}
            for n in names:                                                      # fit_single_frame.py:650-651
                result[n] = res[n][b:b + 1] if n in res else getattr(bm, n).detach().cpu().numpy()
            result["body_pose"] = res["body_pose"][b:b + 1]
            with open(osp.join(curr_result_folder, "{:03d}.pkl".format(it["person"])), "wb") as fh:
                pickle.dump(result, fh, protocol=2)
            if args.get("save_vertices"):
                _write_ply(osp.join(curr_result_folder, "vertices.ply"), res["vertices"][b])
            n_done += 1
    elapsed = time.time() - start
    print("Processing the data took: {}".format(time.strftime("%H hours, %M minutes, %S seconds", time.gmtime(elapsed))))
    return n_done


if __name__ == "__main__":
    from .cmd_parser import parse_config
    main(**parse_config())
