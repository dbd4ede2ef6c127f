"""Configuration contract of the fitting path: the keys, types and defaults of the
reference's flag table (smplifyx/cmd_parser.py:37-301) and its YAML files
(cfg_files/*.yaml), without configargparse (not installed here).

`parse_config(argv)` accepts `--config file.yaml` plus `--key value ...` overrides and
returns the same dict the reference's parser returns, including
  * PyYAML reads `1e-9`, `4.04e2` as strings -> cast by the declared type,
  * booleans from 'true'/'1' strings,
  * `jaw_pose_prior_weights` kept as comma strings (split later, fit_single_frame.py:178),
  * `body_tri_idxs` flat list -> list of pairs (cmd_parser.py:307-316),
  * `focal_length` default None (main.py:212-214 then uses sqrt(W^2+H^2)).
"""
import os

import yaml


def _b_true(x):
    return str(x).lower() == "true"


def _b_true1(x):
    return str(x).lower() in ["true", "1"]


# key: (type or None, default, is_list)
_TABLE = {
    "data_folder": (str, None, False),
    "max_persons": (int, 3, False),
    "loss_type": (str, "smplify", False),
    "interactive": (_b_true, False, False),
    "save_meshes": (_b_true, True, False),
    "visualize": (_b_true, False, False),
    "degrees": (float, [0, 90, 180, 270], True),
    "use_cuda": (_b_true, True, False),
    "format": (str, "coco_wholebody", False),
    "joints_to_ign": (int, -1, True),
    "output_folder": (str, "output", False),
    "img_folder": (str, "images", False),
    "keyp_folder": (str, "keypoints", False),
    "summary_folder": (str, "summaries", False),
    "result_folder": (str, "results", False),
    "mesh_folder": (str, "meshes", False),
    "gender": (str, "neutral", False),
    "float_dtype": (str, "float32", False),
    "model_type": (str, "smpl", False),
    "camera_type": (str, "persp", False),
    "optim_jaw": (_b_true1, True, False),
    "optim_hands": (_b_true1, True, False),
    "optim_expression": (_b_true1, True, False),
    "optim_shape": (_b_true1, True, False),
    "model_folder": (str, "models", False),
    "use_joints_conf": (_b_true1, True, False),
    "batch_size": (int, 1, False),
    "num_gaussians": (int, 8, False),
    "use_pca": (_b_true1, True, False),
    "num_pca_comps": (int, 6, False),
    "flat_hand_mean": (_b_true1, False, False),
    "body_prior_type": (str, "mog", False),
    "left_hand_prior_type": (str, "mog", False),
    "right_hand_prior_type": (str, "mog", False),
    "jaw_prior_type": (str, "l2", False),
    "use_vposer": (_b_true1, False, False),
    "vposer_ckpt": (str, "", False),
    "init_joints_idxs": (int, [9, 12, 2, 5], True),
    "body_tri_idxs": (int, [5, 12, 2, 9], True),
    "prior_folder": (str, "prior", False),
    "focal_length": (float, None, False),
    "rho": (float, 100, False),
    "interpenetration": (_b_true1, False, False),
    "penalize_outside": (_b_true1, False, False),
    "data_weights": (float, None, True),
    "body_pose_prior_weights": (float, [4.04 * 1e2, 4.04 * 1e2, 57.4, 4.78], True),
    "shape_weights": (float, [1e2, 5 * 1e1, 1e1, .5 * 1e1], True),
    "expr_weights": (float, [1e2, 5 * 1e1, 1e1, .5 * 1e1], True),
    "face_joints_weights": (float, [0.0, 0.0, 0.0, 2.0], True),
    "hand_joints_weights": (float, [0.0, 0.0, 0.0, 2.0], True),
    "jaw_pose_prior_weights": (None, None, True),
    "hand_pose_prior_weights": (float, [1e2, 5 * 1e1, 1e1, .5 * 1e1], True),
    "coll_loss_weights": (float, [0.0, 0.0, 0.0, 2.0], True),
    "depth_loss_weight": (float, 1e2, False),
    "df_cone_height": (float, 0.5, False),
    "max_collisions": (int, 8, False),
    "point2plane": (_b_true1, False, False),
    "part_segm_fn": (str, "", False),
    "ign_part_pairs": (str, None, True),
    "use_hands": (_b_true1, False, False),
    "use_face": (_b_true1, False, False),
    "use_face_contour": (_b_true1, False, False),
    "side_view_thsh": (float, 25, False),
    "optim_type": (str, "adam", False),
    "lr": (float, 1e-6, False),
    "gtol": (float, 1e-8, False),
    "ftol": (float, 2e-9, False),
    "maxiters": (int, 100, False),
    "num_betas": (int, 10, False),
    "num_expression_coeffs": (int, 10, False),
    "regression_prior": (str, None, False),
    "pixie_results_directory": (str, None, False),
    "expose_results_directory": (str, None, False),
    "pare_results_directory": (str, None, False),
    "homogeneous_ckpt": (str, "./homogeneous/trained_models/tf/", False),
    "use_camera_prior": (_b_true, False, False),
    "use_conf_for_camera_init": (_b_true, False, False),
    "use_gender_classifier": (_b_true, False, False),
    "save_vertices": (_b_true, False, False),
    "confidence_threshold": (float, 0, False),
}

_CHOICES = {
    "format": ["coco25", "halpe", "coco_wholebody"],
    "gender": ["neutral", "male", "female"],
    "model_type": ["smpl", "smplh", "smplx"],
    "camera_type": ["persp"],
    "regression_prior": ["PIXIE", "ExPose", "PARE", "combined", None],
}


def _cast(key, value):
    typ, _, is_list = _TABLE[key]
    conv = (lambda v: v if typ is None else typ(v))
    if is_list:
        if not isinstance(value, (list, tuple)):
            value = [value]
        out = []
        for v in value:
            out.append(str(v) if typ in (None, str) else conv(v))
        return out
    if value is None:
        return None
    if typ is str:
        return str(value)
    return conv(value)


def load_config(config=None, overrides=None):
    """dict of every key of the contract; `config` = YAML path, `overrides` = dict."""
    args = {k: (list(d) if isinstance(d, list) else d) for k, (_, d, _) in _TABLE.items()}
    args["data_folder"] = os.getcwd()
    args["config"] = config
    raw = {}
    if config is not None:
        with open(config) as f:
            raw.update(yaml.safe_load(f) or {})
    raw.update(overrides or {})
    for key, value in raw.items():
        if key in _TABLE:
            args[key] = _cast(key, value)
        else:
            args[key] = value          # the reference's callees ignore unknown keys (**kwargs sinks)
    for key, allowed in _CHOICES.items():
        if args.get(key) not in allowed:
            raise ValueError("argument --{}: invalid choice: {!r}".format(key, args.get(key)))
    tri = args["body_tri_idxs"]
    if len(tri) % 2 != 0:
        raise AssertionError("Number of body_tri_idxs arguments must be divisble by 2. Got: %d" % len(tri))
    args["body_tri_idxs"] = [(tri[i], tri[i + 1]) for i in range(0, len(tri), 2)]
    return args


def parse_config(argv=None):
    """`--config X.yaml [--key value ...]`; list flags take several values."""
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    config, overrides, i = None, {}, 0
    while i < len(argv):
        tok = argv[i]
        if not tok.startswith("-"):
            raise ValueError("unexpected argument %r" % tok)
        key = tok.lstrip("-")
        vals = []
        i += 1
        while i < len(argv) and not (argv[i].startswith("--") or argv[i] == "-c"):
            vals.append(argv[i]); i += 1
        if key in ("c", "config"):
            config = vals[0]
        elif key in _TABLE and _TABLE[key][2]:
            overrides[key] = vals
        else:
            overrides[key] = vals[0] if vals else "true"
    if config is None:
        raise ValueError("the following arguments are required: -c/--config")
    return load_config(config, overrides)
