"""ctypes binding of include/sfx.h (libsfx.so).  Thin: argument marshalling only.

The library is the product path; there is no Python/CPU fallback.  `load()` raises if the
shared object is missing (run `python __graft_entry__.py` or csrc/build.sh) and
`sfx_model_create` fails loudly when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFX_LIB", os.path.join(_HERE, "libsfx.so"))
_lib = None

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


class ModelDesc(C.Structure):
    _fields_ = [
        ("V", C.c_int32), ("F", C.c_int32), ("J", C.c_int32),
        ("num_betas", C.c_int32), ("num_expr", C.c_int32), ("num_pca", C.c_int32),
        ("v_template", f32p), ("shapedirs", f32p), ("posedirs", f32p), ("J_regressor", f32p),
        ("lbs_weights", f32p), ("parents", i32p), ("hands_comp_l", f32p), ("hands_comp_r", f32p),
        ("pose_mean", f32p), ("faces", i32p),
        ("n_extra", C.c_int32), ("extra_vertex_ids", i32p),
        ("n_lmk", C.c_int32), ("lmk_faces_idx", i32p), ("lmk_bary", f32p),
        ("n_dyn_rows", C.c_int32), ("n_dyn", C.c_int32), ("dyn_lmk_faces_idx", i32p), ("dyn_lmk_bary", f32p),
        ("K", C.c_int32), ("joint_map", i32p),
    ]


class StageWeights(C.Structure):
    _fields_ = [
        ("body_pose_weight", C.c_float), ("shape_weight", C.c_float),
        ("hand_prior_weight", C.c_float), ("expr_prior_weight", C.c_float),
        ("jaw_prior_weight", C.c_float * 3),
        ("hand_joint_weight", C.c_float), ("face_joint_weight", C.c_float),
        ("coll_loss_weight", C.c_float), ("bending_prior_weight", C.c_float),
    ]


class BatchCfg(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("n_stages", C.c_int32), ("use_vposer", C.c_int32),
        ("use_hands", C.c_int32), ("use_face", C.c_int32), ("use_joints_conf", C.c_int32),
        ("has_regression_pose", C.c_int32), ("use_conf_cam_init", C.c_int32),
        ("num_body_joints", C.c_int32), ("maxiters", C.c_int32),
        ("ftol", C.c_double), ("gtol", C.c_double),
        ("lr", C.c_float), ("rho", C.c_float), ("depth_loss_weight", C.c_float),
        ("lbs_mode", C.c_int32), ("reuse_entry_eval", C.c_int32),
        ("side_view_thsh", C.c_float), ("left_shoulder_idx", C.c_int32), ("right_shoulder_idx", C.c_int32),
        ("interpenetration", C.c_int32), ("max_collisions", C.c_int32), ("df_cone_height", C.c_float),
        ("penalize_outside", C.c_int32), ("slots", C.c_int32),
        ("lbfgs_tolerance_grad", C.c_double), ("lbfgs_tolerance_change", C.c_double),
        ("lbfgs_max_eval", C.c_int32), ("lbfgs_history_size", C.c_int32), ("lbfgs_max_iter", C.c_int32), ("high_precision", C.c_int32),
        ("point2plane", C.c_int32),
    ]


# every symbol include/sfx.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sfx_model_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "sfx_model_destroy": (None, [C.c_void_p]),
    "sfx_model_set_parts": (C.c_int, [C.c_void_p, i32p, i32p, i32p, C.c_int32]),
    "sfx_model_set_vposer": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32] + [f32p] * 6),
    "sfx_lbs_forward": (C.c_int, [C.c_void_p, C.c_int32] + [C.c_void_p] * 12 + [C.c_void_p]),
    "sfx_batch_create": (C.c_int, [C.c_void_p, C.POINTER(BatchCfg), C.POINTER(StageWeights), C.POINTER(C.c_void_p)]),
    "sfx_batch_destroy": (None, [C.c_void_p]),
    "sfx_batch_set_frames": (C.c_int, [C.c_void_p] + [f32p] * 5),
    "sfx_batch_set_params": (C.c_int, [C.c_void_p] + [f32p] * 11),
    "sfx_batch_get_params": (C.c_int, [C.c_void_p] + [f32p] * 11),
    "sfx_batch_num_vars": (C.c_int, [C.c_void_p, C.c_int32]),
    "sfx_batch_closure": (C.c_int, [C.c_void_p, C.c_int32, f32p, f32p, C.c_void_p]),
    "sfx_batch_guess_init": (C.c_int, [C.c_void_p, i32p, C.c_int32, C.c_void_p]),
    "sfx_batch_fit": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sfx_batch_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, f32p, C.c_void_p]),
    "sfx_batch_pen_stats": (C.c_int, [C.c_void_p, i32p, i32p]),
    "sfx_batch_get_grad": (C.c_int, [C.c_void_p, C.c_int32, f32p]),
    "sfx_batch_get_stats": (C.c_int, [C.c_void_p, f32p, i32p, i32p]),
    "sfx_lbfgs_two_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sfx_batch_trace": (C.c_int, [C.c_void_p, C.c_int32]),
    "sfx_batch_get_trace": (C.c_int, [C.c_void_p, f32p, i32p]),
    "sfx_batch_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sfx_pen_create": (C.c_int, [C.c_int32, C.c_int32, i32p, i32p, i32p, i32p, C.c_int32, C.c_int32, C.c_int32,
                                 C.POINTER(C.c_void_p)]),
    "sfx_pen_destroy": (None, [C.c_void_p]),
    "sfx_pen_eval": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sfx_pen_set_point2plane": (C.c_int, [C.c_void_p, C.c_int32]),
    "sfx_pen_eval_pairs": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "sfx_pen_stats": (C.c_int, [C.c_void_p, C.c_int32, i32p]),
    "sfx_pen_pairs": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, i32p, C.POINTER(C.c_int32)]),
    "sfx_batch_pen_pairs": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, i32p, C.POINTER(C.c_int32)]),
    "sfx_batch_pen_flags": (C.c_int, [C.c_void_p, i32p]),
    "sfx_batch_pen_launches": (C.c_int, [C.c_void_p]),
    "sfx_pen_work_reset": (C.c_int, []),
    "sfx_pen_work_get": (C.c_int, [C.POINTER(C.c_int64)]),
    "sfx_batch_set_gmm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, f32p, f32p, f32p]),
    "sfx_batch_set_gmm_form": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, f32p, f32p, f32p, f32p]),
    "sfx_batch_debug_read": (C.c_int, [C.c_void_p, C.c_char_p, f32p, C.c_int64]),
    "sfx_loop_host_stats": (C.c_int, [C.POINTER(C.c_double), C.c_int32]),
    "sfx_prof_enable": (C.c_int, [C.c_int32]),
    "sfx_prof_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "sfx_prof_reset": (None, []),
    "sfx_last_error": (C.c_char_p, []),
    "sfx_version": (C.c_char_p, []),
}

# what include/sfx_lab.h adds (libsfx_lab.so: csrc/build.sh with SFX_LAB=1; selected with SFX_LIB=<path> in the environment)
LAB_SYMBOLS = {
    "sfx_debug_lbs_dense_form": (C.c_int, [C.c_int32]),
    "sfx_debug_phase_clocks": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "sfx_pen_phase_clocks": (C.c_int, [C.c_void_p, C.c_int32, i32p]),
    "sfx_debug_pen_form": (C.c_int, [C.c_int32]),
    "sfx_debug_pen_phase_ticks": (C.c_int, [C.POINTER(C.c_int64)]),
    "sfx_debug_clocks": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
}
_has_lab = False


def has_lab():
    """True when the loaded library is the lab build (A/B forms, phase clocks: include/sfx_lab.h)."""
    load()
    return _has_lab


def need_lab(what):
    if not has_lab():
        raise RuntimeError("%s needs the lab build of the library (include/sfx_lab.h): SFX_LAB=1 bash "
                           "smplify-x-partial_amd/csrc/build.sh, then SFX_LIB=smplify-x-partial_amd/libsfx_lab.so" % what)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libsfx.so not built (%s): run `python __graft_entry__.py` "
                           "or smplify-x-partial_amd/csrc/build.sh; there is no CPU fallback" % LIB_PATH)
    # torch ships its own copy of the HIP runtime: it must be the one this process initialises, so
    # that libsfx.so (linked against libamdhip64 by soname) shares it with the torch tensors whose
    # data_ptr()s it is handed.  Loading libsfx first makes two runtimes meet in one process and
    # the second sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    global _has_lab
    _has_lab = all(hasattr(lib, name) for name in LAB_SYMBOLS)
    if _has_lab:
        for name, (res, args) in LAB_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


class SfxError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise SfxError("libsfx error %d: %s" % (rc, load().sfx_last_error().decode()))


def fptr(a):
    """numpy float32 C-contiguous array -> float* (None -> NULL).  Keeps no reference."""
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(f32p)


def iptr(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(i32p)


def f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)
