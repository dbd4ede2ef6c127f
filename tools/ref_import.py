"""Import the reference's in-tree hot-path modules in THIS container (never on the GPU
box: /root/reference does not exist there and nothing under tests/-m gpu, smoke() or
bench.py imports this file).

The reference imports visualisation-only packages that are not installed
(open3d, trimesh, pyrender, cv2, plyfile, human_body_prior, configargparse, smplx):
they are replaced by MagicMock stubs, except `smplx.lbs.transform_mat`, a 2-line pure
function the camera needs (smplifyx/camera.py:27,102), which is supplied from the
oracle restatement.  No reference source is copied; modules are imported from where
they lie.
"""
import os
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = os.environ.get("SFX_REFERENCE_ROOT", "/root/reference")
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "smplifyx"))


def install_mesh_intersection(faces=None, part_filter=None):
    """Register oracle/mesh_intersection_cpu.py (CPU stand-ins built on oracle/penetration.py) under the three module names of
    the external CUDA package the reference imports at fit_single_frame.py:301-303, so that the reference's own interpenetration
    lines (fit_single_frame.py:300-328, fitting.py:437-455) run here.  faces: the model's faces (the stand-in BVH cannot see
    vertex ids in a triangles tensor); part_filter: (segm, parents, ign_part_pairs) for assumption A1's order of operations."""
    if REPO_ROOT not in sys.path:
        sys.path.insert(0, REPO_ROOT)
    from oracle import mesh_intersection_cpu as M
    pkg = types.ModuleType("mesh_intersection")
    for name in ("bvh_search_tree", "loss", "filter_faces"):
        mod = types.ModuleType("mesh_intersection." + name)
        setattr(pkg, name, mod)
        sys.modules["mesh_intersection." + name] = mod
    pkg.bvh_search_tree.BVH = M.BVH
    pkg.loss.DistanceFieldPenetrationLoss = M.DistanceFieldPenetrationLoss
    pkg.filter_faces.FilterFaces = M.FilterFaces
    sys.modules["mesh_intersection"] = pkg
    M.BVH.faces = None if faces is None else __import__("numpy").asarray(faces).astype("int64")
    M.BVH.part_filter = part_filter
    return M


def import_reference():
    """Returns a namespace with the reference modules:
    fitting, camera, prior, utils, lbfgs_ls, optim_factory, fit_single_frame, data_parser."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    if REPO_ROOT not in sys.path:
        sys.path.insert(0, REPO_ROOT)
    from oracle.body_model import transform_mat
    for name in ["open3d", "skimage", "skimage.io", "skimage.transform", "trimesh", "pyrender",
                 "cv2", "plyfile", "human_body_prior", "human_body_prior.tools",
                 "human_body_prior.tools.model_loader",
                 "human_body_prior.tools.visualization_tools",
                 "human_body_prior.body_model", "human_body_prior.body_model.body_model",
                 "configargparse", "mesh_viewer"]:
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    if "smplx" not in sys.modules or isinstance(sys.modules["smplx"], MagicMock):
        smplx = types.ModuleType("smplx")
        lbs = types.ModuleType("smplx.lbs")
        lbs.transform_mat = transform_mat
        smplx.lbs = lbs
        sys.modules["smplx"] = smplx
        sys.modules["smplx.lbs"] = lbs
    ref_pkg = os.path.join(REF_ROOT, "smplifyx")
    if ref_pkg not in sys.path:
        sys.path.insert(0, ref_pkg)
    import importlib
    ns = types.SimpleNamespace()
    ns.utils = importlib.import_module("utils")
    ns.prior = importlib.import_module("prior")
    ns.camera = importlib.import_module("camera")
    ns.fitting = importlib.import_module("fitting")
    ns.lbfgs_ls = importlib.import_module("optimizers.lbfgs_ls")
    ns.optim_factory = importlib.import_module("optimizers.optim_factory")
    ns.fit_single_frame = importlib.import_module("fit_single_frame")
    ns.data_parser = importlib.import_module("data_parser")
    return ns
