#!/usr/bin/env python
"""Build tests/golden/smplx_topology.npz LOCALLY from DATA files of the reference tree (no reference code is imported):
the mesh the reference's interpenetration term is evaluated on (fitting.py:437-455, fit_single_frame.py:300-328), as far as the
tree holds it --

  faces   [20908, 3]  the SMPL-X face topology (demo/ExPose_results/*/*.ply; the four files agree)
  segm, parents [20908]  per-face body part and parent part of smplifyx/smplx_parts_segm.pkl (loaded at fit_single_frame.py:317-324)
  vertices [10475, 3], joints [144, 3]   ExPose's posed body of demo frame 02 (demo/ExPose_results/02_cropped.jpg/*_params.npz)

LICENCE: these arrays derive from SMPL-X / ExPose data under the MPG non-commercial research licence (the reference's LICENSE).
The file is therefore NOT committed (.gitignore) -- ADVICE round 5.  It is a build product like libsfx.so: __graft_entry__.build()
makes it whenever the reference tree is present (SFX_REFERENCE_ROOT, default /root/reference) and it then travels to the GPU box
with the working tree.  Where it is absent the tests on the real surface skip and `bench.py --workload pen` falls back to the
synthetic tube mesh (`--mesh tubes`), saying so.

usage: python tools/make_topology.py [--force]"""
import glob
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ROOT = os.environ.get("SFX_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "smplx_topology.npz")


def read_ply_mesh(path):
    """(vertices [V,3] float64, faces [F,3] int64) of a binary-little-endian .ply with double vertices and uchar/uint
    face lists (what Open3D wrote into demo/ExPose_results)."""
    raw = open(path, "rb").read()
    i = raw.index(b"end_header\n") + len(b"end_header\n")
    hdr = raw[:i].decode().splitlines()
    if "format binary_little_endian 1.0" not in hdr or "property double x" not in hdr:
        raise ValueError("unexpected .ply header: %s" % hdr)
    nv = int([l for l in hdr if l.startswith("element vertex")][0].split()[-1])
    nf = int([l for l in hdr if l.startswith("element face")][0].split()[-1])
    v = np.frombuffer(raw, "<f8", nv * 3, i).reshape(nv, 3)
    f = np.frombuffer(raw, np.dtype([("n", "u1"), ("idx", "<u4", 3)]), nf, i + nv * 24)
    if not (f["n"] == 3).all():
        raise ValueError("non-triangle faces")
    return v.copy(), f["idx"].astype(np.int64)


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "demo", "ExPose_results")) and \
        os.path.isfile(os.path.join(REF_ROOT, "smplifyx", "smplx_parts_segm.pkl"))


def build(force=False, out=OUT):
    """Write the fixture; returns its path, or None when the reference tree is not there.  Deterministic: the same bytes of
    array data every time (test_topology_model.py checks the arrays' shapes and invariants)."""
    if os.path.exists(out) and not force:
        return out
    if not available():
        return None
    demo = os.path.join(REF_ROOT, "demo", "ExPose_results")
    meshes = [read_ply_mesh(p) for p in sorted(glob.glob(os.path.join(demo, "*", "*.ply")))]
    if len(meshes) != 4 or not all(np.array_equal(m[1], meshes[0][1]) for m in meshes):
        raise RuntimeError("the demo meshes do not share one topology")
    faces = meshes[0][1]
    with open(os.path.join(REF_ROOT, "smplifyx", "smplx_parts_segm.pkl"), "rb") as fh:
        parts = pickle.load(fh, encoding="latin1")
    segm, parents = np.asarray(parts["segm"], np.int64), np.asarray(parts["parents"], np.int64)
    if not (segm.shape == parents.shape == (len(faces),)):
        raise RuntimeError("part table does not match the faces")
    z = np.load(os.path.join(demo, "02_cropped.jpg", "02_cropped.jpg_params.npz"), allow_pickle=True)
    verts, joints = np.asarray(z["vertices"], np.float32), np.asarray(z["joints"], np.float32)
    if np.abs(meshes[0][0] - (verts.astype(np.float64) + np.asarray(z["transl"]))).max() >= 1e-6:     # the .ply = vertices + transl
        raise RuntimeError("demo .ply and params disagree")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, faces=faces.astype(np.int32), segm=segm.astype(np.int8), parents=parents.astype(np.int8),
                        vertices=verts, joints=joints)
    return out


if __name__ == "__main__":
    p = build(force="--force" in sys.argv)
    print("wrote %s" % p if p else "reference tree not found at %s: nothing written" % REF_ROOT)
