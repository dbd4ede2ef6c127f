#!/usr/bin/env python
"""Table of tests/golden/e2e_pen_set.npz for LAB_NOTES.md §R6.1: per frame and run of the REAL reference (fp32 / fp64 with the
interpenetration term, fp32 without) the per-stage losses of the kept orientation, closure evaluations, whether a folded mesh was
met (largest unordered pair count of any evaluation; ordered pairs the cap of 128 partners cut) and whether the fitted parameters
are finite.  With gpurun_out/pen_collapse_probe.npz (tools/pen_collapse_probe.py on the GPU box) the device's fit of the same
frames is printed beside it.   usage: python tools/pen_set_report.py [--md]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_pen_set.npz"))
probe = os.path.join(ROOT, "gpurun_out", "pen_collapse_probe.npz")
d = np.load(probe) if os.path.exists(probe) else None
frames = [int(i) for i in g["frames"]]
fmt = lambda a: " ".join("%.0f" % x if np.isfinite(x) else "nan" for x in a)
print("| frame | run | orient. | stage losses (camera, 3 body stages; kept orientation) | evals | max pairs | cut | finite |")
print("|---|---|---|---|---|---|---|---|")
for i in frames:
    for tag, name in (("f32", "ref fp32"), ("f64", "ref fp64"), ("f32_noterm", "ref fp32, no term")):
        k = "f%d_%s_" % (i, tag)
        n_or = 2 if len(g[k + "losses_all"]) == 7 else 1
        print("| %d | %s | %d of %d | %s | %d | %d | %d | %s |" % (i, name, int(g[k + "kept_orientation"]) + 1, n_or, fmt(g[k + "losses"]), int(g[k + "evals_all"].sum()),
              int(g[k + "bvh_max_pairs"]), int(g[k + "bvh_pairs_cut"]), "yes" if bool(g[k + "finite"]) else "NO"))
    if d is not None and i < len(d["stage_loss"]):
        print("| %d | device fp32 | of %d | %s | %d | | | %s (cut-walk flag %d) |" % (i, int(d["n_orient"][i]), fmt(d["stage_loss"][i]), int(d["stage_evals"][i].sum()),
              "yes" if np.isfinite(d["stage_loss"][i]).all() else "NO", int(d["flag"][i])))
folded = lambda tag: [i for i in frames if int(g["f%d_%s_bvh_pairs_cut" % (i, tag)]) > 0]
print()
print("reference frames that met a folded mesh (cap binding): fp32 %s, fp64 %s" % (folded("f32"), folded("f64")))
print("reference frames with non-finite fitted parameters: fp32 %s, fp64 %s" % ([i for i in frames if not bool(g["f%d_f32_finite" % i])], [i for i in frames if not bool(g["f%d_f64_finite" % i])]))
