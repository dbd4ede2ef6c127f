"""Drop-in for the three objects of the external `mesh_intersection` package the reference builds in
fit_single_frame.py:300-328 and hands to create_loss (search_tree, pen_distance, tri_filtering_module).

In this engine the interpenetration term is ONE fused device operator (csrc/collide.hip: broad phase, part filter, cone
distance field and its gradient), so these classes are parameter holders: the fitting closure reads
max_collisions, sigma, penalize_outside and the part labels from them and switches the term on for the stages whose
coll_loss_weight is positive -- the call sequence of the reference runs unmodified.  Stand-alone evaluation on a batch of
meshes: smplifyx_amd.engine.Penetration."""
from . import bvh_search_tree, filter_faces, loss  # noqa: F401
