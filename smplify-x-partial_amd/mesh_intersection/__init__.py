"""Drop-in for the three objects of the external `mesh_intersection` package the reference builds in
fit_single_frame.py:300-328 and hands to create_loss (search_tree, pen_distance, tri_filtering_module).

Inside the fitting loop the interpenetration term is ONE fused device operator (csrc/collide.hip: broad phase, part filter,
cone distance field and its gradient; LAB_NOTES.md §4.6): the fitting closure reads max_collisions, sigma, penalize_outside and the
part labels from these objects and switches the term on for the stages whose coll_loss_weight is positive.  Called on their own
they do what the package's modules do, on the same device operator, so that fitting.py:440-455 runs literally
(tests/test_gpu_topology.py::test_the_reference_lines_run_on_the_stand_alone_modules): BVH(triangles) -> collision tensor,
FilterFaces(collision_idxs) -> filtered tensor, DistanceFieldPenetrationLoss(triangles, collision_idxs) -> loss [B],
differentiable with respect to the triangles."""
from . import bvh_search_tree, filter_faces, loss  # noqa: F401
from ._operator import set_faces  # noqa: F401
