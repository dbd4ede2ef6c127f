"""mesh_intersection.bvh_search_tree.BVH(max_collisions) (fit_single_frame.py:301,310; called at fitting.py:445-447)."""
import numpy as np
import torch.nn as nn

from . import _operator


class BVH(nn.Module):
    """Inside the fitting loop the collision search is the broad phase of the fused device operator (csrc/collide.hip) and this
    object is read for `max_collisions`.  Called on its own -- `collision_idxs = search_tree(triangles)` -- it runs that broad
    phase on the triangles' mesh and returns the package's tensor: int64 [B, F * max_collisions, 2], one row per colliding pair
    of triangles (boxes overlap, no shared vertex; each unordered pair once, lower id first, sorted), -1 where empty.  A triangle
    with more than max_collisions partners keeps its lowest ids (oracle/penetration.py assumption A1)."""

    def __init__(self, max_collisions=8, faces=None):
        super().__init__()
        self.max_collisions = int(max_collisions)
        if faces is not None:                       # (not an argument of the package: spares the topology recovery)
            _operator.set_faces(faces)

    def forward(self, triangles):
        op, verts = _operator.operator_for(triangles, self.max_collisions)
        op.pen.eval(verts, 1.0)                      # (the broad phase does not depend on the cone height)
        pairs = []
        for b in range(verts.shape[0]):
            p = op.pen.pairs(b)
            pairs.append(p[p[:, 0] < p[:, 1]] if len(p) else np.zeros((0, 2), np.int64))
        return _operator.pair_tensor(pairs, op.F, self.max_collisions, triangles.device)
