"""mesh_intersection.bvh_search_tree.BVH(max_collisions) (fit_single_frame.py:301,310)."""
import torch.nn as nn


class BVH(nn.Module):
    def __init__(self, max_collisions=8):
        super().__init__()
        self.max_collisions = int(max_collisions)

    def forward(self, triangles):
        raise RuntimeError("the collision search is part of the fused interpenetration operator (csrc/collide.hip): pass this "
                           "object to create_loss(search_tree=...) or use smplifyx_amd.engine.Penetration on a batch of meshes")
