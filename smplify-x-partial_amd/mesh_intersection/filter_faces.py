"""mesh_intersection.filter_faces.FilterFaces(faces_segm, faces_parents, ign_part_pairs) (fit_single_frame.py:303,325-328):
pairs of triangles of the same part, of parent / child parts, or of a listed pair of parts are not collisions."""
import numpy as np
import torch.nn as nn


class FilterFaces(nn.Module):
    def __init__(self, faces_segm=None, faces_parents=None, ign_part_pairs=None):
        super().__init__()
        self.faces_segm = np.asarray(faces_segm).astype(np.int64)
        self.faces_parents = np.asarray(faces_parents).astype(np.int64)
        self.ign_part_pairs = list(ign_part_pairs) if ign_part_pairs is not None else []

    def forward(self, collision_idxs):
        raise RuntimeError("the part filter is part of the fused interpenetration operator (csrc/collide.hip): pass this "
                           "object to create_loss(tri_filtering_module=...)")
