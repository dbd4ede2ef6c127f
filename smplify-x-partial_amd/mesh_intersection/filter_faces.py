"""mesh_intersection.filter_faces.FilterFaces(faces_segm, faces_parents, ign_part_pairs) (fit_single_frame.py:303,325-328;
called at fitting.py:449-450): pairs of triangles of the same part, of parent / child parts, or of a listed pair of parts are
not collisions."""
import numpy as np
import torch
import torch.nn as nn


class FilterFaces(nn.Module):
    def __init__(self, faces_segm=None, faces_parents=None, ign_part_pairs=None):
        super().__init__()
        self.faces_segm = np.asarray(faces_segm).astype(np.int64)
        self.faces_parents = np.asarray(faces_parents).astype(np.int64)
        self.ign_part_pairs = list(ign_part_pairs) if ign_part_pairs is not None else []
        pairs = [tuple(int(x) for x in str(p).split(",")) if isinstance(p, str) else (int(p[0]), int(p[1])) for p in self.ign_part_pairs]
        self.register_buffer("_segm", torch.as_tensor(self.faces_segm))
        self.register_buffer("_parents", torch.as_tensor(self.faces_parents))
        self.register_buffer("_ign", torch.as_tensor(np.asarray(pairs, np.int64).reshape(-1, 2)))

    def forward(self, collision_idxs):
        """collision tensor [B, N, 2] (-1 = empty) -> the same tensor with the rows of non-colliding part pairs set to -1."""
        idx = collision_idxs
        segm, parents, ign = self._segm.to(idx.device), self._parents.to(idx.device), self._ign.to(idx.device)
        valid = (idx >= 0).all(-1)
        a, b = idx[..., 0].clamp(min=0), idx[..., 1].clamp(min=0)
        sa, sb, pa, pb = segm[a], segm[b], parents[a], parents[b]
        drop = (sa == sb) | (sa == pb) | (sb == pa)
        for k in range(ign.shape[0]):
            drop |= ((sa == ign[k, 0]) & (sb == ign[k, 1])) | ((sa == ign[k, 1]) & (sb == ign[k, 0]))
        keep = valid & ~drop
        return torch.where(keep[..., None], idx, torch.full_like(idx, -1))
