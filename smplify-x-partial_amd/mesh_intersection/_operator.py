"""What the three stand-alone modules share: the mesh behind a `triangles` tensor, and the device operator on it.

The reference hands the package `triangles = index_select(vertices, 1, faces).view(B, F, 3, 3)` (fitting.py:440-442): corner
coordinates, no vertex ids.  The device operator (csrc/collide.hip) works on vertices + faces -- triangles that share a vertex
are never a collision --, so the topology is recovered from the first mesh: corners with bit-identical coordinates are one
vertex (they are copies of one row of `vertices`).  It is kept, with the operator built on it, for as long as the tensors
that arrive keep that shape and that corner structure."""
import numpy as np
import torch

from .. import engine

_CACHE = {}


class MeshOperator(object):
    def __init__(self, triangles, max_collisions):
        B, F = triangles.shape[:2]
        flat = triangles[0].reshape(-1, 3)
        _, inverse = torch.unique(flat, dim=0, return_inverse=True)
        V = int(inverse.max().item()) + 1
        rep = torch.zeros([V], dtype=torch.long, device=triangles.device)
        rep.scatter_(0, inverse, torch.arange(3 * F, device=triangles.device))
        self.F, self.V, self.B = F, V, B
        self.inverse, self.rep = inverse, rep
        self.faces = inverse.view(F, 3)
        self.max_collisions = int(max_collisions)
        self.pen = engine.Penetration(V, self.faces.cpu().numpy(), max_collisions=self.max_collisions, max_batch=B)

    def verts(self, triangles):
        """[B, V, 3] float32 vertices of a `triangles` tensor of this topology (checked)."""
        flat = triangles.reshape(triangles.shape[0], -1, 3)
        v = flat[:, self.rep].to(torch.float32).contiguous()
        if not torch.equal(v[:, self.inverse], flat.to(torch.float32)):
            raise ValueError("the corners of these triangles do not coincide the way the first mesh's did: another topology")
        return v


def operator_for(triangles, max_collisions):
    if not (torch.is_tensor(triangles) and triangles.is_cuda and triangles.dim() == 4 and triangles.shape[2:] == (3, 3)):
        raise TypeError("triangles: CUDA tensor [B, F, 3, 3] (the HIP operator has no CPU fallback)")
    key = (int(triangles.shape[0]), int(triangles.shape[1]), int(max_collisions), triangles.device.index)
    op = _CACHE.get(key)
    if op is not None:
        try:
            return op, op.verts(triangles)
        except ValueError:
            op.pen.close()
    op = MeshOperator(triangles.detach(), max_collisions)
    _CACHE[key] = op
    return op, op.verts(triangles)


def pair_tensor(pairs_per_mesh, F, max_collisions, device):
    """list of [n_b, 2] unordered pairs -> the package's collision tensor [B, F * max_collisions, 2] int64, -1 where empty."""
    out = torch.full([len(pairs_per_mesh), F * max_collisions, 2], -1, dtype=torch.long)
    for b, p in enumerate(pairs_per_mesh):
        n = min(len(p), out.shape[1])
        if n:
            out[b, :n] = torch.as_tensor(np.asarray(p[:n], np.int64))
    return out.to(device)
