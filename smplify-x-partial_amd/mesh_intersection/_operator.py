"""What the three stand-alone modules share: the mesh behind a `triangles` tensor, and the device operator on it.

The reference hands the package `triangles = index_select(vertices, 1, faces).view(B, F, 3, 3)` (fitting.py:440-442): corner
coordinates, no vertex ids.  The device operator (csrc/collide.hip) works on vertices + faces -- triangles that share a vertex
are never a collision --, so the topology is recovered from the first mesh: corners with bit-identical coordinates are one
vertex (they are copies of one row of `vertices`).  A caller that has the faces can hand them over instead
(`set_faces(faces)`; `BVH(..., faces=...)`): nothing is guessed then, and a first mesh with coincident vertices cannot merge them.

ONE operator per (topology, batch size, device) serves BVH and DistanceFieldPenetrationLoss alike, whatever max_collisions the
BVH was built with (the loss evaluates the pairs it is given; the cap only sizes buffers).  The cache holds the most recently
used operator per device and closes the one it evicts."""
import numpy as np
import torch

from .. import engine

_CACHE = {}             # device index -> MeshOperator (at most one per device)
_FACES = {}             # F -> faces [F, 3] supplied by the caller (set_faces)


def set_faces(faces):
    """Tell the stand-alone modules the index array behind the `triangles` tensors they will see ([F, 3]; None forgets it)."""
    if faces is None:
        _FACES.clear()
        return
    f = np.ascontiguousarray(np.asarray(faces.detach().cpu() if torch.is_tensor(faces) else faces).reshape(-1, 3), np.int64)
    _FACES[int(f.shape[0])] = f


class MeshOperator(object):
    def __init__(self, triangles, max_collisions):
        B, F = triangles.shape[:2]
        flat = triangles[0].reshape(-1, 3)
        given = _FACES.get(int(F))
        if given is not None:
            inverse = torch.as_tensor(given.reshape(-1), device=triangles.device)
            V = int(inverse.max().item()) + 1
            self.from_faces = True
        else:
            _, inverse = torch.unique(flat, dim=0, return_inverse=True)
            V = int(inverse.max().item()) + 1
            self.from_faces = False
            # a closed triangle mesh has about F / 2 vertices; far fewer means distinct vertices coincide in this first mesh
            # (a degenerate or zeroed body) and would be merged: adjacent triangles would count as "sharing a vertex"
            if F >= 64 and V < F // 4:
                raise ValueError("the first mesh has %d distinct corner positions for %d triangles: vertices coincide, the "
                                 "topology cannot be recovered from it (hand the faces over: mesh_intersection.set_faces)" % (V, F))
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

    def close(self):
        self.pen.close()


def operator_for(triangles, max_collisions=None):
    """(operator, vertices) for a triangles tensor.  max_collisions=None: whatever operator this topology already has (the loss
    does not care); a number: an operator whose partner lists hold that many (BVH) -- a smaller one is rebuilt."""
    if not (torch.is_tensor(triangles) and triangles.is_cuda and triangles.dim() == 4 and triangles.shape[2:] == (3, 3)):
        raise TypeError("triangles: CUDA tensor [B, F, 3, 3] (the HIP operator has no CPU fallback)")
    dev = triangles.device.index
    B, F = int(triangles.shape[0]), int(triangles.shape[1])
    op = _CACHE.get(dev)
    if op is not None and op.B == B and op.F == F and (max_collisions is None or op.max_collisions >= int(max_collisions)):
        try:
            return op, op.verts(triangles)
        except ValueError:
            pass
    if op is not None:
        op.close()
        del _CACHE[dev]
    op = MeshOperator(triangles.detach(), 128 if max_collisions is None else int(max_collisions))
    _CACHE[dev] = op
    return op, op.verts(triangles)


def pair_tensor(pairs_per_mesh, F, max_collisions, device):
    """list of [n_b, 2] unordered pairs -> the package's collision tensor [B, F * max_collisions, 2] int64, -1 where empty."""
    out = torch.full([len(pairs_per_mesh), F * max_collisions, 2], -1, dtype=torch.long)
    for b, p in enumerate(pairs_per_mesh):
        n = min(len(p), out.shape[1])
        if n:
            out[b, :n] = torch.as_tensor(np.asarray(p[:n], np.int64))
    return out.to(device)
