"""ORACLE (test infrastructure): CPU stand-ins for the three objects of the external CUDA package `mesh_intersection`
that the reference builds at fit_single_frame.py:300-328 and calls at fitting.py:437-455 --

    search_tree = BVH(max_collisions)                           collision_idxs = search_tree(triangles)
    filter_faces = FilterFaces(faces_segm, faces_parents, ign_part_pairs)   collision_idxs = filter_faces(collision_idxs)
    pen_distance = DistanceFieldPenetrationLoss(sigma, point2plane, vectorized, penalize_outside)
                                                                pen_distance(triangles, collision_idxs) -> [B]

-- built on oracle/penetration.py (PARITY UNPINNED: the package is absent from /root/reference and this image; the named
assumptions A1-A7 of that file apply).  tools/ref_import.py registers this module under the package's three module names so that
the REAL reference code (fitting.SMPLifyLoss.forward, FittingMonitor.run_fitting, lbfgs_ls.LBFGS, fit_single_frame) runs WITH
the term on the CPU: the gating `coll_loss_weight.item() > 0`, the `collision_idxs.ge(0).sum() > 0` branch and
`torch.sum(coll_loss_weight * pen_distance(...))` are then the reference's own lines (tests/golden/e2e_pen_set.npz).

Tensor contract: triangles [B, F, 3, 3]; collision_idxs int64 [B, F * max_collisions, 2], one row per colliding
pair of triangles (each unordered pair once, lower id first), -1 where empty.

Never imported by the product (smplify-x-partial_amd has its own device modules of the same names)."""
import numpy as np
import torch
import torch.nn as nn

from . import penetration as P


def _mesh_of(triangles, faces):
    """vertices [V, 3] (float64 numpy) of one mesh's triangles [F, 3, 3] given the faces that produced them."""
    tri = triangles.detach().cpu().numpy().astype(np.float64)
    V = int(faces.max()) + 1
    v = np.zeros((V, 3))
    v[faces.reshape(-1)] = tri.reshape(-1, 3)
    return v


class BVH(nn.Module):
    """Broad phase.  `faces` (class attribute or constructor argument) is the index array behind the triangles tensor -- the
    package compares corner coordinates to find shared vertices; the stand-in is told.  `part_filter` = (segm, parents,
    ign_part_pairs), optional: assumption A1 as oracle/fit_frame.py and csrc/collide.hip have it -- the part rules act inside
    the broad phase, the cap of max_collisions partners per triangle (lowest ids, kept by both sides) on the lists that are left
    -- so that a binding cap cuts the same lists everywhere.  Without it the cap acts on the unfiltered lists (the package's
    order of operations); the two agree whenever no triangle has more than max_collisions partners."""
    faces = None
    part_filter = None
    fast = True

    def __init__(self, max_collisions=8, faces=None, part_filter=None):
        super().__init__()
        self.max_collisions = int(max_collisions)
        if faces is not None:
            self.faces = np.asarray(faces, np.int64)
        if part_filter is not None:
            self.part_filter = part_filter
        self.pairs_cut = 0            # ordered pairs removed by the cap, summed over the calls
        self.calls = 0
        self.max_pairs = 0

    def forward(self, triangles):
        if self.faces is None:
            raise RuntimeError("BVH stand-in: set BVH.faces to the model's faces first")
        faces = np.asarray(self.faces, np.int64)
        B, F = triangles.shape[:2]
        assert F == faces.shape[0]
        out = torch.full([B, F * self.max_collisions, 2], -1, dtype=torch.long)
        pf = self.part_filter or (None, None, None)
        find = P.candidate_pairs_sweep if self.fast else P.candidate_pairs
        for b in range(B):
            v = _mesh_of(triangles[b], faces)
            if not np.isfinite(v).all():          # a diverged fit: NaN boxes overlap nothing
                continue
            pairs = find(v, faces, pf[0], pf[1], pf[2])
            if len(pairs) and np.bincount(pairs.reshape(-1)).max() > self.max_collisions:
                op, cut = P.ordered_pairs_capped(pairs, self.max_collisions)
                self.pairs_cut += cut
                pairs = op[op[:, 0] < op[:, 1]]
            n = min(len(pairs), out.shape[1])
            if n:
                out[b, :n] = torch.as_tensor(pairs[:n])
            self.max_pairs = max(self.max_pairs, len(pairs))
        self.calls += 1
        return out


class FilterFaces(nn.Module):
    """Rows whose two triangles belong to the same part, to parent / child parts or to a listed pair of parts become -1."""

    def __init__(self, faces_segm=None, faces_parents=None, ign_part_pairs=None):
        super().__init__()
        self.segm = torch.as_tensor(np.asarray(faces_segm).astype(np.int64))
        self.parents = torch.as_tensor(np.asarray(faces_parents).astype(np.int64))
        self.ign = P.parse_ign_part_pairs(ign_part_pairs)

    def forward(self, collision_idxs):
        idx = collision_idxs
        valid = (idx >= 0).all(-1)
        a, b = idx[..., 0].clamp(min=0), idx[..., 1].clamp(min=0)
        sa, sb, pa, pb = self.segm[a], self.segm[b], self.parents[a], self.parents[b]
        drop = (sa == sb) | (sa == pb) | (sb == pa)
        for lo, hi in self.ign:
            drop |= ((sa == lo) & (sb == hi)) | ((sa == hi) & (sb == lo))
        return torch.where((valid & ~drop)[..., None], idx, torch.full_like(idx, -1))


class DistanceFieldPenetrationLoss(nn.Module):
    """sum over the listed pairs (f, g) of  sum_{v in g} Psi_f(v)^2 + sum_{v in f} Psi_g(v)^2  (oracle/penetration.py), per mesh:
    [B], differentiable with respect to `triangles` (autograd).  `linear_max` is accepted and not applied (assumption A4)."""

    def __init__(self, sigma=0.5, point2plane=False, vectorized=True, penalize_outside=True, linear_max=1000):
        super().__init__()
        self.sigma, self.point2plane, self.vectorized = float(sigma), bool(point2plane), vectorized
        self.penalize_outside, self.linear_max = bool(penalize_outside), linear_max
        self.calls = 0

    def forward(self, triangles, collision_idxs):
        B = triangles.shape[0]
        out = []
        for b in range(B):
            idx = collision_idxs[b]
            idx = idx[(idx >= 0).all(-1)]
            if idx.shape[0] == 0:
                out.append(triangles[b].sum() * 0.0)
                continue
            A, Bt = triangles[b][idx[:, 0]], triangles[b][idx[:, 1]]
            oa, ra, na = P._cone_geometry(A)
            ob, rb, nb = P._cone_geometry(Bt)
            pa = (P._psi(oa, ra, na, Bt, self.sigma, self.penalize_outside) ** 2).sum(1)
            pb = (P._psi(ob, rb, nb, A, self.sigma, self.penalize_outside) ** 2).sum(1)
            if self.point2plane:
                c = ((na * nb).sum(1)) ** 2
                out.append((c * (pa + pb)).sum())
            else:
                out.append(pa.sum() + pb.sum())
        self.calls += 1
        return torch.stack(out)
