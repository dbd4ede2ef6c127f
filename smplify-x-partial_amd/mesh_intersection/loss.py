"""mesh_intersection.loss.DistanceFieldPenetrationLoss(sigma, point2plane, vectorized, penalize_outside)
(fit_single_frame.py:302,312-315; called at fitting.py:451-455)."""
import torch
import torch.nn as nn

from . import _operator


class _ConeFieldLoss(torch.autograd.Function):
    """loss [B] of the cone distance field on the given pairs; backward: the gradient with respect to every triangle corner,
    both computed by the device operator (sfx_pen_eval_pairs: k_pen_narrow)."""

    @staticmethod
    def forward(ctx, triangles, collision_idxs, module):
        # the operator BVH built for this topology if its partner lists hold what the tensor can carry: [B, F * max_collisions, 2]
        # was made by a BVH(max_collisions), so no triangle has more partners than shape[1] // F
        op, verts = _operator.operator_for(triangles, max(1, min(1024, int(collision_idxs.shape[1]) // max(1, int(triangles.shape[1])))))
        idx = collision_idxs.to(triangles.device)
        B, F = triangles.shape[0], triangles.shape[1]
        # each unordered pair once, whatever the caller's tensor holds (both orders, repeats): canonical keys, unique per mesh
        valid = (idx >= 0).all(-1)
        lo, hi = torch.minimum(idx[..., 0], idx[..., 1]), torch.maximum(idx[..., 0], idx[..., 1])
        key = torch.where(valid & (lo != hi), lo * F + hi, torch.full_like(lo, -1))
        rows = []
        for b in range(B):
            k = torch.unique(key[b][key[b] >= 0])
            rows.append(torch.stack([k // F, k % F], 1))
        n = max(1, max(r.shape[0] for r in rows))
        pairs = torch.full([B, n, 2], -1, dtype=torch.int32, device=triangles.device)
        for b, r in enumerate(rows):
            pairs[b, :r.shape[0]] = r.to(torch.int32)
        loss, _, dtri = op.pen.eval_pairs(verts, pairs, module.sigma, module.penalize_outside, module.point2plane)
        ctx.save_for_backward(dtri)
        return loss.to(triangles.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        (dtri,) = ctx.saved_tensors
        return (dtri * grad_out.reshape(-1, 1, 1, 1).to(dtri.dtype)).to(grad_out.dtype), None, None


class DistanceFieldPenetrationLoss(nn.Module):
    """Inside the fitting loop the distance field is part of the fused device operator and this object is read for its
    parameters.  Called on its own -- `pen_distance(triangles, collision_idxs)` -- it evaluates the term on the pairs it is given
    (the -1 rows of the package's tensor are skipped), loss [B], differentiable with respect to `triangles`.
    `linear_max` is accepted and not applied (oracle/penetration.py assumption A4: believed to cap the linear branch of the
    field for points deeper than linear_max x sigma below a triangle's plane; the reference never passes it,
    fit_single_frame.py:311-314) -- a value other than the package's default 1000 is answered with a warning."""

    def __init__(self, sigma=0.5, point2plane=False, vectorized=True, penalize_outside=True, linear_max=1000):
        super().__init__()
        if linear_max != 1000:
            import warnings
            warnings.warn("DistanceFieldPenetrationLoss(linear_max=%r): the cap is not applied here (oracle/penetration.py "
                          "assumption A4); penetrations deeper than linear_max x sigma = %g m are penalised without it"
                          % (linear_max, float(linear_max) * float(sigma)), RuntimeWarning)
        self.sigma, self.point2plane, self.vectorized = float(sigma), bool(point2plane), vectorized
        self.penalize_outside, self.linear_max = bool(penalize_outside), linear_max

    def forward(self, triangles, collision_idxs):
        return _ConeFieldLoss.apply(triangles, collision_idxs, self)
