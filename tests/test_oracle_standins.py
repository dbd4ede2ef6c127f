"""CPU: the pieces round 6 added to the oracle for the reference-driven fits WITH the interpenetration term
(tools/make_goldens.py e2e_pen_set): the fast broad phase against the plain one, and the CPU stand-ins for the three
mesh_intersection objects (oracle/mesh_intersection_cpu.py) run through the reference's own lines fitting.py:440-455 against
oracle.penetration / oracle.fit_frame.FrameFit.penetration_term."""
import numpy as np
import pytest
import torch

from oracle import mesh_intersection_cpu as M
from oracle import penetration as OP


def _soup(seed, F=600, spread=1.0, size=0.08):
    """F loose triangles (their own vertices) + a strip of triangles that SHARE vertices, three parts."""
    r = np.random.RandomState(seed)
    c = spread * r.rand(F, 3)
    tri = c[:, None, :] + size * r.randn(F, 3, 3)
    verts = tri.reshape(-1, 3)
    faces = np.arange(F * 3).reshape(F, 3)
    strip = np.stack([np.arange(0, 40), np.arange(1, 41), np.arange(2, 42)], 1) + 7          # neighbours share two vertices
    faces = np.concatenate([faces, strip])
    segm = r.randint(0, 3, len(faces))
    parents = np.where(segm == 2, 1, -1)
    return verts, faces, segm, parents


@pytest.mark.parametrize("seed,spread", [(0, 1.0), (1, 0.3), (2, 3.0)])
def test_sweep_broad_phase_equals_the_plain_one(seed, spread):
    verts, faces, segm, parents = _soup(seed, spread=spread)
    for args in ((None, None, None), (segm, parents, None), (segm, parents, ["0,1"])):
        a = OP.candidate_pairs(verts, faces, *args)
        b = OP.candidate_pairs_sweep(verts, faces, *args)
        c = OP.candidate_pairs(verts, faces, *args, sweep=False)
        assert len(a) > 50 and np.array_equal(a, b) and np.array_equal(a, c)
    assert len(OP.candidate_pairs_sweep(verts[:3], faces[:1])) == 0


def test_stand_ins_through_the_reference_lines_equal_the_oracle_term():
    """fitting.py:440-455 literally -- triangles = index_select(vertices, 1, faces); search_tree; tri_filtering_module; the
    `collision_idxs.ge(0).sum() > 0` branch; torch.sum(coll_loss_weight * pen_distance(...)) -- on the stand-ins, against
    oracle.penetration on the same mesh: pair set, loss, and the gradient autograd carries back to the vertices."""
    verts, faces, segm, parents = _soup(3, F=400, spread=0.9)
    ign = ["0,1"]
    sigma, mc = 0.01, 128
    search_tree = M.BVH(max_collisions=mc, faces=faces)
    pen_distance = M.DistanceFieldPenetrationLoss(sigma=sigma, point2plane=False, vectorized=True, penalize_outside=True)
    tri_filtering_module = M.FilterFaces(faces_segm=segm, faces_parents=parents, ign_part_pairs=ign)
    vertices = torch.tensor(verts[None], dtype=torch.float64, requires_grad=True)
    body_model_faces = torch.tensor(faces.reshape(-1))
    coll_loss_weight = torch.tensor(0.1, dtype=torch.float64)
    batch_size = 1
    triangles = torch.index_select(vertices, 1, body_model_faces).view(batch_size, -1, 3, 3)
    with torch.no_grad():
        collision_idxs = search_tree(triangles)
    assert collision_idxs.shape == (1, len(faces) * mc, 2) and collision_idxs.dtype == torch.int64
    n_raw = int(collision_idxs.ge(0).all(-1).sum())
    collision_idxs = tri_filtering_module(collision_idxs)
    assert collision_idxs.ge(0).sum().item() > 0
    pen_loss = torch.sum(coll_loss_weight * pen_distance(triangles, collision_idxs))
    pen_loss.backward()
    pairs = OP.candidate_pairs(verts, faces, segm, parents, ign)
    kept = collision_idxs[0][collision_idxs[0].ge(0).all(-1)].numpy()
    raw = OP.candidate_pairs(verts, faces)
    assert np.bincount(raw.reshape(-1)).max() < mc            # (the cap does not bind: the package's order of operations and A1's agree)
    assert n_raw == len(raw) > len(pairs) > 100 and np.array_equal(kept, pairs)
    lo, go, _ = OP.penetration(verts, faces, segm, parents, ign, sigma=sigma)
    assert abs(float(pen_loss) - 0.1 * lo) <= 1e-12 * lo
    assert np.abs(vertices.grad[0].numpy() - 0.1 * go).max() <= 1e-10 * np.abs(go).max()
    # nothing left after the filter: the reference's branch skips the loss (pen_distance of an empty list is 0 as well)
    none = torch.full_like(collision_idxs, -1)
    assert float(pen_distance(triangles, none).sum()) == 0.0


def test_bvh_stand_in_caps_like_the_oracle_fit():
    """A binding cap: BVH(max_collisions=4) with the part rules inside the broad phase (assumption A1 as oracle/fit_frame.py and
    the device have it) returns exactly oracle.penetration.ordered_pairs_capped's symmetric set, and counts what it cut."""
    verts, faces, segm, parents = _soup(5, F=300, spread=0.25)
    bvh = M.BVH(max_collisions=4, faces=faces, part_filter=(segm, parents, None))
    tri = torch.tensor(verts[faces][None])
    idx = bvh(tri)[0]
    kept = idx[idx.ge(0).all(-1)].numpy()
    pairs = OP.candidate_pairs(verts, faces, segm, parents, None)
    assert np.bincount(pairs.reshape(-1)).max() > 4
    op, cut = OP.ordered_pairs_capped(pairs, 4)
    assert cut > 0 and bvh.pairs_cut == cut and np.array_equal(kept, op[op[:, 0] < op[:, 1]])
    # the loss on that tensor = the oracle's ordered form on the capped set
    loss = M.DistanceFieldPenetrationLoss(sigma=0.01)(tri, idx[None])[0]
    ref = OP.penetration_loss_ordered(torch.tensor(verts), faces, op, 0.01)
    assert abs(float(loss) - float(ref)) <= 1e-12 * float(ref)
