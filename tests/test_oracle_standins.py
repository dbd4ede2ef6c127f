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


@pytest.mark.parametrize("tag", ["w0", "w01", "nopairs"])
def test_objective_with_the_term_matches_the_reference_lines(tag):
    """tests/golden/objective_pen.npz: the REAL fitting.SMPLifyLoss.forward with interpenetration=True (fitting.py:437-455 over the
    stand-ins) -- the oracle's objective + oracle.fit_frame.FrameFit.penetration_term on the same inputs gives the same total and
    the same gradients, with the collision weight at 0 (gate closed: the term is not evaluated, `bvh_calls` 0 in the golden), at 0.1,
    and when the part filter leaves no pair (branch not taken: `pen_calls` 0)."""
    import os
    import types
    from collections import namedtuple
    from oracle import objective as obj
    from oracle.fit_frame import FrameFit
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "objective_pen.npz"))
    assert int(g["w0_bvh_calls"]) == 0 and int(g["w01_bvh_calls"]) == 1 and int(g["w01_pen_calls"]) == 1
    assert int(g["nopairs_bvh_calls"]) == 1 and int(g["nopairs_pen_calls"]) == 0
    T = lambda k: torch.tensor(g[k], dtype=torch.float64, requires_grad=True)
    joints, full_pose, betas, emb = T("joints"), T("full_pose"), T("betas"), T("emb")
    vertices = torch.tensor(g["verts"][None], dtype=torch.float64, requires_grad=True)
    cam_t = torch.tensor([[0.05, 0.1, 20.0]], dtype=torch.float64, requires_grad=True)
    f = torch.full([1], 5000.0, dtype=torch.float64)
    proj = obj.project(joints, torch.eye(3, dtype=torch.float64)[None], cam_t, f, f, torch.tensor([[400.0, 300.0]], dtype=torch.float64))
    MO = namedtuple("MO", ["full_pose", "betas", "body_pose", "left_hand_pose", "right_hand_pose", "expression", "jaw_pose"])
    cw = float(g[tag + "_coll_loss_weight"])
    w = {k: torch.tensor(v, dtype=torch.float64) for k, v in dict(
        data_weight=1000 / 600, body_pose_weight=300.0, shape_weight=50.0, bending_prior_weight=3.17 * 300.0, coll_loss_weight=cw).items()}
    terms = obj.smplify_terms(MO(full_pose, betas, emb, None, None, None, None), proj, torch.tensor(g["gt"]), torch.tensor(g["conf"]),
                              torch.tensor(g["jw"]), w, emb, use_vposer=False, regression_pose=torch.tensor(g["reg"]), stage=1,
                              num_stages=3, use_joints_conf=True, use_hands=False, use_face=False, rho=100)
    ign = None if tag == "nopairs" else ["0,1"]
    fake = types.SimpleNamespace(pen=dict(faces=g["faces"].astype(np.int64), segm=g[tag + "_segm"], parents=g["parents"], ign=ign),
                                 cfg=dict(max_collisions=128, df_cone_height=0.01, penalize_outside=True, point2plane=False))
    pen = FrameFit.penetration_term(fake, vertices, w)
    total = terms["total"] if pen is None else terms["total"] + pen
    assert (pen is None) == (tag == "w0")
    ref_total = float(g[tag + "_total"])
    assert abs(total.item() - ref_total) <= 1e-12 * abs(ref_total)
    if tag == "w01":      # the term itself, not only the total it is 1e-5 of
        ref_pen = ref_total - float(g["w0_total"])
        assert ref_pen > 100 and abs(float(pen) - ref_pen) <= 1e-8 * ref_pen
    total.backward()
    for name, t in (("joints", joints), ("full_pose", full_pose), ("betas", betas), ("emb", emb), ("cam_t", cam_t), ("vertices", vertices)):
        got = t.grad.numpy() if t.grad is not None else np.zeros(tuple(t.shape))
        refg = g[tag + "_d_" + name]
        assert np.allclose(got, refg, rtol=1e-9, atol=1e-12 * max(1.0, np.abs(refg).max())), (tag, name)
    assert (np.abs(g[tag + "_d_vertices"]).sum() > 0) == (tag == "w01")
