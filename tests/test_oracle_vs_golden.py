"""CPU tests (-m "not gpu"): the oracle against golden vectors produced by the REAL reference
(tools/make_goldens.py imports /root/reference/smplifyx; only arrays are committed), plus
the cross-checks that stand in for the unpinned LBS."""
import os

import numpy as np
import pytest
import torch

import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


# ---------------------------------------------------------------- in-tree reference pieces
def test_tables_match_reference():
    from smplifyx_amd import utils as U
    g = gold("tables")
    for key in g.files:
        fmt, flags = key.rsplit("_", 1)
        h, f, c = [bool(int(x)) for x in flags]
        assert np.array_equal(U.smpl_to_annotation("smplx", h, f, c, fmt), g[key]), key


def test_euler_matches_reference():
    from smplifyx_amd import utils as U
    g = gold("euler")
    assert np.abs(U.euler_xyz_from_matrix(g["rand_R"]) - g["rand_euler"]).max() < 2e-6
    for name in ("02_cropped", "18_cropped"):
        pose, glob = U.regression_prior_pose(
            "combined", expose={"body_pose": g[name + "_expose_R"], "global_orient": g[name + "_expose_gR"]},
            pixie={"body_pose": g[name + "_pixie_R"], "global_pose": g[name + "_expose_gR"]})
        assert np.abs(pose - g[name + "_combined_pose"]).max() < 2e-6
        assert np.abs(glob - g[name + "_global"]).max() < 2e-6
    # SURVEY.md 8c probe values
    assert np.allclose(g["02_cropped_combined_pose"][:4], [-1.1551647, -0.18123293, 0.17945433, -1.000505], atol=1e-6)


def test_gmof_camera_priors_match_reference():
    from oracle import objective as obj
    g = gold("objective")
    x = torch.tensor(g["gmof_in"])
    assert np.allclose(obj.gmof(x, 100.0).numpy(), g["gmof_out"], rtol=1e-12)
    pts = torch.tensor(g["cam_pts"], requires_grad=True)
    t = torch.tensor(g["cam_t"], requires_grad=True)
    f = torch.full([1], 1234.5, dtype=torch.float64)
    uv = obj.project(pts, torch.eye(3, dtype=torch.float64)[None], t, f, f, torch.tensor([[400.0, 300.0]], dtype=torch.float64))
    assert np.allclose(uv.detach().numpy(), g["cam_uv"], rtol=1e-12)
    assert np.allclose(obj.angle_prior(torch.tensor(g["angle_in"])).numpy(), g["angle_out"], rtol=1e-12)
    assert np.allclose([obj.rel_change(10.0, 9.0), obj.rel_change(0.5, 0.6), obj.rel_change(-3.0, 2.0)], g["relchange"])


@pytest.mark.parametrize("tag,uh,uf", [("body", False, False), ("full", True, True)])
def test_smplify_loss_value_and_gradients_match_reference(tag, uh, uf):
    from collections import namedtuple
    from oracle import objective as obj
    g = gold("objective")
    T = lambda k: torch.tensor(g[tag + "_" + k], dtype=torch.float64, requires_grad=True)
    joints, full_pose, betas, lh, rh, expr, jaw, emb = [T(k) for k in
                                                        ("joints", "full_pose", "betas", "lh", "rh", "expr", "jaw", "emb")]
    cam_t = torch.tensor([[0.05, 0.1, 20.0]], dtype=torch.float64, requires_grad=True)
    f = torch.full([1], 5000.0, dtype=torch.float64)
    proj = obj.project(joints, torch.eye(3, dtype=torch.float64)[None], cam_t, f, f,
                       torch.tensor([[400.0, 300.0]], dtype=torch.float64))
    MO = namedtuple("MO", ["full_pose", "betas", "body_pose", "left_hand_pose", "right_hand_pose", "expression", "jaw_pose"])
    w = {k: torch.tensor(v, dtype=torch.float64) for k, v in dict(
        data_weight=1000 / 600, body_pose_weight=300.0, shape_weight=50.0, bending_prior_weight=3.17 * 300.0,
        hand_prior_weight=4.78, expr_prior_weight=5.0, jaw_prior_weight=[100.0, 1000.0, 1000.0]).items()}
    terms = obj.smplify_terms(MO(full_pose, betas, emb, lh, rh, expr, jaw), proj, torch.tensor(g[tag + "_gt"]),
                              torch.tensor(g[tag + "_conf"]), torch.tensor(g[tag + "_jw"]), w, emb, use_vposer=False,
                              regression_pose=torch.tensor(g[tag + "_reg"]), stage=1, num_stages=3,
                              use_joints_conf=True, use_hands=uh, use_face=uf, rho=100)
    assert abs(terms["total"].item() - float(g[tag + "_total"])) <= 1e-12 * abs(float(g[tag + "_total"]))
    terms["total"].backward()
    for name, t in (("joints", joints), ("full_pose", full_pose), ("betas", betas), ("lh", lh), ("rh", rh),
                    ("expr", expr), ("jaw", jaw), ("emb", emb), ("cam_t", cam_t)):
        got = t.grad.numpy() if t.grad is not None else np.zeros(tuple(t.shape))
        ref = g[tag + "_d_" + name]
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(ref).max())), name
    # camera-init loss, with and without the use_conf broadcast quirk
    for uc in (0, 1):
        j2 = torch.tensor(g[tag + "_joints"], requires_grad=True)
        ct = torch.tensor([[0.05, 0.1, 20.0]], dtype=torch.float64, requires_grad=True)
        p2 = obj.project(j2, torch.eye(3, dtype=torch.float64)[None], ct, f, f, torch.tensor([[400.0, 300.0]], dtype=torch.float64))
        v = obj.camera_init_loss(p2, torch.tensor(g[tag + "_gt"]), [9, 12, 2, 5], torch.tensor(1000 / 600, dtype=torch.float64),
                                 torch.tensor(1e2, dtype=torch.float64), ct[:, 2], torch.tensor([18.0], dtype=torch.float64),
                                 joints_conf=torch.tensor(g[tag + "_conf"]), use_conf=bool(uc))
        assert abs(v.item() - float(g["%s_caminit%d" % (tag, uc)])) <= 1e-12 * abs(v.item())
        v.backward()
        assert np.allclose(j2.grad.numpy(), g["%s_caminit%d_dj" % (tag, uc)], rtol=1e-10, atol=1e-9)
        assert np.allclose(ct.grad.numpy(), g["%s_caminit%d_dt" % (tag, uc)], rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("fname", ["rosen", "quad"])
@pytest.mark.parametrize("N", [2, 6, 10])
def test_lbfgs_machine_reproduces_reference_trajectory_fp64(fname, N):
    """Every closure evaluation (point and value) of run_fitting+LBFGS('lbfgsls') in fp64."""
    from oracle.lbfgs_machine import StageMachine
    g = gold("lbfgs")
    key = "%s_%d_f64" % (fname, N)
    trace = g[key + "_trace"]

    def fn(x):
        if fname == "rosen":
            return (100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2).sum()
        a = torch.arange(1, x.numel() + 1, dtype=x.dtype) ** 2
        return 0.5 * (a * (x - 0.3) ** 2).sum() + 0.1 * torch.sin(3 * x).sum()
    m = StageMachine(g[key + "_x0"], maxiters=30, dtype=np.float64)
    pts = []
    while not m.done:
        x = torch.tensor(m.x_trial, dtype=torch.float64, requires_grad=True)
        l = fn(x)
        l.backward()
        pts.append(np.concatenate([[l.item()], m.x_trial]))
        m.feed(l.item(), x.grad.numpy())
    pts = np.stack(pts)
    assert pts.shape == trace.shape, (pts.shape, trace.shape)          # same number of evaluations
    assert np.allclose(pts, trace, rtol=1e-7, atol=1e-9)
    assert abs(m.result - float(g[key + "_res"])) <= 1e-7 * max(1.0, abs(m.result))
    assert np.allclose(m.x, g[key + "_xf"], rtol=1e-7, atol=1e-9)


def test_lbfgs_machine_fp32_exact_on_small_problem():
    """N=2 has order-independent dot products: fp32 run is bit-identical to the reference."""
    from oracle.lbfgs_machine import StageMachine
    g = gold("lbfgs")
    trace = g["rosen_2_f32_trace"]
    m = StageMachine(g["rosen_2_f32_x0"], maxiters=30, dtype=np.float32)
    k = 0
    while not m.done:
        x = torch.tensor(m.x_trial, dtype=torch.float32, requires_grad=True)
        l = (100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2).sum()
        l.backward()
        assert np.array_equal(m.x_trial.astype(np.float64), trace[k, 1:]), k
        m.feed(l.item(), x.grad.numpy())
        k += 1
    assert k == trace.shape[0]


def test_reuse_entry_eval_changes_nothing_but_the_count():
    from oracle.lbfgs_machine import run_stage

    def rosen(x):
        x = x.astype(np.float64)
        f = np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
        g = np.zeros_like(x)
        g[:-1] += -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200 * (x[1:] - x[:-1] ** 2)
        return np.float32(f), g.astype(np.float32)
    a = run_stage(rosen, np.full(10, 0.5), maxiters=30)
    b = run_stage(rosen, np.full(10, 0.5), maxiters=30, reuse_entry_eval=True)
    assert np.array_equal(a.x, b.x) and a.result == b.result and b.evals < a.evals


# ---------------------------------------------------------------- LBS (parity unpinned): cross-checks
def test_lbs_two_independent_implementations_agree(synth_model):
    from oracle import lbs_numpy
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml")
    bm = H.oracle_model(synth_model, cfg, torch.float64)
    rng = np.random.RandomState(5)
    P = H.random_params(rng, 1)
    bm.reset_params(**{k: v for k, v in P.items() if k != "pose_embedding"})
    out = bm(return_verts=True, body_pose=torch.tensor(P["pose_embedding"], dtype=torch.float64), return_full_pose=True)
    q = {k: v[0] for k, v in P.items()}
    q["body_pose"] = q.pop("pose_embedding")
    ref = lbs_numpy.forward(synth_model, q, joint_map=H.joint_map_for(cfg))
    assert np.abs(out.vertices[0].detach().numpy() - ref["vertices"]).max() < 1e-12
    assert np.abs(out.joints[0].detach().numpy() - ref["joints"]).max() < 1e-12


def test_lbs_invariants(synth_model):
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml")
    bm = H.oracle_model(synth_model, cfg, torch.float64)
    bm.pose_mean.zero_()
    betas = np.random.RandomState(1).normal(size=(1, 10))
    bm.reset_params(betas=betas)
    with torch.no_grad():
        o = bm(return_verts=True, body_pose=torch.zeros([1, 63], dtype=torch.float64))
    v_shaped = bm.v_template + torch.einsum("bl,mkl->bmk", torch.cat([bm.betas, bm.expression], 1), bm.shapedirs)
    assert (o.vertices - v_shaped).abs().max() < 1e-6       # zero pose => rest shape (eps inside the norm: not exact)
    assert abs(float(bm.lbs_weights.sum(1).min()) - 1) < 1e-6
    # global rotation only => rigid rotation about joint 0
    from oracle.body_model import batch_rodrigues
    go = torch.tensor([[0.3, -0.5, 0.8]], dtype=torch.float64)
    bm.reset_params(betas=betas, global_orient=go)
    with torch.no_grad():
        o2 = bm(return_verts=True, body_pose=torch.zeros([1, 63], dtype=torch.float64))
    R = batch_rodrigues(go)[0]
    J0 = torch.einsum("bik,ji->bjk", v_shaped, bm.J_regressor)[0, 0]
    expect = (v_shaped[0] - J0) @ R.T + J0
    assert (o2.vertices[0] - expect).abs().max() < 1e-6


def test_lbs_gradient_finite_differences(synth_model):
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)
    bm = H.oracle_model(synth_model, cfg, torch.float64)
    rng = np.random.RandomState(2)
    pose = torch.tensor(0.3 * rng.normal(size=(1, 63)), requires_grad=True)
    wts = torch.tensor(rng.normal(size=(1, 25, 3)))
    f = lambda p: (bm(return_verts=False, body_pose=p).joints * wts).sum()
    f(pose).backward()
    g = pose.grad.numpy().ravel()
    for i in (0, 7, 30, 62):
        e = torch.zeros_like(pose); e[0, i] = 1e-6
        fd = (f(pose.detach() + e) - f(pose.detach() - e)).item() / 2e-6
        assert abs(fd - g[i]) < 1e-6 * max(1, abs(g[i]))


# ---------------------------------------------------------------- end-to-end vs the reference driver
@pytest.mark.parametrize("i", [0, 1])
def test_frame_driver_matches_reference_fit(synth_model, i):
    """oracle.fit_frame vs the reference's fit_single_frame on a well-posed synthetic frame
    (fp32 both).  Later stages are chaotic: the reference's own fp32-vs-fp64 spread, stored in
    the same golden file, sets the tolerance."""
    path = os.path.join(GOLD, "e2e_synth.npz")
    if not os.path.exists(path):
        pytest.skip("e2e golden not generated")
    g = np.load(path)
    cfg = H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)
    cfg["use_camera_prior"] = False
    frames = dict(keypoints=g["keypoints"], reg_pose=g["reg_pose"], reg_global=g["reg_global"], H=600, W=800, focal=5000.0)
    torch.set_num_threads(4)
    ref = H.oracle_frame_fit(synth_model, cfg, frames, i, dtype=torch.float32).run()
    got = np.array([ref["cam_loss"]] + list(ref["stage_losses"]))
    want32, want64 = g["f%d_f32_losses" % i], g["f%d_f64_losses" % i]
    spread = np.abs(want32 - want64) / np.abs(want64)
    rel = np.abs(got - want32) / np.abs(want32)
    assert rel[0] < 1e-5                                      # camera stage: well conditioned
    assert np.all(rel < np.maximum(5 * spread, 2e-3)), (rel, spread)


def test_frame_driver_matches_reference_fit_on_the_benchmark_configuration(synth_model):
    """The oracle frame driver is bench.py's cpu_baseline ("port"): on the benchmark's own configuration
    (fit_smplx_smplifyx.yaml weights, 5 body stages, body-only, regression prior) it must reproduce the
    reference's fit of benchmark frame 3 (tests/golden/e2e_bench.npz), fp32 against fp32, within the
    reference's own fp32 / fp64 spread."""
    path = os.path.join(GOLD, "e2e_bench.npz")
    g = np.load(path)
    cfg = H.load_cfg("fit_smplx_smplifyx.yaml", use_hands=False, use_face=False, use_vposer=False)
    cfg["use_camera_prior"] = False
    frames = dict(keypoints=g["keypoints"], reg_pose=g["reg_pose"], reg_global=g["reg_global"], H=600, W=800, focal=5000.0)
    torch.set_num_threads(4)
    i = 3
    ref = H.oracle_frame_fit(synth_model, cfg, frames, i, dtype=torch.float32).run()
    got = np.array([ref["cam_loss"]] + list(ref["stage_losses"]))
    want32, want64 = g["f%d_f32_losses" % i], g["f%d_f64_losses" % i]
    spread = np.abs(want32 - want64) / np.abs(want64)
    rel = np.abs(got - want32) / np.abs(want32)
    assert rel[0] < 1e-5
    assert np.all(rel[1:4] < np.maximum(2 * spread[1:4], 3e-3)), (rel, spread)
    assert np.all(rel[4:] < np.maximum(3 * spread[4:], 5e-2)), (rel, spread)


# ---------------------------------------------------------------------------------------------------------------------------
# The interpenetration oracle (oracle/penetration.py, PARITY UNPINNED: the package is absent) against a second, independent
# statement of the same published algorithm (oracle/penetration_numpy.py: plain loops, other routes to the same quantities),
# finite differences of its gradient, and the invariants the term must have.
def _two_blobs(offset, seed=0):
    """two small closed meshes (octahedra subdivided once) that intersect when offset is small; part labels 0 / 2 (2's parent = 1)"""
    def octa(c, r):
        v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], float)
        f = [[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]]
        vs, fs, mid = [tuple(x) for x in v], [], {}
        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = (np.array(vs[a]) + np.array(vs[b])) / 2
                vs.append(tuple(p / np.linalg.norm(p))); mid[k] = len(vs) - 1
            return mid[k]
        for a, b, c_ in f:
            ab, bc, ca = m(a, b), m(b, c_), m(c_, a)
            fs += [[a, ab, ca], [ab, b, bc], [ca, bc, c_], [ab, bc, ca]]
        return np.array(vs) * r + np.array(c), np.array(fs)
    rng = np.random.RandomState(seed)
    v0, f0 = octa([0, 0, 0], 0.10)
    v1, f1 = octa([offset, 0.013, 0.021], 0.08)
    v = np.concatenate([v0, v1]) + 0.003 * rng.normal(size=(len(v0) + len(v1), 3))
    f = np.concatenate([f0, f1 + len(v0)])
    segm = np.concatenate([np.zeros(len(f0), int), 2 * np.ones(len(f1), int)])
    parents = np.concatenate([-np.ones(len(f0), int), np.ones(len(f1), int)])
    return v, f, segm, parents


@pytest.mark.parametrize("sigma,outside,p2p", [(0.5, True, False), (0.05, False, False), (1e-2, True, True), (1e-4, True, False)])
def test_penetration_two_independent_implementations_agree(sigma, outside, p2p):
    from oracle import penetration as OP, penetration_numpy as PN
    v, f, segm, parents = _two_blobs(0.12)
    pairs = OP.candidate_pairs(v, f, segm, parents)
    assert [tuple(p) for p in pairs.tolist()] == PN.colliding_pairs(v, f, segm, parents) and len(pairs) > 20
    # the part rules, each on its own: same part / parent-child / an ignored pair -> nothing left; no labels -> self-collisions too
    assert PN.colliding_pairs(v, f, np.zeros_like(segm), -np.ones_like(parents)) == [] == OP.candidate_pairs(v, f, np.zeros_like(segm), -np.ones_like(parents)).tolist()
    assert OP.candidate_pairs(v, f, segm, np.where(segm == 2, 0, -1)).tolist() == PN.colliding_pairs(v, f, segm, np.where(segm == 2, 0, -1)) == []
    assert OP.candidate_pairs(v, f, segm, parents, ["2,0"]).tolist() == PN.colliding_pairs(v, f, segm, parents, [(0, 2)]) == []
    assert [tuple(p) for p in OP.candidate_pairs(v, f).tolist()] == PN.colliding_pairs(v, f)
    a = float(OP.penetration_loss(torch.tensor(v, dtype=torch.float64), f, pairs, sigma, outside, p2p))
    b = PN.loss(v, f, [tuple(p) for p in pairs.tolist()], sigma, outside, p2p)
    assert a > 0 and abs(a - b) <= 1e-10 * a, (a, b)


def test_penetration_gradient_finite_differences_and_invariants():
    from oracle import penetration as OP
    v, f, segm, parents = _two_blobs(0.12, seed=1)
    sigma = 0.05
    lo, g, pairs = OP.penetration(v, f, segm, parents, None, sigma=sigma)
    rng = np.random.RandomState(2)
    for _ in range(6):          # central differences along random directions (the field is C1: quadratic band, linear tail)
        d = rng.normal(size=v.shape); d /= np.linalg.norm(d)
        h = 1e-6
        fp = float(OP.penetration_loss(torch.tensor(v + h * d), f, pairs, sigma))
        fm = float(OP.penetration_loss(torch.tensor(v - h * d), f, pairs, sigma))
        assert abs((fp - fm) / (2 * h) - float((g * d).sum())) <= 1e-5 * np.abs(g).sum(), ((fp - fm) / (2 * h), float((g * d).sum()))
    # a rigid motion of the whole scene changes nothing for a FIXED pair list, and the gradient turns with it (the candidate set
    # itself is found on axis-aligned boxes: it may change under a rotation, and does on this scene)
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec([0.3, -0.7, 0.5]).as_matrix()
    vt = torch.tensor(v @ R.T + [0.3, -0.1, 2.0], requires_grad=True)
    lo2 = OP.penetration_loss(vt, f, pairs, sigma)
    lo2.backward()
    assert abs(float(lo2) - lo) <= 1e-9 * lo and np.abs(vt.grad.numpy() - g @ R.T).max() <= 1e-8 * np.abs(g).max()
    lo_t, _, pairs_t = OP.penetration(v + [0.3, -0.1, 2.0], f, segm, parents, None, sigma=sigma)       # a translation keeps the boxes' overlaps
    assert np.array_equal(pairs_t, pairs) and abs(lo_t - lo) <= 1e-9 * lo
    # separated meshes: no candidates, zero loss; relabelling the triangles' corner order (same orientation) changes nothing
    v_far, f_far, s_far, p_far = _two_blobs(0.5)
    assert len(OP.candidate_pairs(v_far, f_far, s_far, p_far)) == 0 and OP.penetration(v_far, f_far, s_far, p_far)[0] == 0.0
    lo3 = float(OP.penetration_loss(torch.tensor(v), np.roll(f, 1, axis=1), pairs, sigma))
    assert abs(lo3 - lo) <= 1e-12 * lo
    # the capped list (assumption A1): symmetric, lowest ids, cut count consistent
    cnt = np.bincount(pairs.reshape(-1), minlength=len(f))
    cap = max(2, int(cnt.max() // 2))
    op, n_cut = OP.ordered_pairs_capped(pairs, cap)
    sset = set(map(tuple, op.tolist()))
    assert n_cut > 0 and len(op) == 2 * len(pairs) - n_cut and all((b_, a_) in sset for a_, b_ in sset)
    for t in np.unique(op[:, 0]):
        mine = np.sort(op[op[:, 0] == t, 1])
        allp = np.sort(np.concatenate([pairs[pairs[:, 0] == t, 1], pairs[pairs[:, 1] == t, 0]]))
        assert len(mine) <= cap and set(mine) <= set(allp[:cap])


def test_cubic_interpolation_overflow_matches_reference():
    """tests/golden/cubic_overflow.npz: the reference's _cubic_interpolate (lbfgs_ls.py:11-36) with Python-float function values and
    0-d TENSOR directional derivatives where a trial point evaluates to 1e19 ... 3e29 (what a unit step along a direction blown up
    by the interpenetration term produces: tools/pen_nan_probe.py).  In fp32 `d1 ** 2` overflows and NaN comes back through
    min(max(...)); in fp64 the same inputs give a finite step.  The specification machine's dual-typed scalars reproduce both --
    the device's NaN on such a fit is the reference's own arithmetic, not a difference from it."""
    from oracle import lbfgs_machine as LM
    g = gold("cubic_overflow")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        old = LM.Sc.dt
        LM.Sc.dt = dt
        try:
            for row, want in zip(g["rows"], g[tag]):
                x1, f1, g1, x2, f2, g2 = (float(v) for v in row)
                got = LM.cubic_interpolate(LM.P(x1), LM.P(f1), LM.T(g1), LM.P(x2), LM.P(f2), LM.T(g2)).v
                assert (np.isnan(got) and np.isnan(want)) or got == dt(want) or abs(float(got) - want) <= 1e-6 * abs(want), (tag, row, got, want)
        finally:
            LM.Sc.dt = old
    assert np.isnan(g["f32"][:3]).all() and np.isfinite(g["f64"]).all()
