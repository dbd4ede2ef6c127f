"""Step-level parity of the on-device optimiser (csrc/lbfgs_body.h) with its specification
oracle/lbfgs_machine.py -- which tests/golden/lbfgs.npz pins to the reference's
FittingMonitor.run_fitting (smplifyx/fitting.py:147-217), LBFGS.step (optimizers/lbfgs_ls.py:256-445) and
_strong_Wolfe / _cubic_interpolate (:11-167) evaluation by evaluation.

Whole fits only show the optimiser through a chaotic trajectory.  Here the device's trace of a stage (every finished
line search: step length, loss, evaluations; every LBFGS.step: entry loss, cumulative evaluations and iterations; the
stage result -- include/sfx.h sfx_batch_trace) is compared record by record with the machine's, the machine being
driven (a) by the SAME HIP closure, so that any difference is the state machine's -- dot products are summed in a
different order and the two-loop recursion is blocked on the device, so directions differ in the last bits: every branch
decision and every count must agree exactly above the fp32 noise floor, accepted losses to 5e-5, step lengths (cubic
interpolations: ill-conditioned) to 5e-3 -- and (b) by the oracle's torch fp32 closure on the camera stage (N = 6):
same decisions and counts, losses along the way within 1e-3, the same stage result.
"""
import numpy as np
import pytest
import torch

import helpers as H
import test_gpu_parity as T

pytestmark = pytest.mark.gpu

@pytest.fixture(scope="module")
def cfg_body():
    return H.load_cfg("fit_smplx_combined_coco25.yaml", use_hands=False, use_face=False)


ORDER = (("betas", 10), ("global_orient", 3), (None, 63), ("left_hand_pose", 12), ("right_hand_pose", 12), ("jaw_pose", 3),
         ("leye_pose", 3), ("reye_pose", 3), ("expression", 10), ("pose_embedding", 63))


def _flat(P, i):
    return np.concatenate([np.zeros(n, np.float32) if k is None else P[k][i] for k, n in ORDER])


def _unflat(x):
    out, o = {}, 0
    for k, n in ORDER:
        if k is not None:
            out[k] = x[o:o + n][None].astype(np.float32)
        o += n
    return out


def _setup(synth_model, cfg_body):
    g = T._golden("e2e_synth")
    cfg = dict(cfg_body); cfg["use_camera_prior"] = False
    dm = T._dm(synth_model, cfg)
    frames = dict(keypoints=g["keypoints"], reg_pose=g["reg_pose"], reg_global=g["reg_global"], H=600, W=800, focal=5000.0)
    return cfg, dm, frames


def _machine(x0, cfg, groups, reuse):
    from oracle.lbfgs_machine import StageMachine
    return StageMachine(x0, groups=groups, maxiters=cfg["maxiters"], ftol=cfg["ftol"], gtol=cfg["gtol"], lr=cfg.get("lr", 1.0),
                        dtype=np.float32, reuse_entry_eval=reuse)


def _above_noise_floor(rec, floor=1e-6):
    """Number of leading records before the first accepted loss within `floor` (relative) of the stage's result: past that
    point successive losses differ by an ulp or two of fp32 and every Armijo / curvature test is a coin toss decided by
    the summation order of a dot product -- only the part of the trajectory above the noise floor can agree event by event."""
    rec = np.asarray(rec, np.float64)
    final = rec[rec[:, 0] == 2][-1, 1] if (rec[:, 0] == 2).any() else rec[rec[:, 0] == 0][-1, 2]
    for k, r in enumerate(rec):
        if r[0] == 0 and abs(r[2] - final) <= floor * abs(final):
            return k
    return len(rec)


def _compare(dev, mac, rtol, what, whole_stage=True, t_rtol=None):
    dev = np.asarray(dev, np.float64); mac = np.asarray(mac, np.float64)
    n = min(_above_noise_floor(dev), _above_noise_floor(mac)) if whole_stage else min(len(dev), len(mac))
    assert n >= 8, (what, "too few events above the noise floor", n)
    d, m = dev[:n], mac[:n]
    assert np.array_equal(d[:, 0], m[:, 0]), what                         # record types: the same sequence of events
    ls, st = d[:, 0] == 0, d[:, 0] == 1
    assert np.array_equal(d[ls, 3], m[ls, 3]), (what, "evaluations per line search")
    assert np.array_equal(d[st, 2], m[st, 2]) and np.array_equal(d[st, 3], m[st, 3]), (what, "evaluations / iterations per LBFGS.step")
    # (a step length comes out of a cubic interpolation -- differences of nearly equal numbers: rounding of the directional
    #  derivative's dot product is amplified ~500 x there, the loss at the accepted point is flat in t and agrees far better)
    if t_rtol is not False:
        assert np.allclose(d[ls, 1], m[ls, 1], rtol=t_rtol or 100 * rtol, atol=0), (what, "step lengths", np.abs(d[ls, 1] / m[ls, 1] - 1).max())
    assert np.allclose(d[ls, 2], m[ls, 2], rtol=rtol, atol=0), (what, "accepted losses", np.abs(d[ls, 2] / m[ls, 2] - 1).max())
    assert np.allclose(d[st, 1], m[st, 1], rtol=rtol, atol=0), (what, "entry losses", np.abs(d[st, 1] / m[st, 1] - 1).max())
    if whole_stage:      # the converged tail: same result, comparable work
        assert dev[-1, 0] == 2 and mac[-1, 0] == 2
        assert abs(dev[-1, 1] - mac[-1, 1]) <= 2e-6 * abs(mac[-1, 1]), (what, "stage result", dev[-1], mac[-1])
        # (on the noise floor a line search costs 1 evaluation or 4, by the toss of a coin, several times in a row)
        assert abs(dev[-1, 2] - mac[-1, 2]) <= max(0.5 * mac[-1, 2], 20), (what, "closure evaluations of the stage", dev[-1], mac[-1])
    return n


@pytest.mark.parametrize("reuse", [False, True])
def test_camera_stage_steps_match_the_machine(synth_model, cfg_body, reuse):
    cfg, dm, frames = _setup(synth_model, cfg_body)
    for i in range(2):
        fb = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=reuse)
        fb.guess_init(cfg["body_tri_idxs"])
        P0 = fb.get_params()
        fb.trace(4096)
        fb.fit(first_stage=-1, last_stage=-1)
        dev = fb.get_trace()[0]
        assert dev[-1, 0] == 2 and dev[-1, 3] == -1
        # (a) the machine on the HIP closure
        fc = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=reuse)
        fc.guess_init(cfg["body_tri_idxs"])
        m = _machine(np.concatenate([P0["cam_translation"][0], P0["global_orient"][0]]), cfg, [(0, 3, True), (3, 3, True)], reuse)
        while not m.done:
            x = m.x_trial
            fc.set_params(regression_pose=frames["reg_pose"][i:i + 1], cam_translation=x[None, :3], global_orient=x[None, 3:],
                          pose_embedding=P0["pose_embedding"])
            f, gr = fc.closure(-1)
            m.feed(f[0], gr[0])
        mac = np.array(m.records); mac[-1, 3] = -1
        n = _compare(dev, mac, 5e-5, "camera stage, frame %d, machine on the HIP closure" % i)
        assert n >= 20, n                      # (the camera stage takes ~30 line searches; the last few sit on the noise floor)
        assert fb.stats()["stage_evals"][0, 0] == dev[-1, 2]
        # (b) the machine on the oracle's torch fp32 closure
        Q = dict(P0); Q["est_tz"] = P0["cam_translation"][:, 2].copy(); Q.pop("body_pose")
        Qf = {k: np.repeat(v, 2, 0) if v.shape[0] == 1 else v for k, v in Q.items()}       # (_oracle_closure indexes by frame)
        m2 = _machine(np.concatenate([P0["cam_translation"][0], P0["global_orient"][0]]), cfg, [(0, 3, True), (3, 3, True)], reuse)
        fr1 = {k: (v[i:i + 1] if isinstance(v, np.ndarray) else v) for k, v in frames.items()}
        while not m2.done:
            x = m2.x_trial
            Qx = {k: v[:1].copy() for k, v in Qf.items()}
            Qx["cam_translation"] = x[None, :3].astype(np.float32); Qx["global_orient"] = x[None, 3:].astype(np.float32)
            f, gr = T._oracle_closure(synth_model, cfg, fr1, 0, Qx, -1, dtype=torch.float32)
            m2.feed(f, gr)
        mac2 = np.array(m2.records); mac2[-1, 3] = -1
        _compare(dev, mac2, 1e-3, "camera stage, frame %d, machine on the oracle's fp32 closure" % i, t_rtol=0.05)


def test_non_default_lbfgs_hyper_parameters_match_the_machine(synth_model, cfg_body):
    """LBFGS(tolerance_grad, tolerance_change, max_eval, history_size) other than the defaults optim_factory.py leaves in
    place (weak point of round 2: the handle refused them): camera stage with history 3, tolerance_grad 1e-3,
    tolerance_change 1e-7, max_eval 20, device trace against the specification machine on the same HIP closure."""
    from oracle.lbfgs_machine import StageMachine
    cfg, dm, frames = _setup(synth_model, cfg_body)
    cfg = dict(cfg, lbfgs_history_size=3, lbfgs_tolerance_grad=1e-3, lbfgs_tolerance_change=1e-7, lbfgs_max_eval=20)
    i = 0
    fb = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=False)
    fb.guess_init(cfg["body_tri_idxs"])
    P0 = fb.get_params()
    fb.trace(4096)
    fb.fit(first_stage=-1, last_stage=-1)
    dev = fb.get_trace()[0]
    fc = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=False)
    fc.guess_init(cfg["body_tri_idxs"])
    m = StageMachine(np.concatenate([P0["cam_translation"][0], P0["global_orient"][0]]), groups=[(0, 3, True), (3, 3, True)],
                     maxiters=cfg["maxiters"], ftol=cfg["ftol"], gtol=cfg["gtol"], lr=cfg.get("lr", 1.0), dtype=np.float32,
                     reuse_entry_eval=False, history=3, tol_grad=1e-3, tol_change=1e-7, max_eval=20)
    while not m.done:
        x = m.x_trial
        fc.set_params(regression_pose=frames["reg_pose"][i:i + 1], cam_translation=x[None, :3], global_orient=x[None, 3:],
                      pose_embedding=P0["pose_embedding"])
        f, gr = fc.closure(-1)
        m.feed(f[0], gr[0])
    mac = np.array(m.records); mac[-1, 3] = -1
    # (history 3 and 20 evaluations per step make a poor optimiser: ~170 evaluations of wandering, in which rounding
    #  separates the two trajectories long before the stage ends -- compared event by event over the first LBFGS.step call)
    k2 = np.flatnonzero(dev[:, 0] == 1)[0] + 1             # (the first LBFGS.step: 16 line searches, the 3-pair history wraps 5 times)
    assert k2 >= 12 and np.array_equal(dev[:k2, 0], mac[:k2, 0])
    _compare(dev[:k2], mac[:k2], 5e-5, "camera stage, non-default LBFGS hyper-parameters", whole_stage=False, t_rtol=1e-3)
    # max_eval 20 binds: an LBFGS.step of the default configuration takes up to 37 evaluations, here none takes more than 20 + line search
    ev = dev[dev[:, 0] == 1][:, 2]
    assert np.all(np.diff(np.concatenate([[0], ev])) <= 20 + 25)
    # and the defaults give another trajectory (the parameters really reach the device)
    fd = H.engine_batch_from_frames(dm, dict(cfg_body, use_camera_prior=False), frames, [i], lbs_mode="rows", reuse=False)
    fd.guess_init(cfg["body_tri_idxs"]); fd.trace(4096); fd.fit(first_stage=-1, last_stage=-1)
    assert len(fd.get_trace()[0]) != len(dev) or not np.array_equal(fd.get_trace()[0], dev)


def test_lbfgs_max_iter_other_than_maxiters_matches_the_machine(synth_model, cfg_body):
    """LBFGS(max_iter=7) under FittingMonitor(maxiters=30) -- two different numbers where optim_factory.py:15 passes one (the
    handle refused that until round 4): an LBFGS.step ends after 7 iterations / 8 evaluations (lbfgs_ls.py:203,304,419), the
    zoom phase is bounded by the same 7 (:397 hands max_iter to _strong_Wolfe), run_fitting still makes up to 30 steps.  Camera
    stage, device trace against the specification machine on the same HIP closure."""
    from oracle.lbfgs_machine import StageMachine
    cfg, dm, frames = _setup(synth_model, cfg_body)
    cfg = dict(cfg, lbfgs_max_iter=7)
    i = 0
    fb = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=False)
    fb.guess_init(cfg["body_tri_idxs"])
    P0 = fb.get_params()
    fb.trace(4096)
    fb.fit(first_stage=-1, last_stage=-1)
    dev = fb.get_trace()[0]
    fc = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=False)
    fc.guess_init(cfg["body_tri_idxs"])
    m = StageMachine(np.concatenate([P0["cam_translation"][0], P0["global_orient"][0]]), groups=[(0, 3, True), (3, 3, True)],
                     maxiters=cfg["maxiters"], ftol=cfg["ftol"], gtol=cfg["gtol"], lr=cfg.get("lr", 1.0), dtype=np.float32,
                     reuse_entry_eval=False, max_iter=7)
    while not m.done:
        x = m.x_trial
        fc.set_params(regression_pose=frames["reg_pose"][i:i + 1], cam_translation=x[None, :3], global_orient=x[None, 3:],
                      pose_embedding=P0["pose_embedding"])
        f, gr = fc.closure(-1)
        m.feed(f[0], gr[0])
    mac = np.array(m.records); mac[-1, 3] = -1
    _compare(dev, mac, 5e-5, "camera stage, LBFGS max_iter 7 under maxiters 30", t_rtol=1e-3)
    st = dev[dev[:, 0] == 1]
    # no LBFGS.step makes more than 7 iterations or 8 evaluations + its last line search, and the stage needs several steps
    assert len(st) >= 3 and np.all(np.diff(np.concatenate([[0], st[:, 3]])) <= 7)
    assert np.all(np.diff(np.concatenate([[0], st[:, 2]])) <= 8 + 25)
    # the default (max_iter = maxiters = 30) takes another path through the same stage
    fd = H.engine_batch_from_frames(dm, dict(cfg_body, use_camera_prior=False), frames, [i], lbs_mode="rows", reuse=False)
    fd.guess_init(cfg["body_tri_idxs"]); fd.trace(4096); fd.fit(first_stage=-1, last_stage=-1)
    assert len(fd.get_trace()[0]) != len(dev) or not np.array_equal(fd.get_trace()[0], dev)


def test_zero_tolerances_reach_the_device(synth_model, cfg_body):
    """LBFGS(tolerance_grad=0, tolerance_change=0) is legal in the reference (lbfgs_ls.py:291,301,437,442: the tests can then
    only fire on exact zeros); until round 4 a zero in sfx_batch_cfg meant "default" and silently became 1e-5 / 1e-9.  The
    camera stage with both set to 0 follows the specification machine run with zeros, and differs from the default run."""
    from oracle.lbfgs_machine import StageMachine
    cfg, dm, frames = _setup(synth_model, cfg_body)
    i = 0
    runs = {}
    for name, extra in (("zero", dict(lbfgs_tolerance_grad=0.0, lbfgs_tolerance_change=0.0)), ("default", {})):
        fb = H.engine_batch_from_frames(dm, dict(cfg, **extra), frames, [i], lbs_mode="rows", reuse=False)
        fb.guess_init(cfg["body_tri_idxs"])
        P0 = fb.get_params()
        fb.trace(8192)
        fb.fit(first_stage=-1, last_stage=-1)
        runs[name] = fb.get_trace()[0]
    dev = runs["zero"]
    assert len(dev) != len(runs["default"]) or not np.array_equal(dev, runs["default"])
    fc = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=False)
    fc.guess_init(cfg["body_tri_idxs"])
    m = StageMachine(np.concatenate([P0["cam_translation"][0], P0["global_orient"][0]]), groups=[(0, 3, True), (3, 3, True)],
                     maxiters=cfg["maxiters"], ftol=cfg["ftol"], gtol=cfg["gtol"], lr=cfg.get("lr", 1.0), dtype=np.float32,
                     reuse_entry_eval=False, tol_grad=0.0, tol_change=0.0)
    while not m.done:
        x = m.x_trial
        fc.set_params(regression_pose=frames["reg_pose"][i:i + 1], cam_translation=x[None, :3], global_orient=x[None, 3:],
                      pose_embedding=P0["pose_embedding"])
        f, gr = fc.closure(-1)
        m.feed(f[0], gr[0])
    mac = np.array(m.records)
    k1 = np.flatnonzero(dev[:, 0] == 1)[0] + 1              # the first LBFGS.step, event by event
    assert np.array_equal(dev[:k1, 0], mac[:k1, 0])
    _compare(dev[:k1], mac[:k1], 5e-5, "camera stage, zero LBFGS tolerances", whole_stage=False, t_rtol=1e-3)


def test_first_body_stage_steps_match_the_machine(synth_model, cfg_body):
    """N = 182 (119 live variables), ~400 evaluations, history filling up to 100 pairs: blocked two-loop recursion on the
    device against the plain one of the machine, both fed by the HIP closure.  Rounding differs (summation order), so the
    trajectories separate slowly: the first ten LBFGS.step calls must agree event by event, the stage as a whole in its
    result (1e-3) and its work (30 %)."""
    cfg, dm, frames = _setup(synth_model, cfg_body)
    i = 0
    fb = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=True)
    fb.guess_init(cfg["body_tri_idxs"])
    fb.fit(first_stage=-1, last_stage=-1)
    P1 = fb.get_params()
    fb.trace(8192)
    fb.fit(first_stage=0, last_stage=0)
    dev = fb.get_trace()[0]
    fc = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=True)
    fc.guess_init(cfg["body_tri_idxs"])
    groups, o = [], 0
    for k, n in ORDER:
        groups.append((o, n, k is not None)); o += n
    m = _machine(_flat(P1, 0), cfg, groups, True)
    while not m.done:
        fc.set_params(regression_pose=frames["reg_pose"][i:i + 1], cam_translation=P1["cam_translation"], **_unflat(m.x_trial))
        f, gr = fc.closure(0)
        assert np.all(gr[0][13:76] == 0)
        m.feed(f[0], gr[0])
    mac = np.array(m.records)
    # event by event for as long as the two trajectories take the same decisions: at least the first 25 line searches
    # (rounding separates them slowly: after a few dozen iterations one Armijo test falls the other way)
    k = 0
    while k < min(len(dev), len(mac)) and dev[k, 0] == mac[k, 0] and (dev[k, 0] != 0 or dev[k, 3] == mac[k, 3]) \
            and (dev[k, 0] != 1 or (dev[k, 2] == mac[k, 2] and dev[k, 3] == mac[k, 3])):
        k += 1
    assert (dev[:k, 0] == 0).sum() >= 25, ((dev[:k, 0] == 0).sum(), k)
    # (step lengths inside zoom phases are cubic interpolations of nearly equal numbers: tens of per cent apart dozens of
    #  iterations in, while the losses they lead to stay within 2e-3 -- compared only over the first ten searches below)
    _compare(dev[:k], mac[:k], 2e-3, "first body stage, common prefix", whole_stage=False, t_rtol=False)
    k10 = np.flatnonzero(dev[:, 0] == 0)[9] + 1                      # ... and tightly while rounding has not yet spread
    _compare(dev[:k10], mac[:k10], 1e-5, "first body stage, first ten line searches", whole_stage=False, t_rtol=1e-3)
    assert abs(dev[-1, 1] - mac[-1, 1]) <= 1e-3 * abs(mac[-1, 1]), (dev[-1], mac[-1])
    # work: once the two trajectories have separated (above) the ftol test of run_fitting fires a few LBFGS.step calls
    # earlier or later -- observed 377 vs 467 evaluations for results 5e-5 apart (round 3, after the adjoint's summation
    # order changed): the evaluation count is a property of the chaotic tail, bounded loosely
    assert abs(dev[-1, 2] - mac[-1, 2]) <= 0.3 * mac[-1, 2], (dev[-1], mac[-1])


def _two_loop_fp64(S, Y, g):
    """lbfgs_ls.py:312-341 in float64: direction from the window's pairs (oldest first)."""
    ro = 1.0 / np.einsum("ij,ij->i", Y, S)
    q = -g.astype(np.float64)
    al = np.zeros(len(S))
    for i in range(len(S) - 1, -1, -1):
        al[i] = ro[i] * S[i].dot(q)
        q = q - al[i] * Y[i]
    r = q * (Y[-1].dot(S[-1]) / Y[-1].dot(Y[-1]))
    for i in range(len(S)):
        be = ro[i] * Y[i].dot(r)
        r = r + (al[i] - be) * S[i]
    return r


@pytest.mark.parametrize("count", [1, 5, 8, 9, 16, 23, 100, 101, 104, 107, 108, 115, 199, 200, 257])
def test_blocked_two_loop_matches_the_sequential_recursion(count):
    """The device's blocked recursion (8 pairs per block, band of S^T Y, mirrored ring) against the sequential one in fp64 on
    the same pairs, for window lengths that leave partial blocks and ring positions that wrap at every offset: 2e-5 of the
    direction's norm (fp32 dot products of 182 terms), i.e. no block, band entry or mirror row is ever the wrong one."""
    import ctypes as C
    from smplifyx_amd import _capi
    rng = np.random.default_rng(count)
    N, W = 182, 192
    Hm = rng.standard_normal((N, N)) / np.sqrt(N); Hm = Hm @ Hm.T + 0.5 * np.eye(N)      # an SPD "Hessian": y = H s keeps y.s > 0
    S = np.zeros((count, W), np.float32); Y = np.zeros((count, W), np.float32)
    S[:, :N] = rng.standard_normal((count, N)) * 0.1
    Y[:, :N] = (S[:, :N].astype(np.float64) @ Hm).astype(np.float32)
    g = np.zeros(W, np.float32); g[:N] = rng.standard_normal(N)
    d = np.zeros(W, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _capi.check(_capi.load().sfx_lbfgs_two_loop(p(S), p(Y), count, 0, p(g), p(d)))
    w = slice(max(0, count - 100), count)
    ref = _two_loop_fp64(S[w, :N].astype(np.float64), Y[w, :N].astype(np.float64), g[:N])
    assert np.all(d[N:] == 0)
    err = np.abs(d[:N] - ref).max() / np.abs(ref).max()
    assert err < 2e-5, (count, err)


@pytest.mark.parametrize("count,history", [(150, 150), (151, 150), (157, 150), (158, 150), (299, 150), (300, 150), (120, 400), (400, 400),
                                           (407, 400), (408, 400), (801, 400), (60, 37), (101, 101)])
def test_blocked_two_loop_with_a_history_size_other_than_100(count, history):
    """LBFGS(history_size=...) (lbfgs_ls.py:200,271,325 takes any; refused beyond 100 until round 5): a history_size above 100
    gets a ring of that many slots (api.hip hist_ring) -- window, wrap and mirror rows at every offset against the sequential
    recursion in fp64 over the LAST history_size pairs; below 100 the window is shorter than the ring's 100 slots."""
    import ctypes as C
    from smplifyx_amd import _capi
    rng = np.random.default_rng(1000 * history + count)
    N, W = 182, 192
    Hm = rng.standard_normal((N, N)) / np.sqrt(N); Hm = Hm @ Hm.T + 0.5 * np.eye(N)
    S = np.zeros((count, W), np.float32); Y = np.zeros((count, W), np.float32)
    S[:, :N] = rng.standard_normal((count, N)) * 0.1
    Y[:, :N] = (S[:, :N].astype(np.float64) @ Hm).astype(np.float32)
    g = np.zeros(W, np.float32); g[:N] = rng.standard_normal(N)
    d = np.zeros(W, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib = _capi.load()
    _capi.check(lib.sfx_lbfgs_two_loop(p(S), p(Y), count, history, p(g), p(d)))
    w = slice(max(0, count - history), count)
    ref = _two_loop_fp64(S[w, :N].astype(np.float64), Y[w, :N].astype(np.float64), g[:N])
    assert np.all(d[N:] == 0)
    err = np.abs(d[:N] - ref).max() / np.abs(ref).max()
    # (more pairs, more fp32 dot products in the chain: the bound of the 100-pair test scaled with the window)
    assert err < 2e-5 * max(1.0, min(count, history) / 100.0), (count, history, err)
    if count > history and history != 100:      # ... and it is not the 100-pair window's direction
        ref100 = _two_loop_fp64(S[max(0, count - 100):, :N].astype(np.float64), Y[max(0, count - 100):, :N].astype(np.float64), g[:N])
        assert np.abs(d[:N] - ref100).max() / np.abs(ref).max() > 50 * err
    assert lib.sfx_lbfgs_two_loop(p(S), p(Y), count, 401, p(g), p(d)) != 0      # beyond the LDS array of the alphas: refused, loudly


def test_first_body_stage_with_history_size_150(synth_model, cfg_body):
    """The first body stage (several hundred iterations: the history fills up) with LBFGS(history_size=150): identical to the
    default's trace while both hold the same pairs -- the first 100 iterations --, another trajectory afterwards, and the
    specification machine run with history 150 on the same HIP closure arrives at the same result."""
    from oracle.lbfgs_machine import StageMachine
    cfg, dm, frames = _setup(synth_model, cfg_body)
    i = 0
    runs = {}
    for h in (100, 150):
        fb = H.engine_batch_from_frames(dm, dict(cfg, lbfgs_history_size=h), frames, [i], lbs_mode="rows", reuse=True)
        fb.guess_init(cfg["body_tri_idxs"])
        fb.fit(first_stage=-1, last_stage=-1)
        P1 = fb.get_params()
        fb.trace(8192)
        fb.fit(first_stage=0, last_stage=0)
        runs[h] = fb.get_trace()[0]
    d100, d150 = runs[100], runs[150]
    ls100 = np.flatnonzero(d100[:, 0] == 0)
    assert len(ls100) > 130, len(ls100)                                        # the stage outlasts both windows
    k = 0
    while k < min(len(d100), len(d150)) and np.array_equal(d100[k], d150[k]):
        k += 1
    n_ls = int((d100[:k, 0] == 0).sum())
    # every finished line search pushes at most one pair: the two windows hold the same pairs for at least 100 of them, bit for bit
    assert n_ls >= 100, n_ls
    assert k < max(len(d100), len(d150)), "history_size 150 left the trace of the default unchanged"
    fc = H.engine_batch_from_frames(dm, cfg, frames, [i], lbs_mode="rows", reuse=True)
    fc.guess_init(cfg["body_tri_idxs"])
    groups, o = [], 0
    for kk, n in ORDER:
        groups.append((o, n, kk is not None)); o += n
    m = StageMachine(_flat(P1, 0), groups=groups, maxiters=cfg["maxiters"], ftol=cfg["ftol"], gtol=cfg["gtol"], lr=cfg.get("lr", 1.0),
                     dtype=np.float32, reuse_entry_eval=True, history=150)
    while not m.done:
        fc.set_params(regression_pose=frames["reg_pose"][i:i + 1], cam_translation=P1["cam_translation"], **_unflat(m.x_trial))
        f, gr = fc.closure(0)
        m.feed(f[0], gr[0])
    mac = np.array(m.records)
    k10 = np.flatnonzero(d150[:, 0] == 0)[9] + 1
    _compare(d150[:k10], mac[:k10], 1e-5, "first body stage, history 150, first ten line searches", whole_stage=False, t_rtol=1e-3)
    assert abs(d150[-1, 1] - mac[-1, 1]) <= 1e-3 * abs(mac[-1, 1]), (d150[-1], mac[-1])
    assert abs(d150[-1, 2] - mac[-1, 2]) <= 0.3 * mac[-1, 2], (d150[-1], mac[-1])
