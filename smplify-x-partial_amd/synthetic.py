"""Seeded synthetic inputs: an SMPL-X-*shaped* body model and 2-D keypoint frames.

The licensed ``SMPLX_{NEUTRAL,MALE,FEMALE}.npz`` files cannot ship, so tests, the
benchmark and the smoke run use a deterministic stand-in with exactly the keys and
shapes ``smplx.SMPLX`` consumes (reference call site
``smplifyx/main.py:109-127``): V=10475 vertices, F=20908 faces, 55 joints with the
true SMPL-X parent table, 486 pose-blendshape rows, 10+10 shape/expression
directions, 45x45 hand PCA bases, 51 static + 79x17 dynamic face landmarks.
The geometry is a crude humanoid (vertices scattered around the bones of a T-pose
skeleton, <=4 skinning weights per vertex) so that fits behave like fits of a body.

Pure numpy (+scipy cKDTree); nothing here touches the GPU or the oracle.
"""
import numpy as np

NUM_JOINTS = 55
NUM_VERTS = 10475
NUM_FACES = 20908
NUM_POSE_BASIS = 9 * (NUM_JOINTS - 1)

# SMPL-X kinematic tree (kintree_table row 0; parents[0] = -1)
SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
     15, 15, 15, 20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int64)

# Vertex picks appended as joints 55..75 by smplx.VertexJointSelector for the real
# SMPL-X topology (nose, reye, leye, rear, lear, 6 feet, 5 left tips, 5 right tips)
SMPLX_EXTRA_VERTEX_IDS = np.array(
    [9120, 9929, 9448, 616, 6,
     5770, 5780, 8846, 8463, 8474, 8635,
     5361, 4933, 5058, 5169, 5286,
     8079, 7669, 7794, 7905, 8022], dtype=np.int64)


def _rest_skeleton():
    """Approximate SMPL-X T-pose joint positions [55,3] (metres, y up, +x = left)."""
    J = np.zeros((55, 3), np.float64)
    J[0] = (0.0, 0.0, 0.0)
    J[1] = (0.06, -0.09, 0.0);   J[2] = (-0.06, -0.09, 0.0)
    J[3] = (0.0, 0.11, -0.02)
    J[4] = (0.10, -0.47, 0.0);   J[5] = (-0.10, -0.47, 0.0)
    J[6] = (0.0, 0.25, 0.0)
    J[7] = (0.09, -0.87, -0.03); J[8] = (-0.09, -0.87, -0.03)
    J[9] = (0.0, 0.30, 0.02)
    J[10] = (0.12, -0.93, 0.09); J[11] = (-0.12, -0.93, 0.09)
    J[12] = (0.0, 0.51, -0.02)
    J[13] = (0.08, 0.42, -0.01); J[14] = (-0.08, 0.42, -0.01)
    J[15] = (0.0, 0.58, 0.02)
    J[16] = (0.19, 0.45, -0.02); J[17] = (-0.19, 0.45, -0.02)
    J[18] = (0.45, 0.44, -0.03); J[19] = (-0.45, 0.44, -0.03)
    J[20] = (0.70, 0.45, -0.03); J[21] = (-0.70, 0.45, -0.03)
    J[22] = (0.0, 0.56, 0.05)
    J[23] = (0.03, 0.63, 0.08);  J[24] = (-0.03, 0.63, 0.08)
    # fingers: order index, middle, pinky, ring, thumb; 3 joints each
    zf = [0.03, 0.01, -0.03, -0.01, 0.05]
    for side, base, wrist in ((1.0, 25, 20), (-1.0, 40, 21)):
        for f in range(5):
            for k in range(3):
                x = 0.09 + 0.03 * k if f < 4 else 0.03 + 0.025 * k
                y = 0.0 if f < 4 else -0.02
                J[base + 3 * f + k] = J[wrist] + (side * x, y, zf[f])
    return J


def _bone_radius(j):
    if j in (0, 3, 6, 9):
        return 0.10
    if j in (12, 15, 22, 23, 24):
        return 0.06
    if j in (1, 2, 4, 5):
        return 0.06
    if j >= 25:
        return 0.007
    return 0.04


def make_synthetic_model(seed=0, num_verts=NUM_VERTS, num_faces=NUM_FACES,
                         dtype=np.float32, surface=False):
    """Return a dict with the key set of an SMPL-X ``.npz`` (appendix A.1) plus
    ``extra_vertex_ids`` (the 21 vertex-joint picks valid for this geometry).
    Deterministic in ``seed``.

    surface=False (default, what every golden fixture is made with): vertices are scattered in the
    volume around the bones -- fine for LBS / keypoint work, but as a mesh it is a triangle soup that
    interpenetrates everywhere.  surface=True moves every vertex onto a tube around its bone
    (same random draws, offset projected off the bone axis and normalised to the tube radius): the
    faces, built from near neighbours, then form surface patches, which is what collision handling
    sees on a real body mesh; the blend shapes become smooth displacement fields for the same reason."""
    from scipy.spatial import cKDTree
    rng = np.random.RandomState(seed)
    V, F, J = num_verts, num_faces, NUM_JOINTS
    parents = SMPLX_PARENTS
    Jrest = _rest_skeleton()

    # ---- vertices scattered around bones -------------------------------------------
    share = np.ones(J)
    share[[0, 3, 6, 9]] = 6.0      # torso
    share[[12, 15]] = 4.0          # neck/head
    share[[22, 23, 24]] = 2.0      # face
    share[[1, 2, 4, 5, 7, 8]] = 3.0
    share[[16, 17, 18, 19, 20, 21]] = 2.5
    share[25:] = 0.6
    # vertices are numbered bone by bone (real meshes are spatially coherent too: neighbouring
    # vertex ids share their few skinning joints)
    bone_of = np.sort(rng.choice(J, size=V, p=share / share.sum()))
    t = rng.uniform(0.0, 1.0, size=V)
    v_template = np.zeros((V, 3))
    weights = np.zeros((V, J))
    for j in range(J):
        idx = np.nonzero(bone_of == j)[0]
        if idx.size == 0:
            continue
        p = parents[j]
        a = Jrest[p] if p >= 0 else Jrest[j] + np.array([0.0, -0.08, 0.0])
        b = Jrest[j]
        tt = t[idx][:, None]
        off = rng.normal(0, _bone_radius(j), (idx.size, 3))
        if surface:
            ax = (b - a) / max(np.linalg.norm(b - a), 1e-9)
            off = off - (off @ ax)[:, None] * ax
            off = off / np.maximum(np.linalg.norm(off, axis=1, keepdims=True), 1e-9) * (1.5 * _bone_radius(j))
        pts = a * (1 - tt) + b * tt + off
        v_template[idx] = pts
        # <=4 nonzero skinning weights: bone owner (parent joint), this joint,
        # grand-parent and one child-side neighbour, blended along the bone
        own = p if p >= 0 else j
        w_this = 0.15 + 0.7 * t[idx] ** 2
        weights[idx, j] += w_this
        weights[idx, own] += (1.0 - w_this) * 0.85
        gp = parents[own] if own >= 0 and parents[own] >= 0 else own
        weights[idx, gp] += (1.0 - w_this) * 0.15
    weights /= weights.sum(axis=1, keepdims=True)

    # ---- joint regressor: sparse, row-stochastic, local --------------------------------
    tree = cKDTree(v_template)
    J_regressor = np.zeros((J, V))
    for j in range(J):
        d, nn = tree.query(Jrest[j], k=24)
        w = np.exp(-(d / (d.mean() + 1e-9)) ** 2)
        J_regressor[j, nn] = w / w.sum()

    # ---- blend shapes ---------------------------------------------------------------
    shapedirs = 0.01 * rng.normal(size=(V, 3, 20))
    # make the first three shape directions smooth (height / girth / limb length)
    shapedirs[:, :, 0] += 0.03 * v_template * np.array([0.3, 1.0, 0.3])
    shapedirs[:, :, 1] += 0.03 * v_template * np.array([1.0, 0.1, 1.0])
    shapedirs[:, :, 2] += 0.02 * v_template * np.array([1.0, 0.0, 0.0])
    posedirs = 0.001 * rng.normal(size=(V, 3, NUM_POSE_BASIS))
    if surface:
        # per-vertex white noise would shred the surface patches as soon as betas != 0; use smooth
        # displacement fields (random plane waves, wavelength >= 0.4 m) of the same magnitude instead
        r2 = np.random.RandomState(seed + 1000)

        def smooth_field(n, amp):
            k = r2.normal(size=(n, 3, 3)) * (2 * np.pi / 0.6)            # [n, component, xyz]
            ph = r2.uniform(0, 2 * np.pi, size=(n, 3))
            return amp * np.sin(np.einsum("vx,ncx->vcn", v_template, k) + ph.T[None])
        shapedirs = smooth_field(20, 0.01 * np.sqrt(2.0))
        shapedirs[:, :, 0] += 0.03 * v_template * np.array([0.3, 1.0, 0.3])
        shapedirs[:, :, 1] += 0.03 * v_template * np.array([1.0, 0.1, 1.0])
        shapedirs[:, :, 2] += 0.02 * v_template * np.array([1.0, 0.0, 0.0])
        posedirs = smooth_field(NUM_POSE_BASIS, 0.001 * np.sqrt(2.0))

    # ---- faces: each vertex + two of its near neighbours ------------------------------
    _, nbr = tree.query(v_template, k=8)
    fa = rng.randint(0, V, size=F)
    pick = np.stack([rng.permutation(7)[:2] + 1 for _ in range(64)])
    pk = pick[rng.randint(0, 64, size=F)]
    faces = np.stack([fa, nbr[fa, pk[:, 0]], nbr[fa, pk[:, 1]]], axis=1)

    # ---- hands PCA ------------------------------------------------------------------
    ql, _ = np.linalg.qr(rng.normal(size=(45, 45)))
    qr_, _ = np.linalg.qr(rng.normal(size=(45, 45)))
    hands_meanl = 0.1 * rng.normal(size=45)
    hands_meanr = 0.1 * rng.normal(size=45)

    # ---- landmarks: faces whose first vertex lies in the head region -------------------
    head_c = Jrest[15] + np.array([0.0, 0.04, 0.05])
    dist_head = np.linalg.norm(v_template[faces[:, 0]] - head_c, axis=1)
    head_faces = np.argsort(dist_head)[:600]
    lmk_faces_idx = rng.choice(head_faces, size=51, replace=False)
    lmk_bary = rng.dirichlet(np.ones(3), size=51)
    dyn_faces = rng.choice(head_faces, size=(79, 17))
    dyn_bary = rng.dirichlet(np.ones(3), size=(79, 17))

    # ---- vertex joints (nose, eyes, ears, feet, finger tips) ---------------------------
    targets = [
        Jrest[15] + (0.0, 0.03, 0.13),                       # nose
        Jrest[24] + (0.0, 0.0, 0.02), Jrest[23] + (0.0, 0.0, 0.02),   # reye, leye
        Jrest[15] + (-0.08, 0.03, 0.0), Jrest[15] + (0.08, 0.03, 0.0),  # rear, lear
        Jrest[10] + (0.02, -0.02, 0.06), Jrest[10] + (0.06, -0.02, 0.04),
        Jrest[7] + (0.0, -0.06, -0.05),                      # LBigToe LSmallToe LHeel
        Jrest[11] + (-0.02, -0.02, 0.06), Jrest[11] + (-0.06, -0.02, 0.04),
        Jrest[8] + (0.0, -0.06, -0.05),                      # R toes, heel
    ]
    for base, side in ((25, 1.0), (40, -1.0)):               # tips: thumb,index,middle,ring,pinky
        for f in (4, 0, 1, 3, 2):
            targets.append(Jrest[base + 3 * f + 2] + (side * 0.025, 0.0, 0.0))
    extra = []
    used = set()
    for tg in targets:
        _, cand = tree.query(np.asarray(tg, np.float64), k=16)
        pickv = next(int(c) for c in cand if int(c) not in used)
        used.add(pickv)
        extra.append(pickv)

    model = dict(
        v_template=v_template.astype(dtype),
        f=faces.astype(np.uint32),
        shapedirs=shapedirs.astype(dtype),
        posedirs=posedirs.astype(dtype),
        J_regressor=J_regressor.astype(dtype),
        weights=weights.astype(dtype),
        kintree_table=np.stack([parents, np.arange(J)]).astype(np.int64),
        hands_componentsl=ql.T.astype(dtype), hands_componentsr=qr_.T.astype(dtype),
        hands_meanl=hands_meanl.astype(dtype), hands_meanr=hands_meanr.astype(dtype),
        lmk_faces_idx=lmk_faces_idx.astype(np.int64),
        lmk_bary_coords=lmk_bary.astype(dtype),
        dynamic_lmk_faces_idx=dyn_faces.astype(np.int64),
        dynamic_lmk_bary_coords=dyn_bary.astype(dtype),
        extra_vertex_ids=np.asarray(extra, np.int64),
    )
    model["kintree_table"][0, 0] = -1
    return model


def make_synthetic_parts(model):
    """Per-face body-part labels in the format of smplifyx/smplx_parts_segm.pkl
    (fit_single_frame.py:318-324): segm[f] = joint with the largest skinning weight on the face's
    first vertex, parents[f] = that joint's kinematic parent (-1 for the root)."""
    W = np.asarray(model["weights"])
    faces = np.asarray(model["f"]).astype(np.int64)
    par = np.asarray(model["kintree_table"])[0].astype(np.int64).copy()
    par[0] = -1
    segm = W[faces[:, 0]].argmax(1).astype(np.int64)
    return dict(segm=segm, parents=par[segm])


TOPOLOGY_FILE = "smplx_topology.npz"


def _topology_path(path=None):
    import os
    if path is None:
        path = os.environ.get("SFX_SMPLX_TOPOLOGY") or os.path.join(
            os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    if os.path.isdir(path):
        path = os.path.join(path, TOPOLOGY_FILE)
    return path


def topology_available(path=None):
    """Is the SMPL-X topology fixture there?  It is built locally from the reference tree's data files
    (tools/make_topology.py, run by __graft_entry__.build()) and never committed (MPG non-commercial licence)."""
    import os
    return os.path.isfile(_topology_path(path))


def load_topology(path=None):
    """The arrays of tests/golden/smplx_topology.npz (tools/make_topology.py: the SMPL-X face topology of the reference's
    demo .ply files, `segm` / `parents` of smplifyx/smplx_parts_segm.pkl, ExPose's posed body of demo frame 02) as a dict
    of int64 / float64 arrays.  path: the .npz, a directory holding it, or None = this working tree's copy
    (SFX_SMPLX_TOPOLOGY overrides).  The file is a LOCAL build product, not part of the repository's history."""
    path = _topology_path(path)
    try:
        g = np.load(path)
    except FileNotFoundError:
        raise FileNotFoundError("%s is not there: it is built from the reference tree by `python tools/make_topology.py` "
                                "(or __graft_entry__.build()) and is not committed (SMPL-X licence); without it use the "
                                "synthetic tube mesh (make_synthetic_model(surface=True), bench.py --mesh tubes)" % path)
    return dict(faces=g["faces"].astype(np.int64), segm=g["segm"].astype(np.int64), parents=g["parents"].astype(np.int64),
                vertices=g["vertices"].astype(np.float64), joints=g["joints"].astype(np.float64))


def make_topology_model(seed=0, topology=None, dtype=np.float32, smooth_rounds=10):
    """A body model on the REAL SMPL-X surface, as far as the reference tree holds it ("faces from the demo
    .ply topology"): `f` = the true 20 908 faces, `v_template` = ExPose's posed body of demo frame 02 (pelvis at the
    origin), rest joints = ExPose's first 55 joints, reproduced by a sparse non-negative row-stochastic `J_regressor`;
    skinning weights grown from the per-face part labels of smplx_parts_segm.pkl (a vertex starts with the labels of its
    faces; `smooth_rounds` rounds of averaging over mesh edges blend neighbouring parts; the 4 largest are kept --
    SMPL-X's own sparsity); the 21 vertex joints are SMPL-X's real vertex ids (they reproduce ExPose's joints 55..75
    exactly); blend shapes are smooth displacement fields (the licensed ones are absent) of the magnitudes SURVEY.md 8d
    names; landmarks are seeded picks among the head / jaw / eye faces.  This is the mesh the interpenetration term
    (fitting.py:437-455) is tested and benchmarked on: real triangles, real part structure, a real self-touching pose.
    Returns the model dict (same keys as make_synthetic_model); the part labels come from topology_parts()."""
    import scipy.sparse as sp
    from scipy.optimize import nnls
    from scipy.spatial import cKDTree
    tp = load_topology(topology) if not isinstance(topology, dict) else topology
    rng = np.random.RandomState(seed)
    faces, segm = tp["faces"], tp["segm"]
    J = NUM_JOINTS
    origin = tp["joints"][0]
    v_template = tp["vertices"] - origin
    Jrest = tp["joints"][:J] - origin
    V = v_template.shape[0]
    assert V == NUM_VERTS and faces.shape == (NUM_FACES, 3) and segm.max() < J

    # ---- skinning weights from the part labels -------------------------------------------------------
    W = np.zeros((V, J))
    for c in range(3):
        np.add.at(W, (faces[:, c], segm), 1.0)
    W /= W.sum(1, keepdims=True)
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    e = np.concatenate([e, e[:, ::-1]])
    A = sp.coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(V, V)).tocsr()
    A.data[:] = 1.0
    deg = np.asarray(A.sum(1)).ravel()
    for _ in range(smooth_rounds):
        W = 0.5 * W + 0.5 * (A @ W) / deg[:, None]
    top = np.argsort(-W, 1)[:, :4]
    weights = np.zeros_like(W)
    np.put_along_axis(weights, top, np.take_along_axis(W, top, 1), 1)
    weights[weights < 0.01] = 0.0
    weights /= weights.sum(1, keepdims=True)

    # ---- joint regressor: half a local Gaussian average, half a non-negative correction that lands on the joint ----
    tree = cKDTree(v_template)
    J_regressor = np.zeros((J, V))
    for j in range(J):
        for k, alpha in ((400, 0.5), (400, 1.0), (2000, 0.5), (2000, 1.0), (V, 1.0)):
            d, nb = tree.query(Jrest[j], k=k)
            w0 = np.exp(-(d / (d[:24].mean() + 1e-9)) ** 2)
            w0 /= w0.sum()
            pts = v_template[nb]
            target = (Jrest[j] - (1.0 - alpha) * (w0 @ pts)) / alpha
            w1, _ = nnls(np.vstack([pts.T, 10.0 * np.ones((1, len(nb)))]), np.concatenate([target, [10.0]]))
            w = (1.0 - alpha) * w0 + alpha * w1
            if np.abs(w @ pts - Jrest[j]).max() < 1e-9 and abs(w.sum() - 1.0) < 1e-9:
                break
        else:
            raise RuntimeError("joint %d is not inside the hull of the vertices" % j)
        J_regressor[j, nb] = w / w.sum()

    # ---- blend shapes: smooth fields (random plane waves, wavelength >= 0.4 m) ------------------------
    r2 = np.random.RandomState(seed + 1000)

    def smooth_field(n, amp):
        k = r2.normal(size=(n, 3, 3)) * (2 * np.pi / 0.6)
        ph = r2.uniform(0, 2 * np.pi, size=(n, 3))
        return amp * np.sin(np.einsum("vx,ncx->vcn", v_template, k) + ph.T[None])
    shapedirs = smooth_field(20, 0.01 * np.sqrt(2.0))
    shapedirs[:, :, 0] += 0.03 * v_template * np.array([0.3, 1.0, 0.3])
    shapedirs[:, :, 1] += 0.03 * v_template * np.array([1.0, 0.1, 1.0])
    shapedirs[:, :, 2] += 0.02 * v_template * np.array([1.0, 0.0, 0.0])
    posedirs = smooth_field(NUM_POSE_BASIS, 0.001 * np.sqrt(2.0))

    ql, _ = np.linalg.qr(rng.normal(size=(45, 45)))
    qr_, _ = np.linalg.qr(rng.normal(size=(45, 45)))
    hands_meanl = 0.1 * rng.normal(size=45)
    hands_meanr = 0.1 * rng.normal(size=45)

    head_faces = np.nonzero(np.isin(segm, (15, 22, 23, 24)))[0]
    lmk_faces_idx = rng.choice(head_faces, size=51, replace=False)
    lmk_bary = rng.dirichlet(np.ones(3), size=51)
    dyn_faces = rng.choice(head_faces, size=(79, 17))
    dyn_bary = rng.dirichlet(np.ones(3), size=(79, 17))

    parents = SMPLX_PARENTS
    model = dict(
        v_template=v_template.astype(dtype), f=faces.astype(np.uint32),
        shapedirs=shapedirs.astype(dtype), posedirs=posedirs.astype(dtype),
        J_regressor=J_regressor.astype(dtype), weights=weights.astype(dtype),
        kintree_table=np.stack([parents, np.arange(J)]).astype(np.int64),
        hands_componentsl=ql.T.astype(dtype), hands_componentsr=qr_.T.astype(dtype),
        hands_meanl=hands_meanl.astype(dtype), hands_meanr=hands_meanr.astype(dtype),
        lmk_faces_idx=lmk_faces_idx.astype(np.int64), lmk_bary_coords=lmk_bary.astype(dtype),
        dynamic_lmk_faces_idx=dyn_faces.astype(np.int64), dynamic_lmk_bary_coords=dyn_bary.astype(dtype),
        extra_vertex_ids=SMPLX_EXTRA_VERTEX_IDS.copy(),
    )
    model["kintree_table"][0, 0] = -1
    return model


def topology_parts(topology=None):
    """`segm` / `parents` per face exactly as smplifyx/smplx_parts_segm.pkl holds them (fit_single_frame.py:317-324)."""
    tp = load_topology(topology) if not isinstance(topology, dict) else topology
    return dict(segm=tp["segm"].copy(), parents=tp["parents"].copy())


def make_synthetic_gmm(seed=0, num_gaussians=8, dim=63):
    """Synthetic stand-in for SMPLify's gmm_08.pkl (not shipped with the reference; 69-D for SMPL):
    dict(means [M,dim], covars [M,dim,dim] SPD, weights [M]) with pose-like scales (means ~0.15 rad,
    standard deviations 0.05-0.4 rad, correlated)."""
    rng = np.random.RandomState(2000 + seed)
    means = 0.15 * rng.normal(size=(num_gaussians, dim))
    covars = []
    for m in range(num_gaussians):
        q, _ = np.linalg.qr(rng.normal(size=(dim, dim)))
        sd = np.exp(rng.uniform(np.log(0.05), np.log(0.4), size=dim))
        covars.append((q * sd ** 2) @ q.T)
    weights = rng.dirichlet(3.0 * np.ones(num_gaussians))
    return dict(means=means, covars=np.stack(covars), weights=weights)


def make_synthetic_vposer(seed=0, latent=32, hidden=512, dtype=np.float32, encoder_inputs=0):
    """Random-init VPoser-v1 *decoder* weights (appendix A.3): fc1 32->512,
    fc2 512->512, out 512->126, leaky_relu(0.2).  Scaled so decoded poses are O(0.3 rad).
    encoder_inputs = 63 or 189 adds random encoder weights (bn1, fc1, bn2, fc2, mu, logvar)."""
    rng = np.random.RandomState(1000 + seed)
    def lin(i, o, s):
        return (rng.normal(size=(o, i)) * s / np.sqrt(i)).astype(dtype), \
               (0.01 * rng.normal(size=o)).astype(dtype)
    w1, b1 = lin(latent, hidden, 1.0)
    w2, b2 = lin(hidden, hidden, 1.0)
    w3, b3 = lin(hidden, 126, 0.4)
    # bias the 6-D output towards identity so z = 0 decodes near the rest pose
    ident6 = np.tile(np.array([1, 0, 0, 1, 0, 0], dtype), 21)   # view(-1,3,2): cols (1,0,0),(0,1,0)
    b3 = b3 + ident6
    out = dict(fc1_w=w1, fc1_b=b1, fc2_w=w2, fc2_b=b2, out_w=w3, out_b=b3)
    if encoder_inputs:
        def bn(n):
            return ((1 + 0.1 * rng.normal(size=n)).astype(dtype), (0.05 * rng.normal(size=n)).astype(dtype),
                    (0.1 * rng.normal(size=n)).astype(dtype), (0.5 + rng.uniform(size=n)).astype(dtype))
        out["enc_bn1_w"], out["enc_bn1_b"], out["enc_bn1_mean"], out["enc_bn1_var"] = bn(encoder_inputs)
        out["enc_fc1_w"], out["enc_fc1_b"] = lin(encoder_inputs, hidden, 1.0)
        out["enc_bn2_w"], out["enc_bn2_b"], out["enc_bn2_mean"], out["enc_bn2_var"] = bn(hidden)
        out["enc_fc2_w"], out["enc_fc2_b"] = lin(hidden, hidden, 1.0)
        out["enc_mu_w"], out["enc_mu_b"] = lin(hidden, latent, 0.5)
        out["enc_logvar_w"], out["enc_logvar_b"] = lin(hidden, latent, 0.5)
    return out


def rodrigues_np(theta):
    """Plain (eps-free) Rodrigues for data generation, [..,3] -> [..,3,3]."""
    theta = np.asarray(theta, np.float64)
    ang = np.linalg.norm(theta, axis=-1, keepdims=True)
    ax = theta / np.maximum(ang, 1e-12)
    K = np.zeros(theta.shape[:-1] + (3, 3))
    K[..., 0, 1] = -ax[..., 2]; K[..., 0, 2] = ax[..., 1]
    K[..., 1, 0] = ax[..., 2];  K[..., 1, 2] = -ax[..., 0]
    K[..., 2, 0] = -ax[..., 1]; K[..., 2, 1] = ax[..., 0]
    s = np.sin(ang)[..., None]; c = np.cos(ang)[..., None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def make_frame_truth(index, seed=0):
    """Ground-truth parameters of synthetic frame ``index``: per-frame
    RandomState(seed*1_000_003 + index) so any shard regenerates identical frames."""
    rng = np.random.RandomState((seed * 1000003 + index) % (2 ** 31 - 1))
    truth = dict(
        body_pose=0.15 * rng.normal(size=63),
        global_orient=0.2 * rng.normal(size=3),
        betas=rng.normal(size=10),
        cam_t=np.array([rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(12., 24.)]),
        prior_noise=0.05 * rng.normal(size=63),
        prior_noise_go=0.05 * rng.normal(size=3),
        kp_noise=rng.normal(size=(144, 2)),
        conf=rng.uniform(0.3, 1.0, size=144),
        conf_drop=rng.uniform(size=144) < 0.10,
    )
    return truth


def make_frames(n, joints_fn, K, start=0, seed=0, focal=5000.0, H=600, W=800, min_camera_keypoints=None,
                camera_keypoints=(9, 12, 2, 5)):
    """Synthetic 2-D keypoint frames `start .. start+n-1`.

    min_camera_keypoints: when given, the synthetic detector never loses more than len(camera_keypoints) - that many of
    the camera-initialisation keypoints (cfg init_joints_idxs: shoulders and hips): the keypoints it would have dropped
    beyond that are kept, highest index first.  bench.py asks for 3 of 4: a frame with two of them missing has an
    under-determined camera -- the reference's own fp32 and fp64 runs end such a frame in different basins, with 2-3 x the
    closure evaluations of a well-posed one -- and one such frame in a rank's share decides that rank's time.

    joints_fn(params) -> [n,K,3] mapped model joints for dict `params` of [n,.] float32
    arrays (global_orient, body_pose, betas; everything else zero) -- the caller supplies
    the LBS forward (the HIP engine on the GPU box, the oracle in CPU tests).
    Keypoints = perspective projection of the model's own joints + 1 px noise;
    confidences U(0.3,1) with 10 % dropped to (0,0,0); "regression prior" = true pose +
    0.05 rad noise pushed through the rotmat -> xyz-euler path of the reference.
    """
    from .utils import euler_xyz_from_matrix
    tr = [make_frame_truth(start + i, seed) for i in range(n)]
    P = dict(global_orient=np.stack([t["global_orient"] for t in tr]).astype(np.float32),
             body_pose=np.stack([t["body_pose"] for t in tr]).astype(np.float32),
             betas=np.stack([t["betas"] for t in tr]).astype(np.float32))
    j3 = np.asarray(joints_fn(P), np.float64)                       # [n,K,3]
    cam_t = np.stack([t["cam_t"] for t in tr])
    pc = j3 + cam_t[:, None, :]
    uv = focal * pc[..., :2] / pc[..., 2:3] + np.array([W * 0.5, H * 0.5])
    noise = np.stack([t["kp_noise"][:K] for t in tr])
    conf = np.stack([t["conf"][:K] for t in tr])
    drop = np.stack([t["conf_drop"][:K] for t in tr])
    if min_camera_keypoints is not None:
        ck = [int(j) for j in camera_keypoints if int(j) < K]
        for i in range(n):
            lost = [j for j in sorted(ck) if drop[i, j]]
            while len(ck) - len(lost) < min(int(min_camera_keypoints), len(ck)):
                drop[i, lost.pop()] = False
    kp = np.concatenate([uv + noise, conf[..., None]], -1)
    kp[drop] = 0.0
    reg_pose = np.stack([euler_xyz_from_matrix(rodrigues_np((t["body_pose"] + t["prior_noise"]).reshape(21, 3))).reshape(-1)
                         for t in tr])
    reg_glob = np.stack([euler_xyz_from_matrix(rodrigues_np(t["global_orient"] + t["prior_noise_go"]))
                         for t in tr])
    return dict(keypoints=kp.astype(np.float32), reg_pose=reg_pose.astype(np.float32),
                reg_global=reg_glob.astype(np.float32), truth=P, cam_t=cam_t.astype(np.float32),
                H=H, W=W, focal=float(focal))
