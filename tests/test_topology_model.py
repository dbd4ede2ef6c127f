"""CPU: the SMPL-X topology fixture (tests/golden/smplx_topology.npz, made LOCALLY by tools/make_topology.py -- never committed,
SMPL-X licence -- from the reference's demo .ply files, smplifyx/smplx_parts_segm.pkl and ExPose's demo result) and the body model
smplifyx_amd.synthetic.make_topology_model builds on it -- the mesh the interpenetration term (fitting.py:437-455,
fit_single_frame.py:300-328) is tested and benchmarked on."""
import numpy as np
import pytest
import torch

from oracle import penetration as OP
from smplifyx_amd import synthetic

pytestmark = pytest.mark.skipif(not synthetic.topology_available(), reason="smplx_topology.npz not built (tools/make_topology.py)")

IGN = ["9,16", "9,17", "6,16", "6,17", "1,2", "12,22"]          # cfg_files/fit_smplx_combined_halpe.yaml: ign_part_pairs


@pytest.fixture(scope="module")
def topo():
    return synthetic.load_topology()


@pytest.fixture(scope="module")
def model():
    return synthetic.make_topology_model(0)


def test_fixture_is_the_smplx_surface(topo):
    f, v = topo["faces"], topo["vertices"]
    assert f.shape == (20908, 3) and v.shape == (10475, 3) and topo["joints"].shape == (144, 3)
    assert f.min() == 0 and f.max() == 10474 and np.unique(f).size == 10475
    # an orientable surface, closed but for the eye sockets / mouth the SMPL-X mesh leaves open: every edge belongs to one or two
    # faces, and two faces never run through a shared edge in the same direction
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    und = np.sort(e, 1)
    _, cnt = np.unique(und, axis=0, return_counts=True)
    assert cnt.max() == 2 and (cnt == 2).mean() > 0.99
    _, dcnt = np.unique(e, axis=0, return_counts=True)
    assert dcnt.max() == 1
    # the 21 vertex joints of smplx.VertexJointSelector are ExPose's joints 55..75 bit for bit: the ids belong to this topology
    assert np.array_equal(v[synthetic.SMPLX_EXTRA_VERTEX_IDS].astype(np.float32), topo["joints"][55:76].astype(np.float32))
    # part labels: the 55 joints, parent table = the kinematic tree
    assert topo["segm"].shape == (20908,) and set(np.unique(topo["segm"])) == set(range(55))
    assert np.array_equal(topo["parents"], synthetic.SMPLX_PARENTS[topo["segm"]])


def test_topology_model_is_a_valid_smplx_file(model, topo):
    keys = set(synthetic.make_synthetic_model(0).keys())
    assert set(model.keys()) == keys
    W = model["weights"].astype(np.float64)
    assert np.allclose(W.sum(1), 1.0, atol=1e-6) and (W >= 0).all() and (W > 0).sum(1).max() <= 4
    # the labels of a vertex' faces carry most of its weight
    seg_v = np.zeros(len(W), np.int64)
    for c in range(3):
        seg_v[topo["faces"][:, c]] = topo["segm"]
    assert (W[np.arange(len(W)), seg_v] > 0).mean() > 0.99
    Jr = model["J_regressor"].astype(np.float64)
    assert (Jr >= 0).all() and np.allclose(Jr.sum(1), 1.0, atol=1e-5)
    rest = Jr @ model["v_template"].astype(np.float64)
    assert np.abs(rest - (topo["joints"][:55] - topo["joints"][0])).max() < 1e-6        # ExPose's joints, pelvis at the origin
    assert np.array_equal(model["f"].astype(np.int64), topo["faces"])
    assert np.array_equal(model["extra_vertex_ids"], synthetic.SMPLX_EXTRA_VERTEX_IDS)
    again = synthetic.make_topology_model(0)
    assert all(np.array_equal(model[k], again[k]) for k in keys)                        # deterministic


def test_oracle_forward_at_rest_is_the_expose_body(model, topo):
    import helpers as H
    from oracle.body_model import SMPLXRef
    cfg = H.load_cfg("fit_smplx_combined_halpe.yaml")
    jm = H.joint_map_for(cfg)
    bm = SMPLXRef(model, joint_map=jm, flat_hand_mean=True, dtype=torch.float64)
    bm.reset_params()
    with torch.no_grad():
        o = bm(return_verts=True, body_pose=torch.zeros([1, 63], dtype=torch.float64))
    # zero pose, zero shape, flat hands: the template = ExPose's body of demo frame 02, and its joints
    assert np.abs(o.vertices[0].numpy() - model["v_template"]).max() < 1e-6
    expose = (topo["joints"] - topo["joints"][0])[np.asarray(jm)[np.asarray(jm) < 76]]
    assert np.abs(o.joints[0].numpy()[np.asarray(jm) < 76] - expose).max() < 1e-6      # (landmarks beyond 76 are seeded picks)
    assert o.joints.shape == (1, len(jm), 3)


def test_candidate_pairs_on_the_real_surface(model):
    """The reference's own mesh in ExPose's pose of demo frame 02, its own part table and the cfg's ign_part_pairs: 13 729
    box-overlapping pairs without a shared vertex, 850 of them between parts that may collide (touching fingers, mostly), at
    most 15 partners per triangle -- nowhere near max_collisions 128.  The x-sweep of the oracle only prunes."""
    parts = synthetic.topology_parts()
    v, f = model["v_template"], model["f"].astype(np.int64)
    allp = OP.candidate_pairs(v, f)
    pairs = OP.candidate_pairs(v, f, parts["segm"], parts["parents"], IGN)
    assert len(allp) == 13729 and len(pairs) == 850
    cnt = np.bincount(pairs.reshape(-1), minlength=len(f))
    assert cnt.max() == 15 and (cnt > 0).sum() == 410
    brute = OP.candidate_pairs(v, f, parts["segm"], parts["parents"], IGN, sweep=False)
    assert np.array_equal(pairs, brute)
    lo, g, _ = OP.penetration(v, f, parts["segm"], parts["parents"], IGN, sigma=1e-4)
    assert lo > 0 and np.isfinite(g).all() and (np.abs(g).sum(1) > 0).sum() > 100
