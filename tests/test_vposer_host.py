"""Host-side VPoser pieces (checkpoint loading, encoder) against the oracle's torch-module
restatement of the package (oracle/vposer.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.vposer import VPoserEncoderRef
from smplifyx_amd import synthetic, vposer


def _rotmats(pose):
    from scipy.spatial.transform import Rotation as Rot
    return Rot.from_rotvec(pose.reshape(-1, 3)).as_matrix().reshape(pose.shape[0], -1)


@pytest.mark.parametrize("n_in", [63, 189])
def test_encode_matches_torch_modules(n_in):
    w = synthetic.make_synthetic_vposer(0, encoder_inputs=n_in)
    pose = 0.3 * np.random.RandomState(3).normal(size=(5, 63))
    pose[0] = 0                                              # zero rotation: the Taylor branch
    ref = VPoserEncoderRef(w)
    x = torch.tensor(pose if n_in == 63 else _rotmats(pose), dtype=torch.float64)
    dist = ref.encode(x)
    z = vposer.encode(w, pose)
    assert z.dtype == np.float32 and z.shape == (5, 32)
    assert np.allclose(z, dist.mean.detach().numpy(), rtol=1e-5, atol=1e-6)
    # the seeded sample is mean + sigma * N(0,1) with the same sigma
    g = np.random.default_rng(7)
    zs = vposer.encode(w, pose, generator=g)
    eps = np.random.default_rng(7).standard_normal((5, 32))
    assert np.allclose(zs, (dist.mean + dist.stddev * torch.tensor(eps)).detach().numpy(), rtol=1e-5, atol=1e-6)


def test_load_vposer_from_state_dict_and_dir(tmp_path):
    w = synthetic.make_synthetic_vposer(1, encoder_inputs=63)
    ref = VPoserEncoderRef(w, dtype=torch.float32)
    sd = dict(ref.state_dict())
    for short, full in (("fc1", "bodyprior_dec_fc1"), ("fc2", "bodyprior_dec_fc2"), ("out", "bodyprior_dec_out")):
        sd[full + ".weight"] = torch.tensor(w[short + "_w"]); sd[full + ".bias"] = torch.tensor(w[short + "_b"])
    os.makedirs(tmp_path / "snapshots")
    torch.save(sd, tmp_path / "snapshots" / "TR00_E096.pt")
    got = vposer.load_vposer(str(tmp_path))                  # experiment directory, as load_vposer(expr_dir)
    for k in w:
        assert np.array_equal(got[k], w[k]), k
    np.savez(tmp_path / "w.npz", **w)
    got2 = vposer.load_vposer(str(tmp_path / "w.npz"))
    assert all(np.array_equal(got2[k], w[k]) for k in w)
    with pytest.raises(KeyError):
        vposer.weights_from_state_dict({"foo": np.zeros(3)})
    with pytest.raises(ValueError):
        vposer.encode(synthetic.make_synthetic_vposer(0), np.zeros((1, 63)))   # decoder-only weights
