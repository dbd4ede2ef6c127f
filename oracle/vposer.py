"""ORACLE (test infrastructure): VPoser-v1 decoder as the reference uses it
(`vposer.decode(pose_embedding, output_type='aa')`, call sites smplifyx/fit_single_frame.py:
241-245,265,515,607,654 and smplifyx/fitting.py:72,197,236).

PARITY UNPINNED: `human_body_prior` (branch cvpr19) and `torchgeometry` 0.1.2 are absent from
/root/reference and from this image; the algorithm follows SURVEY.md appendix A.3:
  fc1 32->512, leaky_relu(0.2), [dropout = identity in eval], fc2 512->512, leaky_relu(0.2),
  out 512->126, view(-1,3,2), Gram-Schmidt (ContinousRotReprDecoder), rotation matrix ->
  quaternion (torchgeometry.rotation_matrix_to_quaternion, eps 1e-6, on R^T) -> angle-axis.
Cross-checked in tests against scipy's Rotation.as_rotvec.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def rotmat_to_aa(R, eps=1e-6):
    """[N,3,3] -> [N,3]; torchgeometry 0.1.2 rotation_matrix_to_angle_axis semantics."""
    rt = R.transpose(1, 2)
    d2 = rt[:, 2, 2] < eps
    d0d1 = rt[:, 0, 0] > rt[:, 1, 1]
    d0nd1 = rt[:, 0, 0] < -rt[:, 1, 1]
    t0 = 1 + rt[:, 0, 0] - rt[:, 1, 1] - rt[:, 2, 2]
    q0 = torch.stack([rt[:, 1, 2] - rt[:, 2, 1], t0, rt[:, 0, 1] + rt[:, 1, 0], rt[:, 2, 0] + rt[:, 0, 2]], -1)
    t1 = 1 - rt[:, 0, 0] + rt[:, 1, 1] - rt[:, 2, 2]
    q1 = torch.stack([rt[:, 2, 0] - rt[:, 0, 2], rt[:, 0, 1] + rt[:, 1, 0], t1, rt[:, 1, 2] + rt[:, 2, 1]], -1)
    t2 = 1 - rt[:, 0, 0] - rt[:, 1, 1] + rt[:, 2, 2]
    q2 = torch.stack([rt[:, 0, 1] - rt[:, 1, 0], rt[:, 2, 0] + rt[:, 0, 2], rt[:, 1, 2] + rt[:, 2, 1], t2], -1)
    t3 = 1 + rt[:, 0, 0] + rt[:, 1, 1] + rt[:, 2, 2]
    q3 = torch.stack([t3, rt[:, 1, 2] - rt[:, 2, 1], rt[:, 2, 0] - rt[:, 0, 2], rt[:, 0, 1] - rt[:, 1, 0]], -1)
    c0 = (d2 & d0d1).to(R.dtype).unsqueeze(-1)
    c1 = (d2 & ~d0d1).to(R.dtype).unsqueeze(-1)
    c2 = (~d2 & d0nd1).to(R.dtype).unsqueeze(-1)
    c3 = (~d2 & ~d0nd1).to(R.dtype).unsqueeze(-1)
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0.unsqueeze(-1) * c0 + t1.unsqueeze(-1) * c1 + t2.unsqueeze(-1) * c2 + t3.unsqueeze(-1) * c3)
    q = q * 0.5                                           # (w, x, y, z)
    x, y, z, w = q[:, 1], q[:, 2], q[:, 3], q[:, 0]
    s2 = x * x + y * y + z * z
    s = torch.sqrt(s2)
    two_theta = 2.0 * torch.where(w < 0.0, torch.atan2(-s, -w), torch.atan2(s, w))
    k = torch.where(s2 > 0.0, two_theta / s, 2.0 * torch.ones_like(s))
    return torch.stack([x * k, y * k, z * k], -1)


class VPoserRef(nn.Module):
    def __init__(self, w, dtype=torch.float32):
        super().__init__()
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
        for k in ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "out_w", "out_b"):
            self.register_buffer(k, t(w[k]))
        self.latentD = self.fc1_w.shape[1]

    def decode(self, z, output_type="aa"):
        assert output_type == "aa"
        B = z.shape[0]
        h = F.leaky_relu(F.linear(z, self.fc1_w, self.fc1_b), negative_slope=0.2)
        h = F.leaky_relu(F.linear(h, self.fc2_w, self.fc2_b), negative_slope=0.2)
        o = F.linear(h, self.out_w, self.out_b).view(-1, 3, 2)
        b1 = F.normalize(o[:, :, 0], dim=1)
        dot = torch.sum(b1 * o[:, :, 1], dim=1, keepdim=True)
        b2 = F.normalize(o[:, :, 1] - dot * b1, dim=-1)
        b3 = torch.cross(b1, b2, dim=1)
        R = torch.stack([b1, b2, b3], dim=-1)
        return rotmat_to_aa(R).view(B, 1, -1, 3)


class VPoserEncoderRef(nn.Module):
    """Encoder of VPoser-v1 built from torch modules the way the package defines it
    (`bodyprior_enc_bn1/fc1/bn2/fc2/mu/logvar`); `encode` returns the
    torch.distributions.Normal the reference samples from (fit_single_frame.py:245)."""

    def __init__(self, w, dtype=torch.float64):
        super().__init__()
        n_in, hid = w["enc_fc1_w"].shape[1], w["enc_fc1_w"].shape[0]
        lat = w["enc_mu_w"].shape[0]
        self.bodyprior_enc_bn1 = nn.BatchNorm1d(n_in)
        self.bodyprior_enc_fc1 = nn.Linear(n_in, hid)
        self.bodyprior_enc_bn2 = nn.BatchNorm1d(hid)
        self.bodyprior_enc_fc2 = nn.Linear(hid, hid)
        self.bodyprior_enc_mu = nn.Linear(hid, lat)
        self.bodyprior_enc_logvar = nn.Linear(hid, lat)
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
        sd = {}
        for short, mod in (("enc_bn1", "bodyprior_enc_bn1"), ("enc_bn2", "bodyprior_enc_bn2")):
            sd[mod + ".weight"], sd[mod + ".bias"] = t(w[short + "_w"]), t(w[short + "_b"])
            sd[mod + ".running_mean"], sd[mod + ".running_var"] = t(w[short + "_mean"]), t(w[short + "_var"])
            sd[mod + ".num_batches_tracked"] = torch.tensor(0)
        for short, mod in (("enc_fc1", "bodyprior_enc_fc1"), ("enc_fc2", "bodyprior_enc_fc2"),
                           ("enc_mu", "bodyprior_enc_mu"), ("enc_logvar", "bodyprior_enc_logvar")):
            sd[mod + ".weight"], sd[mod + ".bias"] = t(w[short + "_w"]), t(w[short + "_b"])
        self.to(dtype)
        self.load_state_dict(sd)
        self.eval()

    def encode(self, x):
        x = x.view(x.size(0), -1)
        x = self.bodyprior_enc_bn1(x)
        x = F.leaky_relu(self.bodyprior_enc_fc1(x), negative_slope=0.2)
        x = self.bodyprior_enc_bn2(x)
        x = F.leaky_relu(self.bodyprior_enc_fc2(x), negative_slope=0.2)
        return torch.distributions.normal.Normal(self.bodyprior_enc_mu(x), F.softplus(self.bodyprior_enc_logvar(x)))
