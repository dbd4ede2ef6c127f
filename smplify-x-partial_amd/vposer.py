"""VPoser-v1 host side: checkpoint loading and the once-per-frame encoder.

The reference loads the external `human_body_prior` (cvpr19 branch) package:
`load_vposer(vposer_ckpt, vp_model='snapshot')` (smplifyx/fit_single_frame.py:241), decodes the
latent inside every closure evaluation (fitting.py:236-238 -- on the MI355X that is
csrc/vposer.h), and, with a regression prior, starts from
`vposer.encode(full_pose_prior).sample()` (fit_single_frame.py:245).  The encoder runs ONCE per
frame, outside the optimisation loop, so it lives here on the host (numpy); its architecture
follows the public VPoser-v1 definition:

    bn1 -> fc1 (in->512) -> leaky_relu(0.2) -> bn2 -> dropout (eval: off) -> fc2 (512->512)
        -> leaky_relu(0.2) -> Normal(mu = fc_mu, sigma = softplus(fc_logvar))

`.sample()` makes the reference non-deterministic; `encode` returns the mean by default and
draws the sample only when given a seeded generator.
"""
import glob
import os

import numpy as np

_DEC = (("fc1_w", "bodyprior_dec_fc1.weight"), ("fc1_b", "bodyprior_dec_fc1.bias"),
        ("fc2_w", "bodyprior_dec_fc2.weight"), ("fc2_b", "bodyprior_dec_fc2.bias"),
        ("out_w", "bodyprior_dec_out.weight"), ("out_b", "bodyprior_dec_out.bias"))
_ENC = (("enc_bn1_w", "bodyprior_enc_bn1.weight"), ("enc_bn1_b", "bodyprior_enc_bn1.bias"),
        ("enc_bn1_mean", "bodyprior_enc_bn1.running_mean"), ("enc_bn1_var", "bodyprior_enc_bn1.running_var"),
        ("enc_fc1_w", "bodyprior_enc_fc1.weight"), ("enc_fc1_b", "bodyprior_enc_fc1.bias"),
        ("enc_bn2_w", "bodyprior_enc_bn2.weight"), ("enc_bn2_b", "bodyprior_enc_bn2.bias"),
        ("enc_bn2_mean", "bodyprior_enc_bn2.running_mean"), ("enc_bn2_var", "bodyprior_enc_bn2.running_var"),
        ("enc_fc2_w", "bodyprior_enc_fc2.weight"), ("enc_fc2_b", "bodyprior_enc_fc2.bias"),
        ("enc_mu_w", "bodyprior_enc_mu.weight"), ("enc_mu_b", "bodyprior_enc_mu.bias"),
        ("enc_logvar_w", "bodyprior_enc_logvar.weight"), ("enc_logvar_b", "bodyprior_enc_logvar.bias"))
BN_EPS = 1e-5


def weights_from_state_dict(sd):
    """numpy weight dict (the keys engine.DeviceModel.set_vposer and encode() read) from a
    VPoser-v1 state dict (torch tensors or arrays, `bodyprior_*` names)."""
    def arr(v):
        return np.ascontiguousarray(v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v), np.float32)
    missing = [k for _, k in _DEC if k not in sd]
    if missing:
        raise KeyError("not a VPoser-v1 state dict, missing %s" % missing)
    w = {n: arr(sd[k]) for n, k in _DEC}
    if all(k in sd for _, k in _ENC):
        w.update({n: arr(sd[k]) for n, k in _ENC})
    return w


def load_vposer(vposer_ckpt):
    """Weights from `vposer_ckpt`: an .npz with the short names, a torch snapshot (.pt state
    dict), or a human_body_prior experiment directory (`snapshots/*.pt`, the newest is used, as
    `load_vposer(expr_dir, vp_model='snapshot')` does)."""
    path = os.path.expandvars(vposer_ckpt)
    if os.path.isdir(path):
        snaps = sorted(glob.glob(os.path.join(path, "snapshots", "*.pt")))
        if not snaps:
            raise ValueError("no snapshots/*.pt under %r" % path)
        path = snaps[-1]
    if path.endswith(".npz"):
        with np.load(path) as z:
            d = {k: z[k] for k in z.files}
        if "fc1_w" in d:
            return {k: np.ascontiguousarray(v, np.float32) for k, v in d.items()}
        return weights_from_state_dict(d)
    import torch
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    return weights_from_state_dict(sd)


def _leaky(x):
    return np.where(x > 0, x, 0.2 * x)


def _softplus(x):
    return np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0)


def _aa_to_matrot(pose):
    """[B, 21*3] axis-angle -> [B, 21*9] rotation matrices (plain Rodrigues, as the package's
    aa2matrot / torchgeometry angle_axis_to_rotation_matrix)."""
    aa = pose.reshape(-1, 3).astype(np.float64)
    ang = np.linalg.norm(aa, axis=1, keepdims=True)
    small = ang[:, 0] < 1e-6
    ax = aa / np.where(ang > 0, ang, 1.0)
    K = np.zeros((aa.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -ax[:, 2], ax[:, 1]
    K[:, 1, 0], K[:, 1, 2] = ax[:, 2], -ax[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -ax[:, 1], ax[:, 0]
    s, c = np.sin(ang)[:, :, None], np.cos(ang)[:, :, None]
    R = np.eye(3)[None] + s * K + (1 - c) * (K @ K)
    # first-order form near zero (torchgeometry's Taylor branch)
    Kt = np.zeros_like(K)
    Kt[:, 0, 1], Kt[:, 0, 2] = -aa[:, 2], aa[:, 1]
    Kt[:, 1, 0], Kt[:, 1, 2] = aa[:, 2], -aa[:, 0]
    Kt[:, 2, 0], Kt[:, 2, 1] = -aa[:, 1], aa[:, 0]
    R[small] = (np.eye(3)[None] + Kt)[small]
    return R.reshape(pose.shape[0], -1)


def encode(w, pose, generator=None):
    """VPoser-v1 encoder on [B, 63] body poses -> latent [B, latentD] (float32).

    Returns the mean of the posterior; with `generator` (numpy Generator) draws
    mean + sigma * N(0, 1), the reference's `.sample()` made reproducible.  An encoder whose
    first layer is 189 wide (rotation-matrix input, `data_shape [1, 21, 9]`) gets the poses as
    rotation matrices, a 63-wide one gets them as they are."""
    if "enc_fc1_w" not in w:
        raise ValueError("these VPoser weights carry no encoder (bodyprior_enc_*)")
    x = np.asarray(pose, np.float64).reshape(-1, 63)
    n_in = w["enc_fc1_w"].shape[1]
    if n_in == 189:
        x = _aa_to_matrot(x)
    elif n_in != 63:
        raise ValueError("unexpected VPoser encoder input width %d" % n_in)
    f = lambda k: w[k].astype(np.float64)
    x = (x - f("enc_bn1_mean")) / np.sqrt(f("enc_bn1_var") + BN_EPS) * f("enc_bn1_w") + f("enc_bn1_b")
    x = _leaky(x @ f("enc_fc1_w").T + f("enc_fc1_b"))
    x = (x - f("enc_bn2_mean")) / np.sqrt(f("enc_bn2_var") + BN_EPS) * f("enc_bn2_w") + f("enc_bn2_b")
    x = _leaky(x @ f("enc_fc2_w").T + f("enc_fc2_b"))
    mu = x @ f("enc_mu_w").T + f("enc_mu_b")
    if generator is not None:
        sigma = _softplus(x @ f("enc_logvar_w").T + f("enc_logvar_b"))
        mu = mu + sigma * generator.standard_normal(mu.shape)
    return mu.astype(np.float32)
