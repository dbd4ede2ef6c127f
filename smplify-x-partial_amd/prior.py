"""Drop-in for smplifyx/prior.py:36-231: `create_prior(prior_type in {'gmm', 'l2', 'angle', 'none',
None})`.  Inside the fitting loop these terms are evaluated by the HIP closure kernel
(csrc/closure_body.h; the mixture prior's buffers go to the device through `sfx_batch_set_gmm`); the
modules below hold the constants and give the same numbers stand-alone."""
import os
import pickle

import numpy as np
import torch
import torch.nn as nn

DEFAULT_DTYPE = torch.float32


def create_prior(prior_type, **kwargs):
    if prior_type == "gmm":
        return MaxMixturePrior(**kwargs)
    if prior_type == "l2":
        return L2Prior(**kwargs)
    if prior_type == "angle":
        return SMPLifyAnglePrior(**kwargs)
    if prior_type == "none" or prior_type is None:
        def no_prior(*args, **kwargs):
            return 0.0
        return no_prior
    raise ValueError("Prior {}".format(prior_type) + " is not implemented")


class SMPLifyAnglePrior(nn.Module):
    """exp(pose[idx] * sign)^2 on left/right elbow and knee bending (prior.py:53-89)."""

    def __init__(self, dtype=torch.float32, **kwargs):
        super().__init__()
        self.register_buffer("angle_prior_idxs", torch.tensor([55, 58, 12, 15], dtype=torch.long))
        self.register_buffer("angle_prior_signs", torch.tensor([1, -1, -1, -1], dtype=dtype))

    def forward(self, pose, with_global_pose=False):
        idx = self.angle_prior_idxs - (not with_global_pose) * 3
        return torch.exp(pose[:, idx] * self.angle_prior_signs).pow(2)


class L2Prior(nn.Module):
    def __init__(self, dtype=DEFAULT_DTYPE, reduction="sum", **kwargs):
        super().__init__()

    def forward(self, module_input, *args):
        return torch.sum(module_input.pow(2))


class MaxMixturePrior(nn.Module):
    """Gaussian-mixture pose prior of SMPLify (prior.py:100-231): reads `gmm_{num_gaussians:02d}.pkl`
    (dict with means / covars / weights, or a pickled sklearn GMM) from `prior_folder`, keeps the
    reference's buffers (means, covs, precisions, nll_weights, weights, pi_term, cov_dets) and evaluates
    min_m [ 0.5 d_m^T P_m d_m - log nll_weights_m ] (`use_merged`, the reference's default) or the
    per-component form.  `gmm` may be passed as a dict directly instead of a file."""

    def __init__(self, prior_folder="prior", num_gaussians=6, dtype=DEFAULT_DTYPE, epsilon=1e-16, use_merged=True,
                 gmm=None, **kwargs):
        super().__init__()
        if dtype == DEFAULT_DTYPE:
            np_dtype = np.float32
        elif dtype == torch.float64:
            np_dtype = np.float64
        else:
            raise ValueError("Unknown float type {}".format(dtype))
        self.num_gaussians, self.epsilon, self.use_merged = num_gaussians, epsilon, use_merged
        if gmm is None:
            full_gmm_fn = os.path.join(prior_folder, "gmm_{:02d}.pkl".format(num_gaussians))
            if not os.path.exists(full_gmm_fn):
                raise FileNotFoundError('The path to the mixture prior "{}" does not exist'.format(full_gmm_fn))
            with open(full_gmm_fn, "rb") as f:
                gmm = pickle.load(f, encoding="latin1")
        if isinstance(gmm, dict):
            means, covs, weights = gmm["means"], gmm["covars"], gmm["weights"]
        elif "sklearn.mixture.gmm.GMM" in str(type(gmm)):
            means, covs, weights = gmm.means_, gmm.covars_, gmm.weights_
        else:
            raise ValueError("Unknown type for the prior: {}".format(type(gmm)))
        covs_in, weights_in = np.asarray(covs), np.asarray(weights)
        means, covs = np.asarray(means).astype(np_dtype), covs_in.astype(np_dtype)
        self.register_buffer("means", torch.tensor(means, dtype=dtype))
        self.register_buffer("covs", torch.tensor(covs, dtype=dtype))
        precisions = np.stack([np.linalg.inv(cov) for cov in covs]).astype(np_dtype)
        self.register_buffer("precisions", torch.tensor(precisions, dtype=dtype))
        # the constant term keeps the reference's literal 69 (the SMPL body pose size), prior.py:157
        sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in covs_in])
        const = (2 * np.pi) ** (69 / 2.)
        nll_weights = np.asarray(weights_in / (const * (sqrdets / sqrdets.min())))
        self.register_buffer("nll_weights", torch.tensor(nll_weights, dtype=dtype).unsqueeze(dim=0))
        self.register_buffer("weights", torch.tensor(weights_in, dtype=dtype).unsqueeze(dim=0))
        self.register_buffer("pi_term", torch.log(torch.tensor(2 * np.pi, dtype=dtype)))
        cov_dets = [np.log(np.linalg.det(cov.astype(np_dtype)) + epsilon) for cov in covs]
        self.register_buffer("cov_dets", torch.tensor(cov_dets, dtype=dtype))
        self.random_var_dim = self.means.shape[1]

    def get_mean(self):
        """Mean of the mixture [1, D]: the initial body pose when there is no regression prior
        (fit_single_frame.py:252)."""
        return torch.matmul(self.weights, self.means)

    def merged_log_likelihood(self, pose, betas=None):
        diff = pose.unsqueeze(dim=1) - self.means
        prec_diff = torch.einsum("mij,bmj->bmi", self.precisions, diff)
        quad = (prec_diff * diff).sum(dim=-1)
        ll = 0.5 * quad - torch.log(self.nll_weights)
        return torch.min(ll, dim=1)[0]

    def log_likelihood(self, pose, betas=None, *args, **kwargs):
        lls = []
        for idx in range(self.num_gaussians):
            diff = pose - self.means[idx]
            ll = torch.einsum("bi,bi->b", torch.einsum("bj,ji->bi", diff, self.precisions[idx]), diff)
            cov_term = torch.log(torch.det(self.covs[idx]) + self.epsilon)
            lls.append(ll + 0.5 * (cov_term + self.random_var_dim * self.pi_term))
        lls = torch.stack(lls, dim=1)
        min_idx = torch.argmin(lls, dim=1)
        return -torch.log(self.nll_weights[:, min_idx]) + lls[:, min_idx]

    def forward(self, pose, betas=None):
        return self.merged_log_likelihood(pose, betas) if self.use_merged else self.log_likelihood(pose, betas)
