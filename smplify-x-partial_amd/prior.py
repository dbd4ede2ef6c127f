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


def read_mixture(prior_folder, num_gaussians):
    """`gmm_{num_gaussians:02d}.pkl` of SMPLify (a dict with means / covars / weights, or a pickled sklearn GMM; prior.py:118-135)
    -> the three arrays as the file holds them."""
    path = os.path.join(prior_folder, "gmm_{:02d}.pkl".format(num_gaussians))
    if not os.path.exists(path):
        raise FileNotFoundError('The path to the mixture prior "{}" does not exist'.format(path))
    with open(path, "rb") as f:
        return mixture_arrays(pickle.load(f, encoding="latin1"))


def mixture_arrays(gmm):
    if isinstance(gmm, dict):
        return np.asarray(gmm["means"]), np.asarray(gmm["covars"]), np.asarray(gmm["weights"])
    if "sklearn.mixture.gmm.GMM" in str(type(gmm)):
        return np.asarray(gmm.means_), np.asarray(gmm.covars_), np.asarray(gmm.weights_)
    raise ValueError("Unknown type for the prior: {}".format(type(gmm)))


def mixture_tables(means, covars, weights, np_dtype=np.float32, epsilon=1e-16):
    """Everything the mixture prior evaluates with, from the mixture's parameters: the table the device receives through
    sfx_batch_set_gmm[_form] and the stand-alone module registers as buffers (names and shapes are the interface:
    fitting.py:399-401 and the result files read them).  Arithmetic as prior.py:137-175, so that tests/golden/gmm*.npz -- the
    reference's own module -- pins every entry: the precisions are inverses of the covariances ROUNDED to the working type,
    the normalisation weights come from the unrounded ones, with the reference's literal 69 in the constant."""
    covs_t = np.asarray(covars).astype(np_dtype)
    root_dets = np.sqrt(np.linalg.det(np.asarray(covars)))
    return {
        "means": np.asarray(means).astype(np_dtype),
        "covs": covs_t,
        "precisions": np.linalg.inv(covs_t).astype(np_dtype),
        "nll_weights": (np.asarray(weights) / ((2 * np.pi) ** (69 / 2.) * (root_dets / root_dets.min())))[None],
        "weights": np.asarray(weights)[None],
        "pi_term": np.log(np.asarray(2 * np.pi, np_dtype)),
        "cov_dets": np.log(np.linalg.det(covs_t) + epsilon),
    }


class MaxMixturePrior(nn.Module):
    """Gaussian-mixture pose prior of SMPLify (prior.py:100-231): `gmm_{num_gaussians:02d}.pkl` from `prior_folder` (or `gmm`:
    the mixture as a dict / sklearn object), the reference's buffers (mixture_tables) and
    min_m [ 0.5 d_m^T P_m d_m - log nll_weights_m ] (`use_merged`, the reference's default) or the per-component form.
    In the fitting loop the evaluation is the HIP closure's (csrc/closure_body.h); this module gives the same numbers
    stand-alone."""

    def __init__(self, prior_folder="prior", num_gaussians=6, dtype=DEFAULT_DTYPE, epsilon=1e-16, use_merged=True,
                 gmm=None, **kwargs):
        super().__init__()
        np_dtype = {torch.float32: np.float32, torch.float64: np.float64}.get(dtype)
        if np_dtype is None:
            raise ValueError("Unknown float type {}".format(dtype))
        self.num_gaussians, self.epsilon, self.use_merged = num_gaussians, epsilon, use_merged
        arrays = mixture_arrays(gmm) if gmm is not None else read_mixture(prior_folder, num_gaussians)
        for name, table in mixture_tables(*arrays, np_dtype=np_dtype, epsilon=epsilon).items():
            self.register_buffer(name, torch.tensor(table, dtype=dtype))
        self.random_var_dim = self.means.shape[1]

    def get_mean(self):
        """Mean of the mixture [1, D]: the initial body pose when there is no regression prior (fit_single_frame.py:252)."""
        return self.weights @ self.means

    def _quadratic_forms(self, pose):
        d = pose[:, None, :] - self.means[None]                       # [B, M, D]
        return (torch.einsum("mij,bmj->bmi", self.precisions, d) * d).sum(-1)

    def merged_log_likelihood(self, pose, betas=None):
        return (0.5 * self._quadratic_forms(pose) - torch.log(self.nll_weights)).min(dim=1)[0]

    def log_likelihood(self, pose, betas=None, *args, **kwargs):
        """The per-component form (prior.py:203-231; written for batch 1 there, and here)."""
        const = 0.5 * (torch.log(torch.det(self.covs) + self.epsilon) + self.random_var_dim * self.pi_term)      # [M]
        lls = torch.stack([torch.einsum("bi,bi->b", (pose - self.means[m]) @ self.precisions[m], pose - self.means[m]) for m in range(self.num_gaussians)], 1) + const
        best = torch.argmin(lls, dim=1)
        return -torch.log(self.nll_weights[:, best]) + lls[:, best]

    def forward(self, pose, betas=None):
        return self.merged_log_likelihood(pose, betas) if self.use_merged else self.log_likelihood(pose, betas)
