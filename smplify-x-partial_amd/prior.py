"""Drop-in for the priors the shipped configurations use (smplifyx/prior.py:36-97):
`create_prior(prior_type in {'l2', 'angle', 'none', None})`.  The Gaussian-mixture prior
('gmm', prior.py:100-231) needs the un-shipped gmm_08.pkl and is a "next" row.
Inside the fitting loop these terms are evaluated by the HIP closure kernel; the modules
below give the same numbers stand-alone."""
import numpy as np
import torch
import torch.nn as nn

DEFAULT_DTYPE = torch.float32


def create_prior(prior_type, **kwargs):
    if prior_type == "gmm":
        raise NotImplementedError("MaxMixturePrior (gmm) is not built: needs gmm_08.pkl (SURVEY.md 8f-2)")
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
