"""ORACLE (test infrastructure): the Gaussian-mixture body pose prior of SMPLify,
smplifyx/prior.py:100-231 (MaxMixturePrior), restated for a mixture given as arrays.

Pinned by tests/golden/gmm.npz: the reference's own MaxMixturePrior, imported from /root/reference
and fed a synthetic gmm_08.pkl (tools/make_goldens.py: golden_gmm), evaluated with values and
autograd gradients at random poses in fp32 and fp64."""
import numpy as np
import torch


class MaxMixtureRef:
    def __init__(self, means, covars, weights, dtype=torch.float32):
        np_dtype = np.float32 if dtype == torch.float32 else np.float64
        covars_in, weights_in = np.asarray(covars), np.asarray(weights)
        means = np.asarray(means).astype(np_dtype)
        covs = covars_in.astype(np_dtype)
        self.means = torch.tensor(means, dtype=dtype)                                        # prior.py:144
        self.precisions = torch.tensor(np.stack([np.linalg.inv(c) for c in covs]).astype(np_dtype), dtype=dtype)   # :148-152
        sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in covars_in])                   # :155-156 (input precision)
        const = (2 * np.pi) ** (69 / 2.)                                                     # :157 (literal 69)
        nll = np.asarray(weights_in / (const * (sqrdets / sqrdets.min())))                   # :159-160
        self.nll_weights = torch.tensor(nll, dtype=dtype).unsqueeze(0)
        self.weights = torch.tensor(weights_in, dtype=dtype).unsqueeze(0)                    # :164

    def get_mean(self):                                                                       # :181-184
        return torch.matmul(self.weights, self.means)

    def __call__(self, pose, betas=None):                                                     # :186-201, use_merged=True
        diff = pose.unsqueeze(1) - self.means
        prec_diff = torch.einsum("mij,bmj->bmi", self.precisions, diff)
        quad = (prec_diff * diff).sum(-1)
        ll = 0.5 * quad - torch.log(self.nll_weights)
        return torch.min(ll, dim=1)[0]
